#!/bin/bash
# Probe of the GPU box promised in BASELINE.md §3 / SURVEY §8d: is a system g2o (or Eigen / Ceres) present, which host CPU, how many cores.
# Output goes to gpurun_out/probe_box.log; the summary judged is copied to profiles/.
out=${1:-gpurun_out/probe_box.log}
mkdir -p "$(dirname "$out")"
{
  echo "== date"; date -u
  echo "== cpu"; grep -m1 'model name' /proc/cpuinfo; echo "nproc=$(nproc)"; echo "sockets=$(grep 'physical id' /proc/cpuinfo | sort -u | wc -l)"
  echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
  echo "== mem"; grep -E 'MemTotal|MemAvailable' /proc/meminfo
  echo "== gpu"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E 'Marketing Name|gfx9|Compute Unit' | sort | uniq -c
  echo "== g2o / Eigen / Ceres / SuiteSparse / OpenCV search (find / ...)"
  find / -xdev \( -name 'block_solver.h' -o -name 'libg2o*' -o -name 'sparse_optimizer.h' -o -name 'signature_of_eigen3_matrix_library' -o -name 'ceres.h' -o -name 'libceres*' -o -name 'cs.h' -o -name 'libcxsparse*' -o -name 'cholmod.h' -o -name 'libopencv_core*' \) 2>/dev/null | grep -v '^/proc' | head -50
  echo "(end of search; empty list above = none found)"
  echo "== dpkg"; dpkg -l 2>/dev/null | grep -iE 'g2o|eigen|ceres|suitesparse|opencv' | head
  echo "(end of dpkg)"
} > "$out" 2>&1
cat "$out"
