#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: the headline bench line, the rocprofv3 kernel statistics of the same
# command, the kernel statistics of the 16-window batched regime, the two HBM-traffic PMC passes (separate runs, --kernel-trace
# only) for C2 and C4 / C4R, and three SQ counter passes for C2 and C4R.  Outputs go to gpurun_out/<tag>_*; copy what should be
# judged into profiles/.   usage: tools/collect_profiles.sh r02_v2
set -o pipefail
TAG=${1:-r02_v2}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$PWD
ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 400 python3 bench.py > "$OUT/${TAG}_c2_bench.json" 2> "$OUT/${TAG}_c2_bench.err"
for CFG in C4 C4R; do
  timeout -k 10 300 python3 bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_bench_$CFG.json" 2>> "$OUT/${TAG}_c2_bench.err"
done
# Optimizer/Framework=1 (the Ceres branch): C2 and the production window, one pass of <= 20 trust-region iterations
for CFG in C2 PROD; do
  timeout -k 10 300 python3 bench.py --config $CFG --framework 1 --steps 20 --warmup 3 > "$OUT/${TAG}_bench_ceres_$CFG.json" 2>> "$OUT/${TAG}_c2_bench.err"
done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats_c2" -o c2 -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/${TAG}_c2_bench_under_rocprof.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats_c4r" -o c4r -- python3 "$ROOT/bench.py" --config C4R --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_c4r_bench_under_rocprof.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats_c5x16" -o c5x16 -- python3 "$ROOT/bench.py" --config C5 --windows-per-gpu 16 --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_c5x16_bench_under_rocprof.json"
for CFG in C2 C4 C4R; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$OUT/${TAG}_pmc_${CFG}_${CTR}" -o pmc -- python3 "$ROOT/bench.py" --config $CFG --steps 4 --warmup 1 --no-cpu-baseline > /dev/null
  done
done
# SQ counters: three passes (8 SQ slots per pass; never combined with the trace domains gpurun refuses)
for CFG in C2 C4R; do
  n=0
  for SET in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
    n=$((n + 1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/${TAG}_sq_${CFG}_$n" -o pmc -- python3 "$ROOT/bench.py" --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
  done
done
cd "$ROOT"
for CFG in C2 C4 C4R; do
  python3 tools/pmc_summary.py $(find "$OUT/${TAG}_pmc_${CFG}_FETCH_SIZE" "$OUT/${TAG}_pmc_${CFG}_WRITE_SIZE" -name "*counter_collection.csv") > "$OUT/${TAG}_pmc_${CFG}_summary.json"
done
for CFG in C2 C4R; do
  python3 tools/pmc_summary.py $(find "$OUT/${TAG}_sq_${CFG}_1" "$OUT/${TAG}_sq_${CFG}_2" "$OUT/${TAG}_sq_${CFG}_3" -name "*counter_collection.csv") > "$OUT/${TAG}_sq_${CFG}_counters.json"
done
python3 tools/pmc_traffic.py "$OUT/${TAG}_pmc_traffic.json" C2="$OUT/${TAG}_pmc_C2_summary.json" C4="$OUT/${TAG}_pmc_C4_summary.json" C4R="$OUT/${TAG}_pmc_C4R_summary.json"
for S in c2 c4r c5x16; do
  cp $(find "$OUT/${TAG}_stats_$S" -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_${S}_kernel_stats.csv"
done
echo done
