#!/bin/bash
# A/B of two builds of the HIP library over the bench configurations (one gpurun call): tools/ab_bench.sh <tag> <variant-lib-name>
# default = visfs_amd/lib/libvisfs_ba_hip.so, variant = visfs_amd/lib/libvisfs_ba_hip_<name>.so (tools/build_variant.sh).
O=gpurun_out; TAG=$1; shift; VAR="$@"
mkdir -p $O
for V in default $VAR; do
  if [ $V != default ]; then export VISFS_BA_LIB=$PWD/visfs_amd/lib/libvisfs_ba_hip_$V.so; else unset VISFS_BA_LIB; fi
  echo "== $V" >> $O/${TAG}_ab.log
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config C3 --steps 40 --warmup 5 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config C4 --steps 10 --warmup 2 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config C4R --steps 10 --warmup 2 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config C5 --windows-per-gpu 8 --steps 10 --warmup 2 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config C5 --windows-per-gpu 16 --steps 10 --warmup 2 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config PROD --iterations 10 --steps 100 --warmup 10 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
done
unset VISFS_BA_LIB
grep -h '"value"\|^==' $O/${TAG}_ab.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('=='): print(ln.strip()); continue
    d = json.loads(ln); r = d.get('roofline') or {}
    print(' ', d['config']['workload'][:4], d['config']['windows_per_gpu'], 'value', d['value'], 'dom', r.get('kernel'), r.get('avg_launch_us'), {k: round(v) for k, v in d['kernel_us_per_step_calibration'].items()})
"
