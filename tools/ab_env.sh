#!/bin/bash
# A/B of an environment switch over the bench configurations (one gpurun call): tools/ab_env.sh <tag> VAR=value [VAR=value ...]
# first the default build as it is, then the same with the variables exported.
O=gpurun_out; TAG=$1; shift
mkdir -p $O
run_all() {
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config C3 --steps 40 --warmup 5 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config C4 --steps 10 --warmup 2 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config C4R --steps 10 --warmup 2 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config C5 --windows-per-gpu 8 --steps 10 --warmup 2 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config C5 --windows-per-gpu 16 --steps 10 --warmup 2 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
  python bench.py --config PROD --iterations 10 --steps 100 --warmup 10 --no-cpu-baseline >> $O/${TAG}_ab.log 2>&1
}
echo "== default" >> $O/${TAG}_ab.log
run_all
echo "== $*" >> $O/${TAG}_ab.log
export "$@"
run_all
grep -h '"value"\|^==' $O/${TAG}_ab.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('=='): print(ln.strip()); continue
    d = json.loads(ln); r = d.get('roofline') or {}
    print(' ', d['config']['workload'][:4], d['config']['windows_per_gpu'], 'value', d['value'], 'dom', r.get('kernel'), r.get('avg_launch_us'), {k: round(v) for k, v in d['kernel_us_per_step_calibration'].items()})
"
