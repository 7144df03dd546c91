#!/bin/bash
# Round-2 experiment batch 5: tile records (N per observation, VISFS_BA_TILE_N=1, the default build) vs the round-1 seeds (variant build).
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/r02_g_pytest_gpu.log 2>&1; tail -3 $O/r02_g_pytest_gpu.log
for V in default seed; do
  if [ $V = seed ]; then export VISFS_BA_LIB=$PWD/visfs_amd/lib/libvisfs_ba_hip_seed.so; else unset VISFS_BA_LIB; fi
  echo "== $V" >> $O/r02_g_tile.log
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline >> $O/r02_g_tile.log 2>&1
  python bench.py --config C3 --steps 40 --warmup 5 --no-cpu-baseline >> $O/r02_g_tile.log 2>&1
  python bench.py --config C4 --steps 10 --warmup 2 --no-cpu-baseline >> $O/r02_g_tile.log 2>&1
  python bench.py --config C4R --steps 10 --warmup 2 --no-cpu-baseline >> $O/r02_g_tile.log 2>&1
  python bench.py --config C5 --windows-per-gpu 8 --steps 10 --warmup 2 --no-cpu-baseline >> $O/r02_g_tile.log 2>&1
  python bench.py --config C5 --windows-per-gpu 16 --steps 10 --warmup 2 --no-cpu-baseline >> $O/r02_g_tile.log 2>&1
  python bench.py --config PROD --iterations 10 --steps 100 --warmup 10 --no-cpu-baseline >> $O/r02_g_tile.log 2>&1
done
unset VISFS_BA_LIB
python tools/soak_diverge.py 756 781 1102 1108 1010 1034 1038 1056 1062 113 1141 1230 164 200 221 288 292 369 427 442 501 685 73 764 792 874 880 917 986 993 > $O/r02_g_soak_diverge.log 2>&1
tail -1 $O/r02_g_soak_diverge.log
grep -h '"value"\|^==' $O/r02_g_tile.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('=='): print(ln.strip()); continue
    d = json.loads(ln); r = d.get('roofline') or {}
    print(' ', d['config']['workload'][:4], d['config']['windows_per_gpu'], 'value', d['value'], 'dom', r.get('kernel'), r.get('avg_launch_us'), {k: round(v) for k, v in d['kernel_us_per_step_calibration'].items()})
"
