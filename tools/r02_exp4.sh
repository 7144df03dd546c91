#!/bin/bash
# Round-2 experiment batch 4: DIP (diagonal blocks finalised in k_schur_partial, k_schur_finalize dropped) on / off.
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/r02_f_pytest_gpu.log 2>&1; tail -3 $O/r02_f_pytest_gpu.log
for DIP in 1 0; do
  echo "== VISFS_BA_DIP=$DIP" >> $O/r02_f_dip.log
  VISFS_BA_DIP=$DIP python bench.py --steps 60 --warmup 10 --no-cpu-baseline >> $O/r02_f_dip.log 2>&1
  VISFS_BA_DIP=$DIP python bench.py --config C3 --steps 40 --warmup 5 --no-cpu-baseline >> $O/r02_f_dip.log 2>&1
  VISFS_BA_DIP=$DIP python bench.py --config C5 --windows-per-gpu 8 --steps 10 --warmup 2 --no-cpu-baseline >> $O/r02_f_dip.log 2>&1
  VISFS_BA_DIP=$DIP python bench.py --config C5 --windows-per-gpu 16 --steps 10 --warmup 2 --no-cpu-baseline >> $O/r02_f_dip.log 2>&1
done
python tools/soak_diverge.py 756 781 1102 1108 1010 1034 1038 1056 1062 113 1141 1230 164 200 221 288 292 369 427 442 501 685 73 764 792 874 880 917 986 993 > $O/r02_f_soak_diverge.log 2>&1
tail -1 $O/r02_f_soak_diverge.log
PYTHONPATH=. python tools/e2e_breakdown.py > $O/r02_f_e2e_breakdown.log 2>&1; cat $O/r02_f_e2e_breakdown.log
grep -h '"value"' $O/r02_f_dip.log | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); r = d.get('roofline') or {}
    print(d['config']['workload'][:3], d['config']['windows_per_gpu'], 'value', d['value'], 'dom', r.get('kernel'), r.get('avg_launch_us'), d['kernel_us_per_step_calibration'])
"
