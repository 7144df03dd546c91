#!/bin/bash
# Round-2 experiment batch 2: XCD-local PCG hand-off (gather variant 3), default hipGraph replay, soak divergence stepping.
O=gpurun_out
mkdir -p $O
for GV in 1 3; do
  echo "== gather variant $GV" >> $O/r02_d_pcg_gather.log
  VISFS_BA_PCG_GATHER=$GV python bench.py --steps 40 --warmup 5 --no-cpu-baseline >> $O/r02_d_pcg_gather.log 2>&1
  VISFS_BA_PCG_GATHER=$GV VISFS_BA_GRAPH=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline >> $O/r02_d_pcg_gather.log 2>&1
  VISFS_BA_PCG_GATHER=$GV python tools/pcg_stamps.py C2 >> $O/r02_d_pcg_gather.log 2>&1
  VISFS_BA_PCG_GATHER=$GV python bench.py --config C3 --steps 40 --warmup 5 --no-cpu-baseline >> $O/r02_d_pcg_gather.log 2>&1
done
VISFS_BA_PCG_GATHER=3 python -m pytest tests -m gpu -x -q > $O/r02_d_pytest_gpu_gv3.log 2>&1; tail -3 $O/r02_d_pytest_gpu_gv3.log
python tools/soak_diverge.py 756 781 1102 1108 1010 1034 1038 1056 1062 113 1141 1230 164 200 221 288 292 369 427 442 501 685 73 764 792 874 880 917 986 993 > $O/r02_d_soak_diverge.log 2>&1
grep -c "^seed" $O/r02_d_soak_diverge.log
python bench.py --config PROD --iterations 10 --steps 100 --warmup 10 --no-cpu-baseline > $O/r02_d_bench_prod.json 2>&1
tail -c 600 $O/r02_d_pcg_gather.log
