#!/bin/bash
# Round-2 experiment batch 1 (one gpurun call): full GPU suite, PCG gather variants, hipGraph replay, soak divergence stepping, C4 lines.
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/r02_c_pytest_gpu.log 2>&1; tail -3 $O/r02_c_pytest_gpu.log
for GV in 0 1 2; do
  echo "== gather variant $GV" >> $O/r02_c_pcg_gather.log
  VISFS_BA_PCG_GATHER=$GV python bench.py --steps 40 --warmup 5 --no-cpu-baseline >> $O/r02_c_pcg_gather.log 2>&1
  VISFS_BA_PCG_GATHER=$GV python tools/pcg_stamps.py C2 >> $O/r02_c_pcg_gather.log 2>&1
  VISFS_BA_PCG_GATHER=$GV python bench.py --config C5 --windows-per-gpu 16 --steps 10 --warmup 2 --no-cpu-baseline >> $O/r02_c_pcg_gather.log 2>&1
done
echo "== eager" > $O/r02_c_hipgraph.log
python bench.py --steps 60 --warmup 10 --no-cpu-baseline >> $O/r02_c_hipgraph.log 2>&1
echo "== VISFS_BA_GRAPH=1" >> $O/r02_c_hipgraph.log
VISFS_BA_GRAPH=1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline >> $O/r02_c_hipgraph.log 2>&1
echo "== eager PROD" >> $O/r02_c_hipgraph.log
python bench.py --config PROD --iterations 10 --steps 100 --warmup 10 --no-cpu-baseline >> $O/r02_c_hipgraph.log 2>&1
echo "== VISFS_BA_GRAPH=1 PROD" >> $O/r02_c_hipgraph.log
VISFS_BA_GRAPH=1 python bench.py --config PROD --iterations 10 --steps 100 --warmup 10 --no-cpu-baseline >> $O/r02_c_hipgraph.log 2>&1
python tools/soak_diverge.py 756 781 1102 1108 1010 1034 1038 1056 1062 113 1141 1230 164 200 221 288 292 369 427 442 501 685 73 764 792 874 880 917 986 993 > $O/r02_c_soak_diverge.log 2>&1
tail -3 $O/r02_c_soak_diverge.log
for CFG in C4 C4R C4C; do
  python bench.py --config $CFG --steps 8 --warmup 2 --no-cpu-baseline > $O/r02_c_bench_$CFG.json 2> $O/r02_c_bench_$CFG.err
done
python bench.py > $O/r02_c_bench_c2.json 2> $O/r02_c_bench_c2.err
cat $O/r02_c_bench_c2.json
