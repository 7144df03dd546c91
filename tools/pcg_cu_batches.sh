#!/bin/bash
# k_pcg_cu (the whole PCG of a window in ONE workgroup, opt-in VISFS_BA_PCG_CU=1) against the default k_pcg1 (one wavefront per block
# row, cross-workgroup hand-offs) in batches of C2-size windows sharing every launch: tools/pcg_cu_batches.sh <tag>
O=gpurun_out; TAG=${1:-r02_v3}; mkdir -p $O
for E in 0 1; do
  for W in 8 16 32; do
    echo "== VISFS_BA_PCG_CU=$E, $W windows" >> $O/${TAG}_pcg_cu_batches.log
    VISFS_BA_PCG_CU=$E python bench.py --config C5 --windows-per-gpu $W --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null >> $O/${TAG}_pcg_cu_batches.log
  done
done
grep -h '"value"\|^==' $O/${TAG}_pcg_cu_batches.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('=='): print(ln.strip(), end=': '); continue
    print(json.loads(ln)['value'], 'it/s')
"
