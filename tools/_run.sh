set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD
rm -f $O/r04_w_ab.log
for rep in 1 2; do
for V in v1 v2 v3; do
  export VISFS_BA_LIB=$PWD/visfs_amd/lib/libvisfs_ba_hip_$V.so
  echo "== $V" >> $O/r04_w_ab.log
  timeout -k 10 200 python bench.py --solver 0 --steps 40 --warmup 5 --no-cpu-baseline --config5 off >> $O/r04_w_ab.log 2>&1
  timeout -k 10 200 python bench.py --solver 0 --config C4 --steps 20 --warmup 3 --no-cpu-baseline --config5 off >> $O/r04_w_ab.log 2>&1
done
done
grep -h '"value"\|^==' $O/r04_w_ab.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('=='): print(ln.strip()); continue
    d = json.loads(ln); r = d.get('roofline') or {}
    print(' ', d['config']['workload'][:4], 'value', d['value'], 'dom', r.get('kernel_symbol'), r.get('avg_launch_us'))
"
