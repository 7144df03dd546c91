# scratch: the command of the last ad-hoc GPU run (gpurun -- 'bash tools/_run.sh'); the round's collections are tools/collect_profiles.sh and tools/_soak.sh
set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD
rm -f $O/r04_z_ab.log
run_set() {
  timeout -k 10 200 python bench.py --config C5 --windows-per-gpu 8 --steps 10 --warmup 2 --no-cpu-baseline --config5 off >> $O/r04_z_ab.log 2>&1
  timeout -k 10 200 python bench.py --config C5 --windows-per-gpu 16 --steps 10 --warmup 2 --no-cpu-baseline --config5 off >> $O/r04_z_ab.log 2>&1
  timeout -k 10 200 python bench.py --config C5 --windows-per-gpu 8 --solver 0 --steps 10 --warmup 2 --no-cpu-baseline --config5 off >> $O/r04_z_ab.log 2>&1
}
for rep in 1 2; do
echo "== default" >> $O/r04_z_ab.log; run_set
echo "== VISFS_BA_DECIDE_FUSED=1" >> $O/r04_z_ab.log; VISFS_BA_DECIDE_FUSED=1 run_set
done
grep -h '"value"\|^==' $O/r04_z_ab.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('=='): print(ln.strip()); continue
    d = json.loads(ln)
    print(' ', d['config']['workload'][:4], d['config']['windows_per_gpu'], 'solver', d['config']['solver'], 'value', d['value'])
"
