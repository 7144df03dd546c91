set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD
timeout -k 10 800 python -m pytest tests/test_gpu_workloads.py -m gpu -x -q -k "single_workgroup or lone_mid or three_pcg or never_depends" > $O/r04_u_tests.log 2>&1; echo "tests rc=$?" >> $O/r04_u_tests.log
tail -4 $O/r04_u_tests.log
grep -q "rc=0" $O/r04_u_tests.log || exit 1
python3 tools/pcg_cu_stamps.py C2 > $O/r04_pcg_cu_stamps3.log 2>&1; cat $O/r04_pcg_cu_stamps3.log
rm -f $O/r04_s_ab.log
run_set() {
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --config5 off >> $O/r04_s_ab.log 2>&1
  timeout -k 10 200 python bench.py --config C3 --steps 40 --warmup 5 --no-cpu-baseline --config5 off >> $O/r04_s_ab.log 2>&1
  timeout -k 10 200 python bench.py --config C5 --windows-per-gpu 8 --steps 10 --warmup 2 --no-cpu-baseline --config5 off >> $O/r04_s_ab.log 2>&1
  timeout -k 10 200 python bench.py --config C5 --windows-per-gpu 16 --steps 10 --warmup 2 --no-cpu-baseline --config5 off >> $O/r04_s_ab.log 2>&1
}
echo "== default" >> $O/r04_s_ab.log; run_set
echo "== VISFS_BA_PCG_CU=1" >> $O/r04_s_ab.log; VISFS_BA_PCG_CU=1 run_set
grep -h '"value"\|^==' $O/r04_s_ab.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('=='): print(ln.strip()); continue
    d = json.loads(ln); r = d.get('roofline') or {}
    print(' ', d['config']['workload'][:4], d['config']['windows_per_gpu'], 'value', d['value'], 'dom', r.get('kernel_symbol'), r.get('avg_launch_us'), {k: round(v) for k, v in d['kernel_us_per_step_calibration'].items()})
"
