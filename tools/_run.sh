set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "optimize_parity or stage_parity or repeated or batch or fused_spec or decision or timed" > $O/r04_n_tests.log 2>&1; echo "tests rc=$?" >> $O/r04_n_tests.log
tail -6 $O/r04_n_tests.log
grep -q "rc=0" $O/r04_n_tests.log || exit 1
rm -f $O/r04_n_ab.log
run_set() {
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --config5 off >> $O/r04_n_ab.log 2>&1
  timeout -k 10 200 python bench.py --config C3 --steps 40 --warmup 5 --no-cpu-baseline --config5 off >> $O/r04_n_ab.log 2>&1
  timeout -k 10 200 python bench.py --config C5 --windows-per-gpu 8 --steps 10 --warmup 2 --no-cpu-baseline --config5 off >> $O/r04_n_ab.log 2>&1
  timeout -k 10 200 python bench.py --config C5 --windows-per-gpu 16 --steps 10 --warmup 2 --no-cpu-baseline --config5 off >> $O/r04_n_ab.log 2>&1
}
echo "== fused finalize + pcg1 (default)" >> $O/r04_n_ab.log; run_set
echo "== VISFS_BA_FIN_PCG=0" >> $O/r04_n_ab.log; VISFS_BA_FIN_PCG=0 run_set
echo "== fused again" >> $O/r04_n_ab.log; run_set
grep -h '"value"\|^==' $O/r04_n_ab.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('=='): print(ln.strip()); continue
    d = json.loads(ln); r = d.get('roofline') or {}
    print(' ', d['config']['workload'][:4], d['config']['windows_per_gpu'], 'value', d['value'], 'per_frame', (d.get('per_frame_call') or {}).get('ms_per_call'), 'dom', r.get('kernel'), r.get('avg_launch_us'), {k: round(v) for k, v in d['kernel_us_per_step_calibration'].items()})
"
