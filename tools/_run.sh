set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD
timeout -k 10 800 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_parity.py tests/test_gpu_ceres.py -m gpu -x -q -k "band or direct or reference_default or ill or ceres or c4 or fallback" > $O/r04_u_tests.log 2>&1; echo "tests rc=$?" >> $O/r04_u_tests.log
tail -4 $O/r04_u_tests.log
grep -q "rc=0" $O/r04_u_tests.log || exit 1
rm -f $O/r04_q_ab.log
timeout -k 10 200 python bench.py --solver 0 --steps 40 --warmup 5 --no-cpu-baseline --config5 off >> $O/r04_q_ab.log 2>&1
timeout -k 10 200 python bench.py --solver 0 --config C4 --steps 20 --warmup 3 --no-cpu-baseline --config5 off >> $O/r04_q_ab.log 2>&1
timeout -k 10 200 python bench.py --framework 1 --steps 40 --warmup 5 --no-cpu-baseline --config5 off >> $O/r04_q_ab.log 2>&1
timeout -k 10 200 python bench.py --solver 0 --config C5 --windows-per-gpu 8 --steps 10 --warmup 2 --no-cpu-baseline --config5 off >> $O/r04_q_ab.log 2>&1
grep -h '"value"' $O/r04_q_ab.log | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); r = d.get('roofline') or {}
    print(' ', d['config']['workload'][:4], d['config']['windows_per_gpu'], 'value', d['value'], 'dom', r.get('kernel_symbol'), r.get('avg_launch_us'))
"
python3 tools/band_stamps.py C2 C4 2>&1 | grep -v "step " > $O/r04_q_band_stamps.log; cat $O/r04_q_band_stamps.log
