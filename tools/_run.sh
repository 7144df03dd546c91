set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD

timeout -k 10 800 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_parity.py -m gpu -x -q -k "single_workgroup or lone_mid or three_pcg or never_depends" > $O/r04_u_tests.log 2>&1; echo "tests rc=$?" >> $O/r04_u_tests.log
tail -4 $O/r04_u_tests.log
rm -f $O/r04_q_ab.log
timeout -k 10 200 python bench.py --solver 0 --steps 40 --warmup 5 --no-cpu-baseline --config5 off >> $O/r04_q_ab.log 2>&1
VISFS_BA_PCG_CU=1 timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --config5 off >> $O/r04_q_ab.log 2>&1
VISFS_BA_PCG_CU=1 timeout -k 10 200 python bench.py --config C5 --windows-per-gpu 16 --steps 10 --warmup 2 --no-cpu-baseline --config5 off >> $O/r04_q_ab.log 2>&1
grep -h '"value"' $O/r04_q_ab.log | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); r = d.get('roofline') or {}
    print(' ', d['config']['workload'][:4], d['config']['windows_per_gpu'], 'value', d['value'], 'dom', r.get('kernel_symbol'), r.get('avg_launch_us'))
"
