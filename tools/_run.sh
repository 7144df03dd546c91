set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r04_c_tests.log 2>&1; echo "tests rc=$?" >> $O/r04_c_tests.log
tail -8 $O/r04_c_tests.log
for CFG in "--solver 0" "--solver 0 --config C4 --iterations 10 --steps 6 --warmup 2" "--framework 1" "--config C5 --windows-per-gpu 8 --solver 0 --steps 10 --warmup 2"; do
  timeout -k 10 300 python bench.py $CFG --no-cpu-baseline >> $O/r04_c_bench.log 2>> $O/r04_c_bench.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r04_c_bench.log'):
    if not ln.startswith('{'): continue
    d=json.loads(ln); r=d.get('roofline') or {}
    print(d['config']['workload'][:60], '| value', d['value'], '|', r.get('kernel_symbol'), r.get('avg_launch_us'))
PY
python tools/stage_precision.py 559 > $O/r04_c_stage_precision.log 2>&1; cat $O/r04_c_stage_precision.log | cut -c1-400
