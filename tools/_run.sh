set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_workloads.py -m gpu -x -q -k "timed_out or fall_back or failed_upload or fused_speculative or c5_share" > $O/r04_d_tests.log 2>&1; echo "tests rc=$?" >> $O/r04_d_tests.log
tail -12 $O/r04_d_tests.log
timeout -k 10 400 python bench.py > $O/r04_d_bench.json 2> $O/r04_d_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_d_bench.json') if l.startswith('{')][0])
print('value', d['value'], 'per_frame', d.get('per_frame_call'), '\nconfig5', d.get('config5'))
print(json.dumps(d['cpu_baseline'], indent=1))
PY
