# scratch: the command of the last ad-hoc GPU run (gpurun -- 'bash tools/_run.sh'); the round's collections are tools/collect_profiles.sh and tools/_soak.sh
set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "tests rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
