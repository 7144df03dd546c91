# scratch: the command of the last ad-hoc GPU run (gpurun -- 'bash tools/_run.sh'); the round's collections are tools/collect_profiles.sh and tools/_soak.sh
set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r04_final_pytest_gpu.log 2>&1; echo "tests rc=$?" >> $O/r04_final_pytest_gpu.log
tail -4 $O/r04_final_pytest_gpu.log
grep -q "rc=0" $O/r04_final_pytest_gpu.log || exit 1
( timeout -k 10 200 python3 tools/soak_batch.py 1500 1800 12 > $O/r04_soak_batch.log 2>&1; echo "rc=$?" >> $O/r04_soak_batch.log; tail -2 $O/r04_soak_batch.log )
bash tools/collect_profiles.sh r04_v2 > $O/r04_v2_collect.log 2>&1; tail -2 $O/r04_v2_collect.log
