set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r04_r_pytest_gpu.log 2>&1; echo "tests rc=$?" >> $O/r04_r_pytest_gpu.log
tail -4 $O/r04_r_pytest_gpu.log
grep -q "rc=0" $O/r04_r_pytest_gpu.log || exit 1
( timeout -k 10 400 python3 tools/soak_random.py 6000 7200 > $O/r04_soak_random.log 2>&1; echo "rc=$?" >> $O/r04_soak_random.log; tail -3 $O/r04_soak_random.log )
( timeout -k 10 250 python3 tools/soak_ceres.py 2400 3000 0 > $O/r04_soak_ceres.log 2>&1; echo "rc=$?" >> $O/r04_soak_ceres.log; tail -2 $O/r04_soak_ceres.log )
( timeout -k 10 250 python3 tools/soak_ceres.py 2400 3000 1 > $O/r04_soak_dogleg.log 2>&1; echo "rc=$?" >> $O/r04_soak_dogleg.log; tail -2 $O/r04_soak_dogleg.log )
( timeout -k 10 100 python3 tools/stage_precision.py 3363 1102 756 > $O/r04_stage_precision.log 2>&1; tail -8 $O/r04_stage_precision.log )
