set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD
timeout -k 10 800 python -m pytest tests/test_gpu_workloads.py -m gpu -x -q -k "odd_shapes or banded" > $O/r04_u_tests.log 2>&1; echo "tests rc=$?" >> $O/r04_u_tests.log
tail -12 $O/r04_u_tests.log
