set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r04_i_tests.log 2>&1; echo "tests rc=$?" >> $O/r04_i_tests.log
tail -6 $O/r04_i_tests.log
