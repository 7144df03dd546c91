set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD
python3 tools/band_stamps.py C2 > $O/r04_p_band_stamps.log 2>&1; cat $O/r04_p_band_stamps.log
