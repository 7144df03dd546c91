set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i "icache\|SQC_" | head -20 > $O/r04_icache_counters_avail.log
for CFG in PROD C2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/ic_$CFG -o pmc -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --config5 off > /dev/null 2> $O/r04_icache_$CFG.err
  python3 $ROOT/tools/pmc_summary.py $(find /tmp/ic_$CFG -name "*counter_collection.csv") > $O/r04_icache_$CFG.json 2>> $O/r04_icache_$CFG.err
done
head -c 600 $O/r04_icache_counters_avail.log; tail -3 $O/r04_icache_PROD.err | cut -c1-300
