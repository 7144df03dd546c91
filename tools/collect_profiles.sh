#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: the headline bench line, the rocprofv3 kernel statistics of the same
# command and the two PMC passes (separate runs, --kernel-trace only) for C2 and C4.  Outputs go to gpurun_out/<tag>_*;
# copy what should be judged into profiles/.
set -eo pipefail
TAG=${1:-r01_v5}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$PWD
timeout -k 10 400 python3 bench.py > "$OUT/${TAG}_c2_bench.json"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats_c2" -o c2 -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline > "$OUT/${TAG}_c2_bench_under_rocprof.json"
for CFG in C2 C4; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$OUT/${TAG}_pmc_${CFG}_${CTR}" -o pmc -- python3 "$GRAFT_REPO_ROOT/bench.py" --config $CFG --steps 4 --warmup 1 --no-cpu-baseline > /dev/null
  done
done
cd "$GRAFT_REPO_ROOT"
for CFG in C2 C4; do
  python3 tools/pmc_summary.py $(find "$OUT/${TAG}_pmc_${CFG}_FETCH_SIZE" "$OUT/${TAG}_pmc_${CFG}_WRITE_SIZE" -name "*counter_collection.csv") > "$OUT/${TAG}_pmc_${CFG}_summary.json"
done
python3 tools/pmc_traffic.py "$OUT/${TAG}_pmc_traffic.json" C2="$OUT/${TAG}_pmc_C2_summary.json" C4="$OUT/${TAG}_pmc_C4_summary.json"
cp $(find "$OUT/${TAG}_stats_c2" -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_c2_kernel_stats.csv"
echo done
