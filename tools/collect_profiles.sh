#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: the headline bench line (with config 5's per-GPU share in the same run), the bench
# lines of the other workloads WITH their CPU baseline and parity legs, the direct-solver / Ceres lines, the per-frame breakdown, the
# rocprofv3 kernel statistics of the headline, of C4R, of the reference-default solver and of the 16-window batch, the two HBM-traffic
# PMC passes (separate runs, --kernel-trace only) and three SQ counter passes.  Outputs go to gpurun_out/<tag>_*; copy what should be
# judged into profiles/.   usage: tools/collect_profiles.sh r03_v1
set -e -o pipefail
TAG=${1:-r04_v2}
OUT=$PWD/gpurun_out
RAW=/tmp/visfs_prof_$TAG            # rocprofv3's raw traces (hundreds of MB): only summaries travel back (gpurun merges <= 64 MiB of gpurun_out/)
rm -rf "$RAW"; mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp PYTHONPATH=$PWD
ROOT=${GRAFT_REPO_ROOT:-$PWD}
step() { echo "[collect] $*" >&2; }
step "headline"
timeout -k 10 400 python3 bench.py --config5 on > "$OUT/${TAG}_c2_bench.json" 2> "$OUT/${TAG}_c2_bench.err"
for CFG in C1 PROD C3 C4 C4R; do
  step "bench $CFG"
  timeout -k 10 400 python3 bench.py --config $CFG --steps 10 --warmup 2 > "$OUT/${TAG}_bench_$CFG.json" 2>> "$OUT/${TAG}_c2_bench.err"
done
step "reference-default solver (Optimizer/Solver=0) and the Ceres branch"
timeout -k 10 300 python3 bench.py --config C2 --solver 0 --steps 20 --warmup 3 > "$OUT/${TAG}_bench_C2_solver0.json" 2>> "$OUT/${TAG}_c2_bench.err"
timeout -k 10 300 python3 bench.py --config C4 --solver 0 --iterations 10 --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_bench_C4_solver0.json" 2>> "$OUT/${TAG}_c2_bench.err"
for CFG in C2 PROD; do
  timeout -k 10 300 python3 bench.py --config $CFG --framework 1 --steps 20 --warmup 3 > "$OUT/${TAG}_bench_ceres_$CFG.json" 2>> "$OUT/${TAG}_c2_bench.err"
done
for CFG in C2 PROD; do
  timeout -k 10 300 python3 bench.py --config $CFG --framework 1 --trust-region 1 --steps 20 --warmup 3 > "$OUT/${TAG}_bench_dogleg_$CFG.json" 2>> "$OUT/${TAG}_c2_bench.err"
done
timeout -k 10 300 python3 bench.py --config C5 --windows-per-gpu 8 --solver 0 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_bench_C5x8_solver0.json" 2>> "$OUT/${TAG}_c2_bench.err"
step "wide-band window: the dense blocked Cholesky (fp64 MFMA SYRK)"
timeout -k 10 300 python3 bench.py --config WB --solver 0 --steps 20 --warmup 3 > "$OUT/${TAG}_bench_WB_solver0.json" 2>> "$OUT/${TAG}_c2_bench.err"
step "batches on one GPU"
for B in 8 16; do
  timeout -k 10 300 python3 bench.py --config C5 --windows-per-gpu $B --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_bench_C5x$B.json" 2>> "$OUT/${TAG}_c2_bench.err"
done
step "the two handle tunings (visfs_ba_set_tuning): batches on the latency tuning, one window on the throughput tuning"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --config5 off --tuning throughput > "$OUT/${TAG}_bench_C2_tuned_throughput.json" 2>> "$OUT/${TAG}_c2_bench.err"
for B in 8 16; do
  timeout -k 10 300 python3 bench.py --config C5 --windows-per-gpu $B --steps 10 --warmup 2 --no-cpu-baseline --tuning latency > "$OUT/${TAG}_bench_C5x${B}_tuned_latency.json" 2>> "$OUT/${TAG}_c2_bench.err"
done
step "per-frame call path"
timeout -k 10 300 python3 tools/e2e_breakdown.py > "$OUT/${TAG}_e2e_breakdown.log" 2>&1
for CFG in PROD C1 C2 C4; do timeout -k 10 120 python3 tools/frame_loop.py $CFG 30; done > "$OUT/${TAG}_frame_loop.log" 2>&1
cd /tmp
step "kernel statistics"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/${TAG}_stats_c2" -o c2 -- python3 "$ROOT/bench.py" --no-cpu-baseline --config5 off > "$OUT/${TAG}_c2_bench_under_rocprof.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/${TAG}_stats_c4r" -o c4r -- python3 "$ROOT/bench.py" --config C4R --steps 6 --warmup 2 --no-cpu-baseline --config5 off > "$OUT/${TAG}_c4r_bench_under_rocprof.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/${TAG}_stats_c2s0" -o c2s0 -- python3 "$ROOT/bench.py" --solver 0 --steps 20 --warmup 3 --no-cpu-baseline --config5 off > "$OUT/${TAG}_c2s0_bench_under_rocprof.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/${TAG}_stats_c5x16" -o c5x16 -- python3 "$ROOT/bench.py" --config C5 --windows-per-gpu 16 --steps 6 --warmup 2 --no-cpu-baseline --config5 off > "$OUT/${TAG}_c5x16_bench_under_rocprof.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/${TAG}_stats_wbs0" -o wbs0 -- python3 "$ROOT/bench.py" --config WB --solver 0 --steps 10 --warmup 2 --no-cpu-baseline --config5 off > "$OUT/${TAG}_wbs0_bench_under_rocprof.json"
step "MFMA utilisation of the dense Cholesky (wide-band window)"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$RAW/${TAG}_mfma_wbs0" -o pmc -- python3 "$ROOT/bench.py" --config WB --solver 0 --steps 4 --warmup 1 --no-cpu-baseline --config5 off > /dev/null
step "PMC traffic"
for CFG in C2 C4 C4R; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$RAW/${TAG}_pmc_${CFG}_${CTR}" -o pmc -- python3 "$ROOT/bench.py" --config $CFG --steps 4 --warmup 1 --no-cpu-baseline --config5 off > /dev/null
  done
done
step "SQ counters"
# three passes (8 SQ slots per pass; never combined with the trace domains gpurun refuses)
for CFG in C2 C4R; do
  n=0
  for SET in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
    n=$((n + 1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$RAW/${TAG}_sq_${CFG}_$n" -o pmc -- python3 "$ROOT/bench.py" --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --config5 off > /dev/null
  done
done
cd "$ROOT"
for CFG in C2 C4 C4R; do
  python3 tools/pmc_summary.py $(find "$RAW/${TAG}_pmc_${CFG}_FETCH_SIZE" "$RAW/${TAG}_pmc_${CFG}_WRITE_SIZE" -name "*counter_collection.csv") > "$OUT/${TAG}_pmc_${CFG}_summary.json"
done
for CFG in C2 C4R; do
  python3 tools/pmc_summary.py $(find "$RAW/${TAG}_sq_${CFG}_1" "$RAW/${TAG}_sq_${CFG}_2" "$RAW/${TAG}_sq_${CFG}_3" -name "*counter_collection.csv") > "$OUT/${TAG}_sq_${CFG}_counters.json"
done
python3 tools/pmc_traffic.py "$OUT/${TAG}_pmc_traffic.json" C2="$OUT/${TAG}_pmc_C2_summary.json" C4="$OUT/${TAG}_pmc_C4_summary.json" C4R="$OUT/${TAG}_pmc_C4R_summary.json"
python3 tools/pmc_summary.py $(find "$RAW/${TAG}_mfma_wbs0" -name "*counter_collection.csv") > "$OUT/${TAG}_mfma_wbs0_summary.json"
for S in c2 c4r c2s0 c5x16 wbs0; do
  F=$(find "$RAW/${TAG}_stats_$S" -name "*kernel_stats.csv" | head -1)
  test -n "$F" || { echo "no kernel statistics for $S" >&2; exit 1; }
  cp "$F" "$OUT/${TAG}_${S}_kernel_stats.csv"
done
echo done
