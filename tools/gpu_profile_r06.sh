#!/bin/bash
# Round-6 evidence: bench line (with `secondary`), rocprofv3 kernel trace + stats, PMC HBM traffic (separate FETCH / WRITE passes),
# SQ counters (MFMA utilisation), two-stream timeline, kernel A/B tables.  Summaries are copied into profiles/ afterwards.
# Fails loudly: a failed rocprofv3 pass must not leave a stale or missing summary behind.
set -euo pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r06
mkdir -p "$O"
one() {                      # the single file a pass must have produced
  local f
  f=$(find "$1" -name "$2" | head -1)
  if [ -z "$f" ]; then echo "gpu_profile_r06: no $2 under $1" >&2; exit 3; fi
  echo "$f"
}
timeout 900 python bench.py --steps 20 --warmup 5 > "$O/bench.json" 2> "$O/bench.err"
export FPLX_SIDE_STREAM=0            # one kernel at a time: clean per-launch durations and counters
PROG="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- $PROG > "$O/trace.log" 2>&1
python tools/trace_summary.py "$(one "$O/trace" '*kernel_trace.csv')" "$O/kernel_trace_by_shape.csv" > "$O/trace_summary.txt"
cp "$(one "$O/trace" '*kernel_stats.csv')" "$O/bench_kernel_stats.csv"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- $PROG > "$O/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- $PROG > "$O/pmc_write.log" 2>&1
PMC_COMMAND="FPLX_SIDE_STREAM=0 rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- $PROG" python tools/pmc_summary.py \
  "$(one "$O/pmc_fetch" '*counter_collection.csv')" "$(one "$O/pmc_write" '*counter_collection.csv')" "$O/pmc_hbm_traffic.json" > "$O/pmc_hbm_traffic.txt"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/pmc_sq1" -- $PROG > "$O/pmc_sq1.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_sq2" -- $PROG > "$O/pmc_sq2.log" 2>&1
mapfile -t SQ_CSVS < <(find "$O/pmc_sq1" "$O/pmc_sq2" -name "*counter_collection.csv")
if [ "${#SQ_CSVS[@]}" -lt 2 ]; then echo "gpu_profile_r06: ${#SQ_CSVS[@]} SQ counter csv(s) under $O/pmc_sq1, $O/pmc_sq2 (2 expected)" >&2; exit 3; fi
python tools/pmc_sq_summary.py "$O/pmc_sq_counters.json" "${SQ_CSVS[@]}" > "$O/pmc_sq_counters.txt" 2>&1
unset FPLX_SIDE_STREAM
# two-stream timeline of the step as shipped (both streams on)
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$O/trace2" -- $PROG > "$O/trace2.log" 2>&1
python tools/timeline_gaps.py "$(one "$O/trace2" '*kernel_trace.csv')" > "$O/two_stream_timeline.txt" 2>&1
( timeout 600 python tools/wgrad_bench.py 0 1 2 3 4; timeout 300 python tools/coresidency_probe.py ) > "$O/kernel_ab.txt" 2>&1
rm -rf "$O/trace" "$O/trace2" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_sq1" "$O/pmc_sq2"
head -c 1500 "$O/bench.json"; echo; head -12 "$O/trace_summary.txt"; head -8 "$O/pmc_hbm_traffic.txt"
