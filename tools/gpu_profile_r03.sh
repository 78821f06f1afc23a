#!/bin/bash
# Round-3 evidence: bench line, rocprofv3 kernel trace + stats, PMC HBM traffic (separate FETCH / WRITE passes), SQ counters
# (MFMA utilisation), secondary measurements.  Summaries are copied into profiles/ by hand afterwards.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r03
mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
( timeout 300 python tools/fpl_infer_bench.py; timeout 300 python tools/cfg5_bench.py; timeout 300 python tools/cfg25_bench.py; timeout 300 python tools/data_path_bench.py ) > $O/secondary.txt 2>&1
export FPLX_SIDE_STREAM=0            # one kernel at a time: clean per-launch durations and counters
PROG="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $PROG > $O/trace.log 2>&1
python tools/trace_summary.py $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/kernel_trace_by_shape.csv > $O/trace_summary.txt
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $PROG > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $PROG > $O/pmc_write.log 2>&1
PMC_COMMAND="FPLX_SIDE_STREAM=0 rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- $PROG" python tools/pmc_summary.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_hbm_traffic.json > $O/pmc_hbm_traffic.txt
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -- $PROG > $O/pmc_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -- $PROG > $O/pmc_sq2.log 2>&1
python tools/pmc_sq_summary.py $O/pmc_sq_counters.json $(find $O/pmc_sq1 $O/pmc_sq2 -name "*counter_collection.csv") > $O/pmc_sq_counters.txt 2>&1
unset FPLX_SIDE_STREAM
# in-kernel clock and per-phase cycles of the dominant kernel (stand-alone harness, random data)
# two-stream timeline of the step as shipped (both streams on)
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -- $PROG > $O/trace2.log 2>&1
python tools/timeline_gaps.py $(find $O/trace2 -name "*kernel_trace.csv" | head -1) > $O/two_stream_timeline.txt 2>&1
( timeout 300 python tools/wgrad_bench.py; timeout 300 python tools/r03_ab.py; timeout 300 python tools/edge_bench.py ) > $O/kernel_ab.txt 2>&1
rm -rf $O/trace $O/trace2 $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2
head -c 1500 $O/bench.json; echo; cat $O/secondary.txt; head -12 $O/trace_summary.txt; head -8 $O/pmc_hbm_traffic.txt
