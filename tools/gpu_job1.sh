#!/bin/bash
# round-2 job 1: race locator, new parity tests, bench, SQ counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/race25.py 30 > gpurun_out/race25.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_net_parity.py tests/test_gpu_fullsize.py "tests/test_gpu_25d.py" tests/test_gpu_loss_filter_parity.py -m gpu -x -q > gpurun_out/tests_new.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
timeout 120 python tools/cfg5_bench.py > gpurun_out/cfg5.txt 2>&1
export FPLX_SIDE_STREAM=0
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_sq1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/pmc_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/pmc_sq2.log 2>&1
find gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 -name "*counter_collection.csv" | head
python tools/pmc_sq_summary.py gpurun_out/pmc_sq.json $(find gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 -name "*counter_collection.csv") > gpurun_out/pmc_sq.txt 2>&1
# keep the merged-back payload small
find gpurun_out -name "*.csv" -size +20M -delete
tail -5 gpurun_out/race25.txt gpurun_out/tests_new.txt gpurun_out/cfg5.txt
head -c 600 gpurun_out/bench.json
