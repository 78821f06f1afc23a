#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 FPLX_DDP_FORCE=1 FPLX_DDP_LAZY=1
rm -rf gpurun_out/prof_rccl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rccl -- python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-kernel-timing --no-cpu-baseline > gpurun_out/prof_rccl.log 2>&1
f=$(find gpurun_out/prof_rccl -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -c1-150
grep -i "nccl\|rccl" $f | cut -c1-200
find gpurun_out/prof_rccl -name "*kernel_trace.csv" -size +30M -delete
