#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/rccl_single_rank2.txt
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('  ms_per_step', d['ms_per_step'])"; }
B="bench.py --gpus 1 --steps 20 --warmup 5 --no-kernel-timing --no-cpu-baseline"
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 FPLX_DDP_FORCE=1"
echo "no process group:" > $O; python $B 2>/dev/null | ms >> $O
echo "no process group, FPLX_SIDE_STREAM=0 (no overlap at all):" >> $O; FPLX_SIDE_STREAM=0 python $B 2>/dev/null | ms >> $O
echo "nccl group + collectives:" >> $O; env $D MASTER_PORT=29571 python $B 2>/dev/null | ms >> $O
echo "nccl group + collectives, GPU_MAX_HW_QUEUES=8:" >> $O; env $D MASTER_PORT=29572 GPU_MAX_HW_QUEUES=8 python $B 2>/dev/null | ms >> $O
echo "nccl group + collectives, GPU_MAX_HW_QUEUES=16:" >> $O; env $D MASTER_PORT=29573 GPU_MAX_HW_QUEUES=16 python $B 2>/dev/null | ms >> $O
echo "no process group, GPU_MAX_HW_QUEUES=8:" >> $O; GPU_MAX_HW_QUEUES=8 python $B 2>/dev/null | ms >> $O
cat $O
