#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net_parity.py tests/test_gpu_fullsize.py tests/test_gpu_25d.py -m gpu -q -x ) > gpurun_out/tests_fused.txt 2>&1
tail -n 12 gpurun_out/tests_fused.txt
for v in 0 1 0 1; do FPLX_FUSED_POOL=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSED_POOL=$v ms_per_step', d['ms_per_step'], 'loss', d['final_loss'])"; done
