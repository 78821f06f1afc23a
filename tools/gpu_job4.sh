#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -n 3
echo "--- L1/L2 timing"
timeout 300 python tools/brick_check.py l1 2>&1 | tail -n 6
timeout 300 python tools/brick_check.py 2>&1 | tail -n 6
echo "--- bench"
for i in 1 2; do
timeout 600 python bench.py 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['avg_ms'])"
done
