#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "brick" 2>&1 | tail -n 2
timeout 300 python tools/brick_check.py l1 2>&1 | grep "time"
timeout 300 python tools/brick_check.py 2>&1 | grep "time"
