#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_net_parity.py tests/test_gpu_loss_filter_parity.py -m gpu -x -q 2>&1 | tail -n 6
timeout 300 python tools/fpl_infer_bench.py 2>&1 | tail -n 4
