#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_net_parity.py tests/test_gpu_kernels.py -m gpu -x -q --durations=5 -k "end_to_end or brick" 2>&1 | tail -n 12
cat gpurun_out/parity_bf16_b2.txt | head -8
