#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/test_gpu_net_parity.py -m gpu -q -k march ) > gpurun_out/tests_all.txt 2>&1
tail -n 15 gpurun_out/tests_all.txt
cat gpurun_out/parity_bf16_m1.txt gpurun_out/parity_bf16_m4.txt 2>/dev/null | head -60
