#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/test_gpu_net_parity.py tests/test_gpu_checkpoint.py -m gpu -q -x ) > gpurun_out/tests_bl.txt 2>&1
tail -n 40 gpurun_out/tests_bl.txt
