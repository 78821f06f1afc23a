#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_loss_filter_parity.py tests/test_gpu_fullsize.py -m gpu -q -x ) > gpurun_out/tests_ddp.txt 2>&1
tail -n 40 gpurun_out/tests_ddp.txt
