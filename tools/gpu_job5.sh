#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2500 python tools/dice_proxy.py --iters 400 --base 16 --oracle-iters 3 --seeds 6 ) > gpurun_out/dice_proxy.txt 2>&1
cat gpurun_out/dice_proxy.txt
