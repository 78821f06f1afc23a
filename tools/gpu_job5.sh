#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python tools/dice_proxy.py --base 32 --shape 64,128,128 --iters 300 --oracle-iters 0 --seeds 3 ) > gpurun_out/dice_proxy32.txt 2>&1
tail -n 12 gpurun_out/dice_proxy32.txt
