#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/race25.py 20 2>&1 | grep -v "^   \|join only\|with taps" > gpurun_out/race25_fixed.txt
cat gpurun_out/race25_fixed.txt
