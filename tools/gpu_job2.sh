#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/race25.py 3 > gpurun_out/race25b.txt 2>&1
grep -v "^   " gpurun_out/race25b.txt
