#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=tools/micro/bin
( for r in 1 2; do for v in base v1 ""; do for a in "32 32" "64 64 2 40 80 80"; do echo -n "[$v] "; $B/march_bench${v:+_$v} $a | grep -v checksum; done; done; done ) > gpurun_out/march_ab.txt 2>&1
cat gpurun_out/march_ab.txt
