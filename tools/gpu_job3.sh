#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/tests_all.txt 2>&1
tail -n 12 gpurun_out/tests_all.txt
