#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/brick_check.py check 2>&1 | tail -n 20
echo "--- tile kernel"
FPLX_BRICK=0 timeout 300 python tools/brick_check.py 2>&1 | tail -n 8
