#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 3
echo "--- bench"
for i in 1 2; do
timeout 600 python bench.py 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['avg_ms'])"
done
FPLX_BRICK=0 timeout 600 python bench.py 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('brick=0', {k:d[k] for k in ('value','ms_per_step')}, d['roofline']['avg_ms'])"
