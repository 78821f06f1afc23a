#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -n 2
for p in 1 0 1 0; do
  FPLX_V3_ASWZ=$p timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('aswz $p', d['value'], d['ms_per_step'], [ (k['kernel'][:60], k['avg_ms']) for k in d['kernels'][:3]])
"
done
