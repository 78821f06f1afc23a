#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_25d.py tests/test_gpu_net_parity.py -m gpu -x -q 2>&1 | tail -n 3
for p in 2 1 2 1; do
  FPLX_WG_COT=$p timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('cot $p', d['value'], d['ms_per_step'])
"
done
