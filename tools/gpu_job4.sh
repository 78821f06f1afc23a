#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_25d.py tests/test_gpu_net_parity.py tests/test_gpu_loss_filter_parity.py -m gpu -x -q 2>&1 | tail -n 3
for p in 1 0 1 0; do
FPLX_OVERLAP_PACKS=$p timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('overlap_packs $p', d['value'], d['ms_per_step'])"
done
