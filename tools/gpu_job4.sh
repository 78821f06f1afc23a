#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( for r in 1 2; do for v in 0 1 3 4; do FPLX_MARCH32_V2=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V2=$v ms_per_step', d['ms_per_step'], [ (k['kernel'][14:60],k['avg_ms']) for k in d['kernels'] if ('32, 32' in k['kernel'] or '32, 64' in k['kernel']) and 'fwd' in k['kernel']])"; done; done
) > gpurun_out/march_v3_bench.txt 2>&1
cat gpurun_out/march_v3_bench.txt
