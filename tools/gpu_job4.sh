#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net_parity.py tests/test_gpu_25d.py -m gpu -x -q 2>&1 | tail -n 3
export TMPDIR=/tmp FPLX_SIDE_STREAM=0
mkdir -p gpurun_out/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/tl/trace.log 2>&1
f=$(find gpurun_out/tl/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $f gpurun_out/tl/by_shape.csv > /dev/null
grep -i "outconv\|stem" gpurun_out/tl/by_shape.csv
rm -rf gpurun_out/tl/trace
unset FPLX_SIDE_STREAM
for p in 1 0 1 0; do
FPLX_OUTCONV_DGRAD_MFMA=$p timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('dgrad_mfma $p', d['value'], d['ms_per_step'], d.get('final_loss'))"
done
