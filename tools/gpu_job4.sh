#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "outconv or out_conv or edge" 2>&1 | tail -n 2
export TMPDIR=/tmp
mkdir -p gpurun_out/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/tl/trace.log 2>&1
f=$(find gpurun_out/tl/trace -name "*kernel_trace.csv" | head -1)
python tools/timeline_gaps.py $f 4 | head -3
python tools/trace_summary.py $f gpurun_out/tl/by_shape.csv > /dev/null
grep -i "outconv\|stem\|march32v2" gpurun_out/tl/by_shape.csv
rm -rf gpurun_out/tl/trace
