#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=tools/micro/bin
( for r in 1 2; do
  FPLX_MARCH32_V2=0 MB_DUMP=/tmp/v1.bin $B/march_bench 32 32 | grep -v checksum
  FPLX_MARCH32_V2=1 MB_DUMP=/tmp/v2.bin $B/march_bench 32 32 | grep -v checksum
  done
  python tools/micro/cmp_bf16.py /tmp/v1.bin /tmp/v2.bin
  FPLX_MARCH32_V2=1 $B/march_bench_stamp 32 32 | grep -v checksum
) > gpurun_out/march_v2.txt 2>&1
cat gpurun_out/march_v2.txt
