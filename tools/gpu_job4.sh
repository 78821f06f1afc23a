#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'])"
