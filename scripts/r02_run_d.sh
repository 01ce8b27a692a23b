#!/bin/bash
# small refresh after stage c: the configurations the early-exit change touches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 python scripts/bench_case30_quick.py; timeout 200 python scripts/classes_bench.py ) 2>&1 | grep -v amdgpu > gpurun_out/r02_d_case30.txt
timeout 300 python bench.py --steps 200 --warmup 20 > gpurun_out/r02_d_bench.json.log 2>&1
cat gpurun_out/r02_d_case30.txt; tail -c 300 gpurun_out/r02_d_bench.json.log
