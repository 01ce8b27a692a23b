#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/r03_profile.sh d head mpc thr > gpurun_out/r03_d_profile.log 2>&1
tail -5 gpurun_out/r03_d_profile.log
timeout 600 python bench.py > gpurun_out/r03_d_bench.json.log 2>&1
tail -c 600 gpurun_out/r03_d_bench.json.log
