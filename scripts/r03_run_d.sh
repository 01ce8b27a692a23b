#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03_d_pytest_gpu.txt 2>&1
tail -6 gpurun_out/r03_d_pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r03_d_bench.json.log 2>&1
tail -c 3000 gpurun_out/r03_d_bench.json.log
bash scripts/r03_profile.sh d head mpc thr > gpurun_out/r03_d_profile.log 2>&1
tail -5 gpurun_out/r03_d_profile.log
