#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03_k_pytest_gpu.txt 2>&1
tail -6 gpurun_out/r03_k_pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r03_k_bench.json.log 2>&1
tail -c 3000 gpurun_out/r03_k_bench.json.log
timeout 300 python scripts/mpc_bench.py > gpurun_out/r03_k_mpc_bench.txt 2>&1
tail -12 gpurun_out/r03_k_mpc_bench.txt
