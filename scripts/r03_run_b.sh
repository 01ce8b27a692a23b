#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_mpc.py -m gpu -q ) > gpurun_out/r03_b_pytest_mpc.txt 2>&1
tail -8 gpurun_out/r03_b_pytest_mpc.txt
timeout 600 python scripts/mpc_bench.py > gpurun_out/r03_b_mpc_bench.txt 2>&1
cat gpurun_out/r03_b_mpc_bench.txt | grep -v amdgpu | tail -14
