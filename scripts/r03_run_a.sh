#!/bin/bash
# round 3, first GPU call: the MPC kernel (parity tests, timing), then the whole GPU tier
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_mpc.py -m gpu -q -x ) > gpurun_out/r03_a_pytest_mpc.txt 2>&1
tail -5 gpurun_out/r03_a_pytest_mpc.txt
timeout 600 python scripts/mpc_bench.py > gpurun_out/r03_a_mpc_bench.txt 2>&1
cat gpurun_out/r03_a_mpc_bench.txt | grep -v amdgpu | tail -12
( time timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_mpc.py ) > gpurun_out/r03_a_pytest_gpu.txt 2>&1
tail -5 gpurun_out/r03_a_pytest_gpu.txt
