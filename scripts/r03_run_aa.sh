#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/mpc_case30.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_aa_mpc_case30.txt
cat gpurun_out/r03_aa_mpc_case30.txt
( time timeout 900 python -m pytest tests/test_mpc.py -m gpu -q ) > gpurun_out/r03_aa_pytest_mpc.txt 2>&1
tail -4 gpurun_out/r03_aa_pytest_mpc.txt
