#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_mpc.py -m gpu -q ) > gpurun_out/r03_ab_pytest_mpc.txt 2>&1
tail -4 gpurun_out/r03_ab_pytest_mpc.txt
