#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -q -k "larger_than" ) > gpurun_out/r03_l_pytest_large.txt 2>&1
tail -30 gpurun_out/r03_l_pytest_large.txt
