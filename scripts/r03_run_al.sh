#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_mpc.py -m gpu -q -k "generic_mode" ) > gpurun_out/r03_al_pytest.txt 2>&1
tail -30 gpurun_out/r03_al_pytest.txt
