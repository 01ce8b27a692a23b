#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q ) > gpurun_out/h_pytest.log 2>&1
tail -12 gpurun_out/h_pytest.log
timeout 300 python scripts/handoff_sweep.py 65536 > gpurun_out/h_sweep.log 2>&1; cat gpurun_out/h_sweep.log
