#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q ) > gpurun_out/e_pytest.log 2>&1
tail -25 gpurun_out/e_pytest.log
timeout 300 python scripts/obs_modes_bench.py 65536 > gpurun_out/e_obs_modes.log 2>&1
cat gpurun_out/e_obs_modes.log
