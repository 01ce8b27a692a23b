#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q ) > gpurun_out/k_pytest.log 2>&1
tail -8 gpurun_out/k_pytest.log
timeout 300 python scripts/handoff_sweep.py 65536 > gpurun_out/k_sweep.log 2>&1; cat gpurun_out/k_sweep.log
timeout 300 python scripts/obs_modes_bench.py 65536 > gpurun_out/k_obs_modes.log 2>&1; cat gpurun_out/k_obs_modes.log
