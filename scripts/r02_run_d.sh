#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -q -x ) > gpurun_out/d_pytest.log 2>&1
tail -5 gpurun_out/d_pytest.log
timeout 300 python scripts/handoff_sweep.py 65536 16384 > gpurun_out/d_sweep.log 2>&1
cat gpurun_out/d_sweep.log
