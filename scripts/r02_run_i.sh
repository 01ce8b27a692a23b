#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q ) > gpurun_out/i_pytest.log 2>&1
tail -6 gpurun_out/i_pytest.log
timeout 900 bash scripts/r02_profile.sh a
timeout 600 python bench.py --steps 200 --warmup 20 > gpurun_out/r02_a_bench.json.log 2>&1; tail -c 2500 gpurun_out/r02_a_bench.json.log
