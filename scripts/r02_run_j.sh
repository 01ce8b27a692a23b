#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q -k "mesh" ) > gpurun_out/j_pytest.log 2>&1
tail -25 gpurun_out/j_pytest.log
timeout 600 python scripts/bench_mesh.py > gpurun_out/j_mesh.log 2>&1; cat gpurun_out/j_mesh.log
