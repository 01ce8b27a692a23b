#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "# scripts/bench_mesh.py: operands of a step unpacked in the shadow of the previous step's stores"; timeout 600 python scripts/bench_mesh.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r03_ah_mesh.txt
cat gpurun_out/r03_ah_mesh.txt
( time timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -q -k "mesh" ) > gpurun_out/r03_ah_pytest_mesh.txt 2>&1
tail -3 gpurun_out/r03_ah_pytest_mesh.txt
