#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "# scripts/bench_mesh.py: ANM_MESH_FUSED_LEVELS=1 (round-3 experiment: product and subtraction of a level in ONE step)"; ANM_MESH_FUSED_LEVELS=1 timeout 600 python scripts/bench_mesh.py 2>&1 | grep -v amdgpu.ids
  echo "# the default schedule (a products step + a sums step per level)"; timeout 600 python scripts/bench_mesh.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r03_n_mesh_fused_levels.txt
cat gpurun_out/r03_n_mesh_fused_levels.txt
( time timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -q -k "mesh" ) > gpurun_out/r03_n_pytest_mesh.txt 2>&1
tail -5 gpurun_out/r03_n_pytest_mesh.txt
