#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "# scripts/bench_large_networks.py: ANM_MESH_FUSED_LEVELS=1 (round-3 experiment)"; ANM_MESH_FUSED_LEVELS=1 timeout 600 python scripts/bench_large_networks.py 2>&1 | grep -v amdgpu.ids | cut -c1-190
  echo "# the default schedule (a products step + a sums step per level)"; timeout 600 python scripts/bench_large_networks.py 2>&1 | grep -v amdgpu.ids | cut -c1-190 ) > gpurun_out/r03_o_large_fused_levels.txt
cat gpurun_out/r03_o_large_fused_levels.txt
