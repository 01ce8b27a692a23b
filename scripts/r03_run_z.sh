#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "# scripts/soak_two_launch.py 1048576 300: the automatic step at 1 M environments (two straggler levels) against the in-wave path, bit for bit at checkpoints"
  timeout 900 python scripts/soak_two_launch.py 1048576 300 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r03_z_soak_two_levels.txt
cat gpurun_out/r03_z_soak_two_levels.txt
