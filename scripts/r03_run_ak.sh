#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "# scripts/soak_families.py on the final tree"; timeout 900 python scripts/soak_families.py 2>&1 | grep -v amdgpu.ids | tail -8
  echo "# scripts/soak_two_launch.py (262 144 environments, 600 steps) on the final tree"; timeout 900 python scripts/soak_two_launch.py 2>&1 | grep -v amdgpu.ids | tail -4 ) > gpurun_out/r03_ak_soaks.txt
cat gpurun_out/r03_ak_soaks.txt
