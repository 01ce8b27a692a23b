#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 4 5; do
  echo "== ANM_RADIAL_WAVES=$w"
  ANM_BUILD_TAG=rw$w ANM_EXTRA_HIPCC_FLAGS="-DANM_RADIAL_WAVES=$w" timeout 300 python scripts/bench_case30_quick.py 2>&1 | grep -v amdgpu
done
echo "== default"; timeout 300 python scripts/bench_case30_quick.py 2>&1 | grep -v amdgpu
