#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export ANM_BUILD_TAG=phases ANM_EXTRA_HIPCC_FLAGS=-DANM_PHASE_TIMING
timeout 900 python scripts/phase_times_radial.py 20 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r03_aj_radial_phases.txt
