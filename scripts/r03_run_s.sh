#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/r03_profile.sh s mpc > gpurun_out/r03_s_profile.log 2>&1
tail -3 gpurun_out/r03_s_profile.log
grep -A9 derived gpurun_out/r03_s_pmc_mpc_N1.txt
head -5 gpurun_out/r03_s_mpc_N1_kernel_trace.txt | cut -c1-200
