#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/r03_profile.sh x mpc > gpurun_out/r03_x_profile.log 2>&1
tail -3 gpurun_out/r03_x_profile.log
grep -A5 derived gpurun_out/r03_x_pmc_mpc_N10.txt
sed -n 3,5p gpurun_out/r03_x_mpc_N10_kernel_trace.txt | cut -c1-200
sed -n 3,5p gpurun_out/r03_x_mpc_N1_kernel_trace.txt | cut -c1-200
