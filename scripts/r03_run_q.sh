#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/r03_profile.sh q mpc wg > gpurun_out/r03_q_profile.log 2>&1
tail -5 gpurun_out/r03_q_profile.log
head -50 gpurun_out/r03_q_pmc_mesh200.txt
