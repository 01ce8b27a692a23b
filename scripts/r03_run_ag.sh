#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/r03_profile.sh ag head > gpurun_out/r03_ag_profile.log 2>&1
tail -3 gpurun_out/r03_ag_profile.log
cat gpurun_out/r03_ag_pmc_headline.txt | head -45
head -6 gpurun_out/r03_ag_bench_headline_kernel_trace.txt | cut -c1-200
