#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scripts/bench_large_networks.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_m_large_networks.txt
cat gpurun_out/r03_m_large_networks.txt
