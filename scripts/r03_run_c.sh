#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/mpc_trace_dump.py cuda:0 > gpurun_out/r03_c_trace_default.txt 2>&1
cut -c1-230 gpurun_out/r03_c_trace_default.txt | grep -v "nan        nan" | head -150
