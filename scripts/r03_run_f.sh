#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x -k "two_phase or two_launch or graph or stragglers or headline or sharded or soak" ) > gpurun_out/r03_f_pytest.txt 2>&1
tail -5 gpurun_out/r03_f_pytest.txt
for E in 131072 262144 524288 1048576; do python scripts/throughput_workload.py $E 60 2>&1 | grep "^E "; done | tee gpurun_out/r03_f_throughput.txt
