#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x -k "two_phase or two_launch or two_level or graph or stragglers or sharded" ) > gpurun_out/r03_ad_pytest.txt 2>&1
tail -4 gpurun_out/r03_ad_pytest.txt
( echo "# scripts/throughput_workload.py after the coalesced scatter (16 lanes per record), automatic straggler policy"
  for E in 131072 262144 524288 1048576; do python scripts/throughput_workload.py $E 60 2>&1 | grep "^E "; done ) | tee gpurun_out/r03_ad_throughput.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ad -- python $GRAFT_REPO_ROOT/scripts/throughput_workload.py 1048576 40 > /tmp/prof_ad.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py /tmp/prof_ad "command: scripts/throughput_workload.py 1048576 40 (coalesced scatter)" > $GRAFT_REPO_ROOT/gpurun_out/r03_ad_thr_1048576_kernel_trace.txt
head -8 $GRAFT_REPO_ROOT/gpurun_out/r03_ad_thr_1048576_kernel_trace.txt | cut -c1-180
