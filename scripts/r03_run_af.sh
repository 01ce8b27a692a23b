#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for cfg in "4 12" "6 12"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats -d /tmp/prof_af_$1 -- python $R/scripts/throughput_workload.py 1048576 40 $1 $2 > /tmp/prof_af_$1.log 2>&1
  ( python $R/scripts/prof_summary.py /tmp/prof_af_$1 "command: scripts/throughput_workload.py 1048576 40 $1 $2 (hand-over to the straggler launch after $1 iterations, second level at $2)" | head -8 | cut -c1-170; grep "^E " /tmp/prof_af_$1.log )
done > $R/gpurun_out/r03_af_handover_4_vs_6.txt
cat $R/gpurun_out/r03_af_handover_4_vs_6.txt
