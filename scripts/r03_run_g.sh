#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tag in "" w3 w4; do
  for E in 65536 524288 1048576; do
    if [ -z "$tag" ]; then python scripts/throughput_workload.py $E 60 2>&1 | grep "^E " | sed "s/^/default  /";
    else ANM_BUILD_TAG=$tag ANM_EXTRA_HIPCC_FLAGS="-DANM_ROWS_WAVES=${tag#w}" python scripts/throughput_workload.py $E 60 2>&1 | grep "^E " | sed "s/^/$tag       /"; fi
  done
done | tee gpurun_out/r03_g_rows_waves.txt
