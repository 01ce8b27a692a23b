#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "# scripts/straggler_levels_sweep.py after the coalesced scatter"; timeout 900 python scripts/straggler_levels_sweep.py 2>&1 | grep "^E " ) | tee gpurun_out/r03_ae_straggler_levels.txt
