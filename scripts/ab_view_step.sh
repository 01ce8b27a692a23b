#!/bin/bash
# usage (GPU box): bash scripts/ab_view_step.sh  -- k_step_view budgeted for one (288 registers) or two (256 + 80 B scratch)
# wavefronts per SIMD: tags v1 / v2 = -DANM_VIEW_WAVES=1|2, built beforehand
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for tag in v1 v2; do
    ANM_BUILD_TAG=$tag ANM_EXTRA_HIPCC_FLAGS="-DANM_VIEW_WAVES=${tag#v}" python scripts/view_step_bench.py 16384 131072 524288 2>&1 | grep "view step\|Error\|error" | head -5
  done
done
