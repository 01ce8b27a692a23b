#!/bin/bash
# usage (GPU box): bash scripts/ab_variants.sh "<tag>=<hipcc flags>" ...   -- headline kernel + 524 288-environment step of
# side-by-side tuning builds of the tree (ANM_BUILD_TAG / ANM_EXTRA_HIPCC_FLAGS), same box; "base=" is the tree as it is
R=$GRAFT_REPO_ROOT
cd $R
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  if [ "$tag" = base ]; then unset ANM_BUILD_TAG ANM_EXTRA_HIPCC_FLAGS; else export ANM_BUILD_TAG=$tag ANM_EXTRA_HIPCC_FLAGS="$flags"; fi
  for i in 1 2; do
    h=$(python bench.py --headline-only --no-cpu-baseline --steps 300 --warmup 30 2>&1 | grep -o "kernel_ms\": [0-9.]*")
    t=$(python scripts/throughput_workload.py 524288 100 2>&1 | tail -1 | grep -o "^E [0-9]*: [0-9.]* us")
    echo "$tag  headline $h   $t"
  done
done
