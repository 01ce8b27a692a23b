#!/bin/bash
# usage (GPU box): bash scripts/ab_headline.sh [rounds]  -- same-box A/B of the headline kernel: the tree against the copy
# of an earlier commit under _ab/base (git worktree add -f _ab/base <commit>; build its ANM6 library there first)
R=$GRAFT_REPO_ROOT
n=${1:-3}
for i in $(seq $n); do
  for w in base new; do
    d=$R; [ $w = base ] && d=$R/_ab/base
    r=$(cd $d && python bench.py --headline-only --no-cpu-baseline --steps 300 --warmup 30 2>&1 | grep -o "\"ms_per_step\": [0-9.]*\|kernel_ms\": [0-9.]*" | tr '\n' ' ')
    echo "$w $r"
  done
done
