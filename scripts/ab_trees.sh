#!/bin/bash
# usage (GPU box): bash scripts/ab_trees.sh <rounds> "<tree> <tree> ..." <workload> ...   -- trees: "." or _ab/<name>
R=$GRAFT_REPO_ROOT
n=$1; trees=$2; shift; shift
for i in $(seq $n); do
  for t in $trees; do
    ( cd $R/$t && python $R/scripts/ab_trees.py "$@" 2>&1 | grep " us " | sed "s|^|$t  |" )
  done
done
