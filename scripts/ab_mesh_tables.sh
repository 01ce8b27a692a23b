#!/bin/bash
# usage (GPU box): bash scripts/ab_mesh_tables.sh  -- general lane-group kernel, the lists and the step program staged in LDS
# or read from global memory (ANM_MESH_TABLES=lds|global; default: mesh::launch_of), same box
cd $GRAFT_REPO_ROOT
run() { python scripts/mesh_occupancy_bench.py "$@" 2>&1 | grep "us per launch"; }
for i in 1 2; do
  for t in lds global auto; do
    if [ $t = auto ]; then unset ANM_MESH_TABLES; else export ANM_MESH_TABLES=$t; fi
    echo "== tables $t"; run mesh20 mesh30 mesh50 mesh64
  done
done
