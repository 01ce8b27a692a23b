#!/bin/bash
# usage (GPU box): bash scripts/ab_mesh_occupancy.sh  -- the general lane-group kernel budgeted for two and for three
# wavefronts per SIMD (ANM_MESH_SIMD_WAVES; default: mesh::simd_waves picks by the LDS of the plan)
cd $GRAFT_REPO_ROOT
run() { python scripts/mesh_occupancy_bench.py "$@" 2>&1 | grep "us per launch"; }
for i in 1 2; do
  for sw in 2 3; do
    export ANM_MESH_SIMD_WAVES=$sw
    run mesh12 mesh20 mesh30
    ANM_IMPL=mesh run anm6
  done
  unset ANM_MESH_SIMD_WAVES
  run mesh12 mesh20 mesh30
done
