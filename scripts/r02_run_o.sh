#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q -k "mesh" ) 2>&1 | tail -4
timeout 600 python scripts/bench_mesh.py 2>&1 | grep -v amdgpu
