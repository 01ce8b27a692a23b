#!/bin/bash
cd $GRAFT_REPO_ROOT
( time timeout 1700 python -m pytest tests -m gpu -q -x ) 2>&1 | tail -6
timeout 300 python scripts/mesh_caps.py 2>&1 | grep -v amdgpu
timeout 300 python scripts/bench_mesh.py 2>&1 | grep -v amdgpu
