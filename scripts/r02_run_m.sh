#!/bin/bash
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -q -x -k "mesh" ) 2>&1 | tail -5
timeout 300 python scripts/mesh_caps.py 2>&1 | grep -v amdgpu
