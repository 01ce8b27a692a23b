#!/bin/bash
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q -x -k "radial or case30 or sampler or golden" ) 2>&1 | tail -3
timeout 200 python scripts/bench_case30_quick.py 2>&1 | grep -v amdgpu
