#!/bin/bash
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -q -k "headline_config or sharded or fused" ) 2>&1 | tail -3
timeout 300 python scripts/handoff_sweep.py 65536 2>&1 | grep -v amdgpu
