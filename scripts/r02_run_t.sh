#!/bin/bash
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_phase" ) 2>&1 | tail -3
timeout 400 python scripts/handoff_sweep.py 65536 262144 524288 1048576 2>&1 | grep -v amdgpu
