#!/bin/bash
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_headline.py -m gpu -q -x -k "headline_config or sharded or fused or two_devices" ) 2>&1 | tail -2
timeout 200 python scripts/bench_case30_quick.py 2>&1 | grep -v amdgpu
timeout 200 python bench.py --headline-only --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms[a-z_]*": [0-9.]*'
