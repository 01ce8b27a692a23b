#!/bin/bash
# stage g (end of round): after the register diet of the specialised radial loop
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02_g_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r02_g_pytest_gpu.txt
( timeout 300 python scripts/bench_case30_quick.py; timeout 200 python scripts/classes_bench.py ) 2>&1 | grep -v amdgpu > gpurun_out/r02_g_case30.txt
timeout 300 python bench.py --steps 200 --warmup 20 > gpurun_out/r02_g_bench.json.log 2>&1
cat gpurun_out/r02_g_case30.txt; grep -o '"ms_per_step": [0-9.]*\|"kernel_ms[a-z_]*": [0-9.]*\|"us_per_launch": [0-9.]*' gpurun_out/r02_g_bench.json.log
