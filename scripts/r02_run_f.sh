#!/bin/bash
# stage f (end of round): full GPU suite, bench line, mesh family after the dense two-bus tail step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02_f_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r02_f_pytest_gpu.txt
( timeout 300 python scripts/bench_mesh.py; timeout 200 python scripts/mesh_caps.py ) 2>&1 | grep -v amdgpu > gpurun_out/r02_f_mesh_family.txt
timeout 300 python bench.py --steps 200 --warmup 20 > gpurun_out/r02_f_bench.json.log 2>&1
cat gpurun_out/r02_f_mesh_family.txt; grep -o '"ms_per_step": [0-9.]*\|"kernel_ms[a-z_]*": [0-9.]*\|"us_per_launch": [0-9.]*' gpurun_out/r02_f_bench.json.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02_f -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --headline-only --steps 200 --warmup 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_r02_f.log 2>&1
cd $GRAFT_REPO_ROOT
( python scripts/prof_summary.py gpurun_out/prof_r02_f "command: bench.py --no-cpu-baseline --headline-only --steps 200 --warmup 20"; echo "# bench.py line of the same run:"; grep '^{' gpurun_out/prof_r02_f.log ) > gpurun_out/r02_f_bench_headline_kernel_trace.txt
head -5 gpurun_out/r02_f_bench_headline_kernel_trace.txt | cut -c1-160
