#!/bin/bash
# end-of-round evidence: everything profiles/r02_<stage>_* is made of (stage = $1, default c)
ST=${1:-c}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q ) > gpurun_out/r02_${ST}_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r02_${ST}_pytest_gpu.txt
timeout 600 python bench.py --steps 200 --warmup 20 > gpurun_out/r02_${ST}_bench.json.log 2>&1; tail -c 600 gpurun_out/r02_${ST}_bench.json.log
timeout 400 python scripts/handoff_sweep.py 65536 4096 262144 1048576 2>&1 | grep -v amdgpu > gpurun_out/r02_${ST}_handoff_sweep.txt
timeout 300 python scripts/obs_modes_bench.py 65536 2>&1 | grep -v amdgpu > gpurun_out/r02_${ST}_obs_modes.txt
timeout 300 python scripts/classes_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r02_${ST}_parameter_classes.txt
( timeout 300 python scripts/bench_case30_quick.py; ANM_RADIAL_GENERIC=1 timeout 300 python scripts/bench_case30_quick.py ) 2>&1 | grep -v amdgpu > gpurun_out/r02_${ST}_case30_specialised_vs_generic.txt
( timeout 300 python scripts/bench_mesh.py; timeout 200 python scripts/mesh_caps.py ) 2>&1 | grep -v amdgpu > gpurun_out/r02_${ST}_mesh_family.txt
timeout 600 python scripts/mpc_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r02_${ST}_mpc.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02_${ST} -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --headline-only --steps 200 --warmup 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_r02_${ST}.log 2>&1
cd $GRAFT_REPO_ROOT
( python scripts/prof_summary.py gpurun_out/prof_r02_${ST} "command: bench.py --no-cpu-baseline --headline-only --steps 200 --warmup 20"; echo "# bench.py line of the same run:"; grep '^{' gpurun_out/prof_r02_${ST}.log ) > gpurun_out/r02_${ST}_bench_headline_kernel_trace.txt
head -5 gpurun_out/r02_${ST}_bench_headline_kernel_trace.txt | cut -c1-160
