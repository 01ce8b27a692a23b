#!/bin/bash
# stage e: after the LDS hand-overs of the specialised radial loop and the 128-bit LDS accesses of the general kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02_e_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r02_e_pytest_gpu.txt
( timeout 300 python scripts/bench_case30_quick.py; ANM_RADIAL_GENERIC=1 timeout 300 python scripts/bench_case30_quick.py; timeout 200 python scripts/classes_bench.py ) 2>&1 | grep -v amdgpu > gpurun_out/r02_e_case30.txt
( timeout 300 python scripts/bench_mesh.py; timeout 200 python scripts/mesh_caps.py ) 2>&1 | grep -v amdgpu > gpurun_out/r02_e_mesh_family.txt
timeout 300 python bench.py --steps 200 --warmup 20 > gpurun_out/r02_e_bench.json.log 2>&1
cat gpurun_out/r02_e_case30.txt gpurun_out/r02_e_mesh_family.txt; grep -o '"ms_per_step": [0-9.]*\|"kernel_ms[a-z_]*": [0-9.]*\|"us_per_launch": [0-9.]*' gpurun_out/r02_e_bench.json.log
bash scripts/r02_pmc_mesh.sh 100 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_r02_e; mkdir -p $out
C30="python $R/scripts/bench_case30_quick.py"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_c30/p1 -- $C30 > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_c30/p2 -- $C30 > $out/p2.log 2>&1
ANM_PMC_KERNEL=k_radial python $R/scripts/pmc_summary.py $out/pmc_c30 "config 4: case30 radial, 16384 envs, Simulator.transition with the electrical-state dump, caps 100 and 20 mixed; LDS hand-overs" > $R/gpurun_out/r02_e_pmc_case30.txt
tail -12 $R/gpurun_out/r02_e_pmc_case30.txt
