#!/bin/bash
# PMC passes (kernel trace only) of the general lane-group kernel on mesh30, 16384 environments; $1 = Newton cap
cap=${1:-8}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_mesh_$cap
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/mesh_one.py $cap"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/p1 -- $CMD > $out.p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -- $CMD > $out.p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES --output-format csv -d $out/p3 -- $CMD > $out.p3.log 2>&1
ANM_PMC_KERNEL=k_mesh python $R/scripts/pmc_summary.py $out "mesh30, 16384 envs, cap $cap" > $R/gpurun_out/r02_pmc_mesh_$cap.txt
cat $R/gpurun_out/r02_pmc_mesh_$cap.txt
tail -3 $out.p3.log
