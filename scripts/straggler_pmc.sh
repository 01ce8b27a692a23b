#!/bin/bash
# usage: scripts/straggler_pmc.sh <tag> [num_envs] [after]   (on the GPU box) -- PMC passes for k_step_stragglers
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/p1 -- python $GRAFT_REPO_ROOT/scripts/straggler_workload.py "$@" > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -- python $GRAFT_REPO_ROOT/scripts/straggler_workload.py "$@" > $out/p2.log 2>&1
