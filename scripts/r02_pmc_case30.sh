#!/bin/bash
# PMC passes (kernel trace only) of k_radial on config 4 -> gpurun_out/r02_<stage>_pmc_case30.txt
stage=${1:-g}
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_r02_${stage}_c30; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
C30="python $R/scripts/bench_case30_quick.py"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/p1 -- $C30 > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -- $C30 > $out/p2.log 2>&1
ANM_PMC_KERNEL=k_radial python $R/scripts/pmc_summary.py $out "config 4: case30 radial, 16384 envs, Simulator.transition with the electrical-state dump, caps 100 and 20 mixed; LDS hand-overs, child slots three at a time (159 VGPRs)" > $R/gpurun_out/r02_${stage}_pmc_case30.txt
cat $R/gpurun_out/r02_${stage}_pmc_case30.txt
