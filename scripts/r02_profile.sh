#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/r02_profile.sh <stage>
# kernel trace + PMC passes (one counter group per pass, kernel-trace only) of the headline configuration and of
# BASELINE config 4 (case30); summaries land in gpurun_out/r02_<stage>_*.txt ready to be copied to profiles/.
stage=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_r02_$stage
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
HEAD="python $R/bench.py --no-cpu-baseline --headline-only --steps 200 --warmup 20"
C30="python $R/scripts/bench_case30_quick.py"
rocprofv3 --kernel-trace --stats -d $out/trace -- $HEAD > $out/trace.log 2>&1
( python $R/scripts/prof_summary.py $out/trace "command: bench.py --no-cpu-baseline --headline-only --steps 200 --warmup 20"; echo "# bench.py line of the same run:"; grep '^{' $out/trace.log ) > $R/gpurun_out/r02_${stage}_bench_headline_kernel_trace.txt
pmc() {  # pmc <dir> <cmd...> : four passes
  d=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $d/p1 -- "$@" > $d.p1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $d/p2 -- "$@" > $d.p2.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d/p3 -- "$@" > $d.p3.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $d/p4 -- "$@" > $d.p4.log 2>&1
}
pmc $out/pmc_head $HEAD
ANM_PMC_KERNEL=k_step_rows python $R/scripts/pmc_summary.py $out/pmc_head "headline: ANM6Easy 65536 envs, tol 1e-6, cap 100, autoreset, in-wave lane-group hand-over" > $R/gpurun_out/r02_${stage}_pmc_headline.txt
pmc $out/pmc_c30 $C30
ANM_PMC_KERNEL=k_radial python $R/scripts/pmc_summary.py $out/pmc_c30 "config 4: case30 radial, 16384 envs, Simulator.transition with the electrical-state dump, caps 100 and 20 mixed" > $R/gpurun_out/r02_${stage}_pmc_case30.txt
cat $R/gpurun_out/r02_${stage}_bench_headline_kernel_trace.txt | cut -c1-200 | head -8
cat $R/gpurun_out/r02_${stage}_pmc_headline.txt | tail -30
cat $R/gpurun_out/r02_${stage}_pmc_case30.txt | tail -22
