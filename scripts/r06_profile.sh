#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/r06_profile.sh <stage> [what...]
# kernel traces + PMC passes (one counter group per pass, kernel-trace only); summaries land in
# gpurun_out/r06_<stage>_*.txt ready to be copied to profiles/.  what: head mpc thr c30 wg trn (default: head mpc thr c30)
stage=$1; shift
what="${@:-head thr c30 mix}"
R=$GRAFT_REPO_ROOT
out=/tmp/prof_r06_$stage   # raw rocprofv3 output stays on the box: only the summaries travel
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pmc() {  # pmc <dir> <cmd...> : four passes
  d=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $d/p1 -- "$@" > $d.p1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $d/p2 -- "$@" > $d.p2.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d/p3 -- "$@" > $d.p3.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $d/p4 -- "$@" > $d.p4.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $d/p5 -- "$@" > $d.p5.log 2>&1
}
trace() {  # trace <name> <note> <cmd...>
  nm=$1; note=$2; shift; shift
  rocprofv3 --kernel-trace --stats -d $out/trace_$nm -- "$@" > $out/trace_$nm.log 2>&1
  ( python $R/scripts/prof_summary.py $out/trace_$nm "$note"; echo "# output of the command:"; grep -v amdgpu.ids $out/trace_$nm.log | tail -12 ) > $R/gpurun_out/r06_${stage}_${nm}_kernel_trace.txt
}
for w in $what; do
case $w in
head)
  HEAD="python $R/bench.py --no-cpu-baseline --headline-only --steps 200 --warmup 20"
  trace bench_headline "command: bench.py --no-cpu-baseline --headline-only --steps 200 --warmup 20" $HEAD
  pmc $out/pmc_head $HEAD
  ANM_PMC_KERNEL=k_step_rows python $R/scripts/pmc_summary.py $out/pmc_head "headline: ANM6Easy 65536 envs, tol 1e-6, cap 100, autoreset, in-wave lane-group hand-over" > $R/gpurun_out/r06_${stage}_pmc_headline.txt
  ;;
mpc)
  for N in 1 10; do
    CMD="python $R/scripts/mpc_workload.py $N 20"
    trace mpc_N$N "command: scripts/mpc_workload.py $N 20 (65536 ANM6Easy programs per launch of k_mpc, $N stage(s))" $CMD
    pmc $out/pmc_mpc$N $CMD
    ANM_PMC_KERNEL=k_mpc python $R/scripts/pmc_summary.py $out/pmc_mpc$N "k_mpc: 65536 programs, N = $N" > $R/gpurun_out/r06_${stage}_pmc_mpc_N$N.txt
  done
  ;;
thr)
  for E in 524288 1048576; do
    CMD="python $R/scripts/throughput_workload.py $E 40"
    trace thr_$E "command: scripts/throughput_workload.py $E 40 (two-launch step)" $CMD
    pmc $out/pmc_thr$E $CMD
    for k in k_step_rows k_step_stragglers k_step_scatter; do
      ANM_PMC_KERNEL=$k python $R/scripts/pmc_summary.py $out/pmc_thr$E "throughput regime: ANM6Easy $E envs on one GPU, two-launch step; kernel $k"
    done > $R/gpurun_out/r06_${stage}_pmc_throughput_$E.txt
  done
  ;;
trn)
  CMD="python $R/scripts/mesh_occupancy_bench.py anm6"
  trace transition_anm6 "command: scripts/mesh_occupancy_bench.py anm6 (Simulator.transition, 65 536 ANM6 transitions per launch of k_transition, dump on)" $CMD
  ;;
wg)
  for NB in 30 200; do
    CMD="python $R/scripts/large_network_workload.py $NB 4096 12"
    trace mesh$NB "command: scripts/large_network_workload.py $NB 4096 12 (general lane-group family; 200 buses: one workgroup of 256 lanes per environment)" $CMD
    pmc $out/pmc_mesh$NB $CMD
    ANM_PMC_KERNEL=k_mesh python $R/scripts/pmc_summary.py $out/pmc_mesh$NB "k_mesh: synthetic meshed network of $NB buses, 4096 transitions per launch, cap 100" > $R/gpurun_out/r06_${stage}_pmc_mesh$NB.txt
  done
  ;;
mix)
  CMD="python $R/scripts/mixed_workload.py"
  trace mixed "command: scripts/mixed_workload.py (bench.mixed_side_figure: ANM6 + 3-bus loop + meshed 20 + 30-bus feeder dealt to 16 384 environments)" $CMD
  pmc $out/pmc_mix $CMD
  for k in k_step_view k_mesh k_radial; do
    ANM_PMC_KERNEL=$k python $R/scripts/pmc_summary.py $out/pmc_mix "mixed batch of four topologies, 16 384 environments; kernel $k"
  done > $R/gpurun_out/r06_${stage}_pmc_mixed.txt
  ;;
c30)
  C30="python $R/scripts/bench_case30_quick.py"
  pmc $out/pmc_c30 $C30
  ANM_PMC_KERNEL=k_radial python $R/scripts/pmc_summary.py $out/pmc_c30 "config 4: case30 radial, 16384 envs, Simulator.transition with the electrical-state dump, caps 100 and 20 mixed" > $R/gpurun_out/r06_${stage}_pmc_case30.txt
  ;;
esac
done
ls $R/gpurun_out/r06_${stage}_* | head -30
