#!/bin/bash
# usage: scripts/pmc_run.sh <tag> <bench args...>   (run on the GPU box via gpurun)
# Separate rocprofv3 passes (PMC slot limits: SQ 8, TCC 4): instruction mix, stall breakdown, HBM bytes.
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/p1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p3 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/p4 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/p4.log 2>&1
find $out -name "*.csv" | head -20
