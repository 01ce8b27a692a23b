#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q -k "radial or headline or list or golden" ) > gpurun_out/f_pytest.log 2>&1
tail -12 gpurun_out/f_pytest.log
timeout 300 python scripts/bench_case30_quick.py > gpurun_out/f_case30_spec.log 2>&1; cat gpurun_out/f_case30_spec.log
ANM_RADIAL_GENERIC=1 timeout 300 python scripts/bench_case30_quick.py > gpurun_out/f_case30_gen.log 2>&1; cat gpurun_out/f_case30_gen.log
