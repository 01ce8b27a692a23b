#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_headline.py "tests/test_gpu_parity.py::test_two_phase_step_is_bit_identical" -m gpu -q -x ) > gpurun_out/c_pytest.log 2>&1
tail -5 gpurun_out/c_pytest.log
timeout 300 python scripts/handoff_sweep.py 65536 > gpurun_out/c_sweep_dpp.log 2>&1
cat gpurun_out/c_sweep_dpp.log
export ANM_BUILD_TAG=nodpp ANM_NO_DPP=1
( timeout 600 python -m pytest tests/test_gpu_headline.py -m gpu -q -x -k "headline_config" ) > gpurun_out/c_pytest_nodpp.log 2>&1
tail -3 gpurun_out/c_pytest_nodpp.log
timeout 300 python scripts/handoff_sweep.py 65536 > gpurun_out/c_sweep_nodpp.log 2>&1
cat gpurun_out/c_sweep_nodpp.log
