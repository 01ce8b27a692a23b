#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q ) > gpurun_out/m_pytest.log 2>&1
tail -6 gpurun_out/m_pytest.log
timeout 900 python scripts/mpc_bench.py > gpurun_out/m_mpc.log 2>&1; cat gpurun_out/m_mpc.log
