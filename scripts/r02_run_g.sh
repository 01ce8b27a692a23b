#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q ) > gpurun_out/g_pytest.log 2>&1
tail -8 gpurun_out/g_pytest.log
