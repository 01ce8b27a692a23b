#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q ) > gpurun_out/b_pytest.log 2>&1
tail -15 gpurun_out/b_pytest.log
for tag in fake; do
  ANM_BUILD_TAG=$tag ANM_EXTRA_HIPCC_FLAGS="-DANM_GROUP_FAKE_XFER" timeout 300 python scripts/handoff_sweep.py 65536 > gpurun_out/b_sweep_$tag.log 2>&1
  cat gpurun_out/b_sweep_$tag.log
done
