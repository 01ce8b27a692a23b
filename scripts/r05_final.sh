#!/bin/bash
# usage (GPU box): bash scripts/r05_final.sh <stage>  -- end-of-round evidence: pytest -m gpu, the bench.py line, kernel trace + PMC
# passes of the headline command (scripts/r05_profile.sh head), PMC of config 4 and of the throughput regime
stage=${1:-z}
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r05_${stage}_pytest_gpu.txt
python bench.py > gpurun_out/r05_${stage}_bench.json.log 2>&1
bash scripts/r05_profile.sh $stage head c30 thr > gpurun_out/r05_${stage}_profile.log 2>&1
tail -3 gpurun_out/r05_${stage}_pytest_gpu.txt
ls gpurun_out | grep r05_${stage}
