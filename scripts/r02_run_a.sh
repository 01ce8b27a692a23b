#!/bin/bash
# round 2, GPU call A: whole -m gpu suite, hand-over sweep, bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/a_pytest.log 2>&1
tail -15 gpurun_out/a_pytest.log
timeout 600 python scripts/handoff_sweep.py 65536 4096 131072 1048576 > gpurun_out/a_sweep.log 2>&1
cat gpurun_out/a_sweep.log
timeout 600 python bench.py --steps 200 --warmup 20 > gpurun_out/a_bench.log 2>&1
tail -c 3000 gpurun_out/a_bench.log
