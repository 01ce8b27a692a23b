#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03_an_pytest_gpu.txt 2>&1
tail -6 gpurun_out/r03_an_pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r03_an_bench.json.log 2>&1
tail -c 1500 gpurun_out/r03_an_bench.json.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
