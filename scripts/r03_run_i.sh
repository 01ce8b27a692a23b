#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x -k "different_network or heterogeneous or parameter_classes" ) > gpurun_out/r03_i_pytest.txt 2>&1
tail -15 gpurun_out/r03_i_pytest.txt
timeout 900 python scripts/classes_bench.py 2>&1 | grep "^impl" | tee gpurun_out/r03_i_parameter_classes.txt
