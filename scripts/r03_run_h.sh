#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x -k "heterogeneous or parameter_classes" ) > gpurun_out/r03_h_pytest.txt 2>&1
tail -15 gpurun_out/r03_h_pytest.txt
timeout 900 python scripts/precision_report.py > gpurun_out/r03_h_precision_report_fp64_vs_fp32.txt 2>&1
grep -c "^==" gpurun_out/r03_h_precision_report_fp64_vs_fp32.txt; grep "flag mismatches\|^==" gpurun_out/r03_h_precision_report_fp64_vs_fp32.txt | tail -24
