#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "# scripts/bench_case30_quick.py with ANM_RADIAL_GENERIC=1: the table-driven Newton loop of the radial family (any tree up to 64 buses), stop test on lane masks"
  ANM_RADIAL_GENERIC=1 timeout 300 python scripts/bench_case30_quick.py 2>&1 | grep -v amdgpu.ids
  echo "# the specialised loop (config 4), unchanged"
  timeout 300 python scripts/bench_case30_quick.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r03_y_radial_generic.txt
cat gpurun_out/r03_y_radial_generic.txt
( time timeout 1200 python -m pytest tests -m gpu -q -k "radial or golden or random or feeder or tree" ) > gpurun_out/r03_y_pytest_radial.txt 2>&1
tail -4 gpurun_out/r03_y_pytest_radial.txt
