#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in graph_bench.py obs_modes_bench.py classes_bench.py bench_radial_quick.py mesh_caps.py; do
  echo "== $s"; timeout 300 python scripts/$s 2>&1 | grep -v amdgpu.ids | tail -6
done > gpurun_out/r03_ao_scripts.txt 2>&1
cat gpurun_out/r03_ao_scripts.txt | cut -c1-200
