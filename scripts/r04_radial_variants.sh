#!/bin/bash
# config 4 (case30, 16384 transitions, caps 100 / 20) through side-by-side builds of the lane-group kernel:
# child slots fetched per round (ANM_LDSX_FETCH) x wavefronts per SIMD the kernel is compiled for (ANM_RADIAL_WAVES)
run() { # tag flags
  echo "== $1 ($2)"
  ANM_BUILD_TAG=$1 ANM_EXTRA_HIPCC_FLAGS="$2" timeout 200 python scripts/bench_case30_quick.py 2>&1 | grep -v amdgpu.ids
}
run "" ""
run f1w4 "-DANM_LDSX_FETCH=1 -DANM_RADIAL_WAVES=4"
run f2w4 "-DANM_LDSX_FETCH=2 -DANM_RADIAL_WAVES=4"
run f3w4 "-DANM_LDSX_FETCH=3 -DANM_RADIAL_WAVES=4"
run f1w3 "-DANM_LDSX_FETCH=1 -DANM_RADIAL_WAVES=3"
run f2w3 "-DANM_LDSX_FETCH=2 -DANM_RADIAL_WAVES=3"
