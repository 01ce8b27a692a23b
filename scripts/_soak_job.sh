cd $GRAFT_REPO_ROOT
( timeout 600 python scripts/soak_families.py 2>&1 | tail -3; timeout 900 python scripts/soak_two_launch.py 262144 600 2>&1 | tail -4; timeout 600 python scripts/soak_two_launch.py 1048576 150 2>&1 | tail -3 ) | grep -v amdgpu.ids | tee gpurun_out/r06_soak.txt
