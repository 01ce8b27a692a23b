cd $GRAFT_REPO_ROOT
python scripts/straggler_levels_r06.py 2>&1 | grep "^E " | tee gpurun_out/r06_y_levels.txt
