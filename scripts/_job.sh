cd $GRAFT_REPO_ROOT
python scripts/handoff_r06.py 2>&1 | grep "@" | tee gpurun_out/r06_z_handoff.txt
