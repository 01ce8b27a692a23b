cd $GRAFT_REPO_ROOT
bash scripts/r06_profile.sh j head thr c30 mix > gpurun_out/r06_j_profile.log 2>&1
tail -12 gpurun_out/r06_j_profile.log
