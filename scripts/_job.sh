cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r06_i_gputests.txt 2>&1
tail -8 gpurun_out/r06_i_gputests.txt
python scripts/ab_trees.py mesh30 mesh200 case30 case30_20 2>&1 | grep " us "
