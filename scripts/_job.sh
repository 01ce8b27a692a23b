cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
( for i in 1 2 3; do
    python scripts/ab_trees.py case30 case30_20 2>&1 | grep " us " | sed "s|^|base  |"
    ANM_BUILD_TAG=la0 ANM_EXTRA_HIPCC_FLAGS="-DANM_HYB_LIGHT_ALL=0" python scripts/ab_trees.py case30 case30_20 2>&1 | grep " us " | sed "s|^|la0   |"
    ANM_BUILD_TAG=bs0 ANM_EXTRA_HIPCC_FLAGS="-DANM_LDSX_BS_ALL=0" python scripts/ab_trees.py case30 case30_20 2>&1 | grep " us " | sed "s|^|bs0   |"
  done ) > gpurun_out/r06_p_ab_case30.txt 2>&1
cat gpurun_out/r06_p_ab_case30.txt
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06_p_gputests.txt
tail -3 gpurun_out/r06_p_gputests.txt
python bench.py --headline-only --no-cpu-baseline --steps 300 --warmup 30 2>&1 | grep -o "kernel_ms\": [0-9.]*\|ms_per_step\": [0-9.]*" | tee gpurun_out/r06_p_headline.txt
