cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06_m_gputests.txt
tail -3 gpurun_out/r06_m_gputests.txt
( bash scripts/ab_variants.sh "base=" "old=-DANM_HORNER_PAIR=0 -DANM_GROUP_POLY_UNMASKED=0 -DANM_GROUP_MERGED_REGIONS=0" "base=" "old=-DANM_HORNER_PAIR=0 -DANM_GROUP_POLY_UNMASKED=0 -DANM_GROUP_MERGED_REGIONS=0" ) > gpurun_out/r06_m_ab_headline.txt 2>&1
cat gpurun_out/r06_m_ab_headline.txt
( for i in 1 2 3; do
    python scripts/ab_trees.py case30 case30_20 2>&1 | grep " us " | sed "s|^|base  |"
    ANM_BUILD_TAG=pf0 ANM_EXTRA_HIPCC_FLAGS="-DANM_RADIAL_PREFETCH=0" python scripts/ab_trees.py case30 case30_20 2>&1 | grep " us " | sed "s|^|pf0   |"
    ANM_BUILD_TAG=old ANM_EXTRA_HIPCC_FLAGS="-DANM_HORNER_PAIR=0 -DANM_GROUP_POLY_UNMASKED=0 -DANM_GROUP_MERGED_REGIONS=0" python scripts/ab_trees.py case30 case30_20 2>&1 | grep " us " | sed "s|^|old   |"
  done ) > gpurun_out/r06_m_ab_case30.txt 2>&1
cat gpurun_out/r06_m_ab_case30.txt
