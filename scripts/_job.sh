cd $GRAFT_REPO_ROOT
(
for i in 1 2 3; do
  ( cd _ab/head && python $GRAFT_REPO_ROOT/scripts/ab_trees.py case30 case30_20 2>&1 | grep " us " | sed "s|^|head  |" )
  ANM_BUILD_TAG=lgf1 ANM_EXTRA_HIPCC_FLAGS="-DANM_DEV_LANE_GROUPS_ONLY -DANM_HYB_FETCH=1" python scripts/ab_trees.py case30 case30_20 2>&1 | grep " us " | sed "s|^|lgf1  |"
  ANM_BUILD_TAG=lgnh ANM_EXTRA_HIPCC_FLAGS="-DANM_DEV_LANE_GROUPS_ONLY" python scripts/ab_trees.py case30 case30_20 2>&1 | grep " us " | sed "s|^|lgnh  |"
done
( cd _ab/head && python $GRAFT_REPO_ROOT/scripts/radial_trip_cost.py 2>&1 | grep "per trip" | sed "s|^|head  |" )
ANM_BUILD_TAG=lgf1 ANM_EXTRA_HIPCC_FLAGS="-DANM_DEV_LANE_GROUPS_ONLY -DANM_HYB_FETCH=1" python scripts/radial_trip_cost.py 2>&1 | grep "per trip" | sed "s|^|lgf1  |"
ANM_BUILD_TAG=lgnh ANM_EXTRA_HIPCC_FLAGS="-DANM_DEV_LANE_GROUPS_ONLY" python scripts/radial_trip_cost.py 2>&1 | grep "per trip" | sed "s|^|lgnh  |"
) > gpurun_out/r06_e_radial_hybrid_fetch1.txt 2>&1
cat gpurun_out/r06_e_radial_hybrid_fetch1.txt
