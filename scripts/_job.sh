cd $GRAFT_REPO_ROOT
( bash scripts/ab_variants.sh "base=" "ra0=-DANM_INWAVE_REDUCE_ALWAYS=0" "base=" "ra0=-DANM_INWAVE_REDUCE_ALWAYS=0" ) > gpurun_out/r06_u_ab_headline.txt 2>&1
cat gpurun_out/r06_u_ab_headline.txt
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r06_u_gputests.txt
tail -2 gpurun_out/r06_u_gputests.txt
python scripts/mesh_occupancy_bench.py anm6 2>&1 | tail -3 | tee gpurun_out/r06_u_transition.txt
