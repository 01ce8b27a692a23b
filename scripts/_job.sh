cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
OLD="old=-DANM_HORNER_PAIR=0 -DANM_GROUP_POLY_UNMASKED=0 -DANM_GROUP_MERGED_REGIONS=0"
( bash scripts/ab_variants.sh "base=" "$OLD" "base=" "$OLD" ) > gpurun_out/r06_n_ab_headline.txt 2>&1
cat gpurun_out/r06_n_ab_headline.txt
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06_n_gputests.txt
tail -3 gpurun_out/r06_n_gputests.txt
python scripts/handoff_fine.py > gpurun_out/r06_n_handoff.txt 2>&1
cat gpurun_out/r06_n_handoff.txt
