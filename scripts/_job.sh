cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/r06_q_bench.json 2> gpurun_out/r06_q_bench.err
tail -c 400 gpurun_out/r06_q_bench.json
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r06_q_gputests.txt
tail -2 gpurun_out/r06_q_gputests.txt
bash scripts/r06_profile.sh q head thr c30 > gpurun_out/r06_q_profile.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
ls gpurun_out | grep r06_q
