cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/r06_t_bench.json 2> gpurun_out/r06_t_bench.err
tail -c 300 gpurun_out/r06_t_bench.json
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r06_t_gputests.txt
tail -2 gpurun_out/r06_t_gputests.txt
bash scripts/r06_profile.sh t head thr c30 mix > gpurun_out/r06_t_profile.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r06_t_gputests.txt
ls gpurun_out | grep r06_t
