cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r06_l_bench.json 2> gpurun_out/r06_l_bench.err
tail -c 300 gpurun_out/r06_l_bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
