cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 300 gpurun_out/bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
