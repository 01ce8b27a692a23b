cd $GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/test_gpu_headline.py -m gpu -q -k "dpp_plan or neighbours" 2>&1 | tail -15 ) | tee gpurun_out/r06_x_newtests.txt
