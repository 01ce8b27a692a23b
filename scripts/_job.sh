cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_headline.py -m gpu -q -k "small_tree or hybrid_tree_kernel" 2>&1 | tail -5
