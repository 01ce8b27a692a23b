cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python bench.py --headline-only --no-cpu-baseline --steps 300 --warmup 30 2>&1 | grep -o "kernel_ms\": [0-9.]*\|\"ms_per_step\": [0-9.]*" | tr '\n' ' '; echo; done | tee gpurun_out/r06_s_headline.txt
python scripts/throughput_workload.py 524288 100 2>&1 | tail -1 | tee -a gpurun_out/r06_s_headline.txt
python scripts/ab_trees.py case30 case30_20 2>&1 | grep " us " | tee -a gpurun_out/r06_s_headline.txt
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r06_s_gputests.txt
tail -2 gpurun_out/r06_s_gputests.txt
