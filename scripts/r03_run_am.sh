#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --global-envs 524288 --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: d[k] for k in ('metric', 'value', 'ms_per_step', 'n_gpus', 'scaling', 'steps', 'warmup')}))
print(json.dumps({k: d['config'][k] for k in ('workload', 'num_envs_per_gpu', 'global_num_envs', 'parallelism', 'rccl_ranks')}))
print(json.dumps(d['roofline'])[:400])
" | tee gpurun_out/r03_am_strong_1rank.txt
