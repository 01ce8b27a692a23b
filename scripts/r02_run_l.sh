#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/classes_bench.py > gpurun_out/l_classes.log 2>&1; cat gpurun_out/l_classes.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --headline-only > gpurun_out/l_torchrun1.log 2>&1; tail -c 1500 gpurun_out/l_torchrun1.log
timeout 300 python scripts/handoff_sweep.py 262144 1048576 > gpurun_out/l_sweep_big.log 2>&1; cat gpurun_out/l_sweep_big.log
timeout 300 python scripts/bench_mesh.py --thread > gpurun_out/l_mesh.log 2>&1; cat gpurun_out/l_mesh.log
