#!/usr/bin/env python
"""Does capturing env.step() launches in a HIP graph (torch.cuda.CUDAGraph) pay?  Eager vs replay."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_anm_amd.envs import ANM6EasyVec

dev = torch.device("cuda", 0)
for E, cap in ((65536, 100), (65536, 20), (4096, 100), (4096, 20)):
    env = ANM6EasyVec(num_envs=E, device=dev, seed=1234, tol=1e-6, max_iter=cap, autoreset=True)
    env.check_actions = False
    env.reset(seed=1234)
    g = torch.Generator(device=dev).manual_seed(99)
    lo = torch.as_tensor(env.action_space.low, device=dev); hi = torch.as_tensor(env.action_space.high, device=dev)
    pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=dev) for _ in range(16)]
    for i in range(32):
        env.step(pool[i % 16])
    torch.cuda.synchronize()
    n = 320
    t0 = time.perf_counter()
    for i in range(n):
        env.step(pool[i % 16])
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / n
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(16):
            env.step(pool[i])
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        for i in range(16):
            env.step(pool[i])
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // 16):
        graph.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / n
    print("E=%6d cap=%3d  eager %.1f us/step  graph(16 steps) %.1f us/step  term=%.4f" %
          (E, cap, eager * 1e6, rep * 1e6, float(env.terminated.double().mean())), flush=True)
