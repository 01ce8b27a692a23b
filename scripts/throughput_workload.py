"""Workload for profiling the throughput regime (SURVEY 8d config 5's per-GPU load and beyond): ANM6Easy, random agent,
the two-launch step (k_step_rows + k_step_stragglers + k_step_scatter), E environments on one GPU.

    python scripts/throughput_workload.py 524288 [steps [straggler_after [straggler_mid]]]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_anm_amd.envs import ANM6EasyVec

DEV = "cuda:0"
E = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
kw = {}
if len(sys.argv) > 3:
    kw["straggler_after"] = int(sys.argv[3])
if len(sys.argv) > 4:
    kw["straggler_mid"] = None if sys.argv[4] == "none" else int(sys.argv[4])
env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, autoreset=True, tol=1e-6, max_iter=100, **kw)
env.check_actions = False
env.reset(seed=1)
g = torch.Generator(device=DEV).manual_seed(0)
lo, hi = torch.as_tensor(env.action_space.low, device=DEV), torch.as_tensor(env.action_space.high, device=DEV)
pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=DEV) for _ in range(4)]
for i in range(8):
    env.step(pool[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    env.step(pool[i % 4])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("E %d: %.1f us per step, %.3e env-steps/s, %.1f GB/s algorithmic (250 B per env-step), launches per step %d, terminated %.4f"
      % (E, dt * 1e6, E / dt, 250 * E / dt / 1e9, 3 if env._ws is not None else 1, float(env.terminated.double().mean())))
