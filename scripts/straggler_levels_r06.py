import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from gym_anm_amd.envs import ANM6EasyVec
DEV = "cuda:0"
for rep in range(2):
  for E in (131072, 262144, 524288, 1048576):
    for mid in (None, 8, 9, 10, 12):
        env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, autoreset=True, tol=1e-6, max_iter=100, straggler_after=6, straggler_mid=mid)
        env.check_actions = False
        env.reset(seed=1)
        g = torch.Generator(device=DEV).manual_seed(0)
        lo, hi = torch.as_tensor(env.action_space.low, device=DEV), torch.as_tensor(env.action_space.high, device=DEV)
        pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=DEV) for _ in range(4)]
        for i in range(8): env.step(pool[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(60): env.step(pool[i % 4])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 60
        print("E %8d  second level at %-4s  %.1f us per step" % (E, mid, dt * 1e6), flush=True)
        del env
