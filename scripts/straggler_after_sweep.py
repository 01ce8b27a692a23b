"""Step time as a function of the hand-over point of the two-launch step (straggler_after), 65 536 and
131 072 environments, reference iteration cap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec
DEV = "cuda:0"
for E in (65536, 131072):
    row = []
    for sa in (None, 4, 6, 9, 12):
        env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, autoreset=True, tol=1e-6, max_iter=100, straggler_after=sa)
        env.check_actions = False
        env.reset(seed=1)
        g = torch.Generator(device=DEV).manual_seed(0)
        lo = torch.as_tensor(env.action_space.low, device=DEV); hi = torch.as_tensor(env.action_space.high, device=DEV)
        pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=DEV) for _ in range(8)]
        for i in range(10): env.step(pool[i % 8])
        torch.cuda.synchronize()
        t = time.perf_counter(); n = 100
        for i in range(n): env.step(pool[i % 8])
        torch.cuda.synchronize()
        row.append((sa, (time.perf_counter() - t) / n * 1e6))
    print(E, " ".join("after=%s: %.1f us" % r for r in row), flush=True)
