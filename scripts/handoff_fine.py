"""Hand-over point 4..8 at 65 536 environments, reference cap: finer than handoff_sweep.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec
DEV = "cuda:0"
def run(E, cap, n=150, **kw):
    env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, autoreset=True, tol=1e-6, max_iter=cap, **kw)
    env.check_actions = False
    env.reset(seed=1)
    g = torch.Generator(device=DEV).manual_seed(0)
    lo = torch.as_tensor(env.action_space.low, device=DEV); hi = torch.as_tensor(env.action_space.high, device=DEV)
    pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=DEV) for _ in range(8)]
    for i in range(10): env.step(pool[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): env.step(pool[i % 8])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    print("  ".join("@%d %6.1f" % (h, run(65536, 100, handoff_after=h, straggler_after=None)) for h in (4, 5, 6, 7, 8)), flush=True)
