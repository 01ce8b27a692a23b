import os, sys
sys.path.insert(0, os.getcwd())
import torch
from gym_anm_amd.envs import ANM6EasyVec
DEV = "cuda:0"
def run(E, cap, n=300, **kw):
    env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, autoreset=True, tol=1e-6, max_iter=cap, **kw)
    env.check_actions = False
    env.reset(seed=1)
    g = torch.Generator(device=DEV).manual_seed(0)
    lo = torch.as_tensor(env.action_space.low, device=DEV); hi = torch.as_tensor(env.action_space.high, device=DEV)
    pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=DEV) for _ in range(16)]
    for i in range(20): env.step(pool[i % 16])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): env.step(pool[i % 16])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(4):
    print("  ".join("@%d %6.2f" % (h, run(65536, 100, handoff_after=h, straggler_after=None)) for h in (5, 6, 7)), flush=True)
