"""Workload for profiling the straggler launch alone: ANM6Easy, random agent, two-launch step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec
DEV = "cuda:0"
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
after = int(sys.argv[2]) if len(sys.argv) > 2 else 8
env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, autoreset=True, tol=1e-6, max_iter=100, straggler_after=after)
env.check_actions = False
env.reset(seed=1)
g = torch.Generator(device=DEV).manual_seed(0)
lo = torch.as_tensor(env.action_space.low, device=DEV); hi = torch.as_tensor(env.action_space.high, device=DEV)
pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=DEV) for _ in range(8)]
for i in range(40):
    env.step(pool[i % 8])
torch.cuda.synchronize()
print("terminated frac", float(env.terminated.double().mean()))
