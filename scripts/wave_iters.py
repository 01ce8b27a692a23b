"""Wave-level iteration statistics of the bench workload (ANM6Easy, 65536 envs, random agent):
how many Newton trips the waves execute, to turn PMC instruction/cycle totals into per-trip figures."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec
dev = torch.device("cuda", 0)
E = 65536
env = ANM6EasyVec(num_envs=E, device=dev, seed=1234, tol=1e-6, max_iter=100, autoreset=True)
env.check_actions = False
env.reset(seed=1234)
g = torch.Generator(device=dev).manual_seed(99)
lo = torch.as_tensor(env.action_space.low, device=dev); hi = torch.as_tensor(env.action_space.high, device=dev)
pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=dev) for _ in range(16)]
tot100 = tot20 = 0.0; n = 0; nstr = 0; lanes = 0; mx = []
for i in range(40):
    env.step(pool[i % 16])
    if i < 10:
        continue
    it = env.simulator.nr_iters.view(-1, 64)
    wmax = it.max(dim=1).values.double()
    tot100 += float(wmax.sum()); tot20 += float(wmax.clamp(max=20).sum()); n += 1
    nstr += int((wmax > 20).sum()); lanes += int((it > 20).sum())
print("per launch: wave-trips cap100 %.0f, cap20 %.0f, delta %.0f; waves with >20 iterations %.1f; lanes with >20 iterations %.1f"
      % (tot100 / n, tot20 / n, (tot100 - tot20) / n, nstr / n, lanes / n))
h = torch.bincount(env.simulator.nr_iters.clamp(max=100), minlength=101)
print("iteration histogram (last step):", {int(k): int(v) for k, v in enumerate(h) if v})
