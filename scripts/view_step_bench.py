"""The step through a batch view in the thread-per-environment family (k_step_view): two ANM6 tasks (the second with other
line ratings) dealt to E environments, MixedBatchedANMEnv, random actions.  usage: python scripts/view_step_bench.py [E ...]"""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_anm_amd import networks
from gym_anm_amd.envs import MixedBatchedANMEnv
from gym_anm_amd.envs.anm6 import anm6easy_series

dev = torch.device("cuda", 0)
net_a = networks.anm6_network()
net_b = copy.deepcopy(net_a)
net_b["branch"][:, 5] *= 1.1   # other ratings: a second model
for E in ([int(a) for a in sys.argv[1:]] or [16384, 262144]):
    tasks = [dict(network=net_a, series=anm6easy_series(), costs_clipping=(1, 100)), dict(network=net_b, series=anm6easy_series(), costs_clipping=(1, 100))]
    env = MixedBatchedANMEnv(tasks, np.arange(E) % 2, device=dev, seed=1, tol=1e-6, autoreset=True)
    env.check_actions = False
    env.reset(seed=1)
    g = torch.Generator(device=dev).manual_seed(0)
    lo, hi = env._act_low, env._act_high
    pool = [lo + (hi - lo) * torch.rand(lo.shape, generator=g, dtype=torch.float64, device=dev) for _ in range(4)]
    for i in range(5): env.step(pool[i % 4])
    torch.cuda.synchronize()
    t = time.perf_counter(); n = 40
    for i in range(n): env.step(pool[i % 4])
    torch.cuda.synchronize()
    print("view step E=%d tag=%s: %.1f us per step (families %s)" % (E, os.environ.get("ANM_BUILD_TAG", "-"), (time.perf_counter() - t) / n * 1e6, env.impls), flush=True)
