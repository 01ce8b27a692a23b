"""Throughput cost of per-environment heterogeneous networks (parameter classes): ANM6Easy, 65536 environments,
tol 1e-6, autoreset, random agent -- one network vs 64 / 1024 perturbed copies (one class per aligned block of
1024 / 64 environments), both kernel families."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_anm_amd import networks
from gym_anm_amd.envs import ANM6EasyVec
DEV = "cuda:0"; E = 65536
base = networks.anm6_network()
def run(n_var, impl, n=100, scattered=False):
    kw = {}
    if n_var > 1:
        ev = np.repeat(np.arange(n_var), E // n_var)
        if scattered:   # a different network from one environment to the next (lane-group families only)
            ev = np.random.default_rng(0).permutation(ev)
        kw = dict(variants=[networks.perturbed_network(base, 100 + k, rel=0.05) for k in range(1, n_var)], env_variant=ev)
    env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, autoreset=True, tol=1e-6, impl=impl, **kw)
    env.check_actions = False; env.reset(seed=1)
    g = torch.Generator(device=DEV).manual_seed(0)
    lo = torch.as_tensor(env.action_space.low, device=DEV); hi = torch.as_tensor(env.action_space.high, device=DEV)
    pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=DEV) for _ in range(8)]
    for i in range(10): env.step(pool[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): env.step(pool[i % 8])
    e1.record(); torch.cuda.synchronize()
    print("impl=%-6s networks=%4d %-22s %7.1f us/step  terminated %.4f" % (env.simulator.impl, n_var,
          "(per environment)" if scattered else "(blocks of 64 or more)", e0.elapsed_time(e1) / n * 1e3,
          float(env.terminated.double().mean())), flush=True)
for impl in ("thread", "radial", "mesh"):
    for n_var in (1, 64, 1024):
        run(n_var, impl)
for impl in ("radial", "mesh"):
    for n_var in (64, 1024):
        run(n_var, impl, scattered=True)
