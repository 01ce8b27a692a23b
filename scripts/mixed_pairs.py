"""Which launches of a mixed batch overlap: every pair (and all four) of bench.mixed_side_figure's topologies, ~4 096 environments
each, per-topology streams against one stream.  usage: python scripts/mixed_pairs.py"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import _timed_env_steps
from gym_anm_amd import networks
from gym_anm_amd.envs import MixedBatchedANMEnv
from gym_anm_amd.envs.anm6 import anm6easy_series
from gym_anm_amd.model import NetworkModel

dev = torch.device("cuda", 0)
def table(net, period, seed):
    m = NetworkModel(net, 0.25, 100)
    r, t = np.random.default_rng(seed), np.arange(period) / period
    rows = [m.dev_p_min[k] * m.baseMVA * (0.25 + 0.3 * (1 + np.sin(2 * np.pi * (t + r.uniform())))) for k in m.load_idx]
    rows += [m.dev_p_max[k] * m.baseMVA * (0.1 + 0.45 * (1 + np.sin(2 * np.pi * (t + r.uniform())))) for k in m.gen_idx]
    return np.array(rows)
names = ["anm6", "3bus", "mesh20", "case30"]
nets = [networks.anm6_network(), networks.three_bus_loop_network(gen_max=1.5), networks.synthetic_meshed_network(20, 3, 6), networks.synthetic_radial_network(30, 0)]
series = [anm6easy_series(), table(nets[1], 24, 1), table(nets[2], 48, 2), table(nets[3], 96, 3)]
tasks = [dict(network=nw, series=s, delta_t=dt, costs_clipping=(1, 100)) for nw, s, dt in zip(nets, series, (0.25, 0.5, 0.25, 0.25))]
combos = [(k,) for k in range(4)] + list(itertools.combinations(range(4), 2)) + [(0, 1, 2, 3), (0, 1, 2, 3)]
if len(sys.argv) > 1 and sys.argv[1] == "all4":
    combos = [(0, 1, 2, 3)] * 4
for n_c, combo in enumerate(combos):
    sub = [tasks[k] for k in combo]
    E = 4096 * len(combo)
    res = []
    deal = np.arange(E) % len(combo) if n_c % 2 == 0 else np.random.default_rng(5).integers(0, len(combo), E)   # round robin / at random
    for streams in (True, False):
        env = MixedBatchedANMEnv(sub, deal, device=dev, seed=7, tol=1e-6, max_iter=100, autoreset=True, streams=streams)
        env.check_actions = False
        env.reset(seed=7)
        g = torch.Generator(device=dev).manual_seed(1)
        lo, hi = env._act_low, env._act_high
        pool = [lo + (hi - lo) * torch.rand(lo.shape, generator=g, dtype=torch.float64, device=dev) for _ in range(4)]
        wall, evs = _timed_env_steps(env, pool, 40, dev)
        res.append(evs * 1e6)
    print("%-28s %-11s streams %.1f us   one stream %.1f us" % ("+".join(names[k] for k in combo), "round robin" if n_c % 2 == 0 else "at random", res[0], res[1]), flush=True)
