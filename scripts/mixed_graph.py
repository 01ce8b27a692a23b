"""The four-topology step of bench.mixed_side_figure captured once into a HIP graph (torch.cuda.graph: the fork / join over the
side streams becomes graph edges) and replayed, against the eager step.  usage: python scripts/mixed_graph.py"""
import os, sys
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
nets = [networks.anm6_network(), networks.three_bus_loop_network(gen_max=1.5), networks.synthetic_meshed_network(20, 3, 6), networks.synthetic_radial_network(30, 0)]
series = [anm6easy_series(), table(nets[1], 24, 1), table(nets[2], 48, 2), table(nets[3], 96, 3)]
tasks = [dict(network=nw, series=s, delta_t=dt, costs_clipping=(1, 100)) for nw, s, dt in zip(nets, series, (0.25, 0.5, 0.25, 0.25))]
E = 16384
env = MixedBatchedANMEnv(tasks, np.random.default_rng(5).integers(0, 4, E), device=dev, seed=7, tol=1e-6, max_iter=100, autoreset=True)
env.check_actions = False
env.reset(seed=7)
g = torch.Generator(device=dev).manual_seed(1)
lo, hi = env._act_low, env._act_high
pool = [lo + (hi - lo) * torch.rand(lo.shape, generator=g, dtype=torch.float64, device=dev) for _ in range(4)]
wall, evs = _timed_env_steps(env, pool, 40, dev)
print("eager: %.1f us per step (events %.1f)" % (wall * 1e6, evs * 1e6), flush=True)
static_a = pool[0].clone()
graph = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(graph):
    env.step(static_a)
torch.cuda.synchronize()
for i in range(10):
    static_a.copy_(pool[i % 4]); graph.replay()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); ev0.record()
for i in range(40):
    static_a.copy_(pool[i % 4]); graph.replay()
ev1.record(); torch.cuda.synchronize()
print("graph replay (+ action copy): %.1f us per step; terminated now %d of %d" % (ev0.elapsed_time(ev1) / 40 * 1e3, int(env.terminated.sum()), E), flush=True)
