"""rocprofv3 workload: `launches` Simulator.transition launches on a synthetic meshed network of n_bus buses
(one workgroup per environment above 65 buses).  usage: large_network_workload.py n_bus num_envs launches [cap]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gym_anm_amd import networks
from gym_anm_amd.simulator import BatchedSimulator

n_bus, E, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 100
seed, chords = {30: (6, 4), 64: (10, 24), 100: (12, 12), 200: (13, 30), 300: (14, 40), 513: (15, 40)}[n_bus]
DEV = "cuda:0"
sim = BatchedSimulator(networks.synthetic_meshed_network(n_bus, seed, chords), 0.25, 100, num_envs=E, device=DEV, max_iter=cap, impl="mesh")
m, b, f = sim.model, sim.model.baseMVA, min(1.0, 40.0 / n_bus)
rng = np.random.default_rng(seed)
U = lambda lo, hi, s: torch.as_tensor(np.asarray(lo) * s + (np.asarray(hi) * s - np.asarray(lo) * s) * rng.uniform(size=(E, len(lo))), device=DEV)  # noqa: E731
pl = U(m.dev_p_min[m.load_idx], 0 * m.dev_p_min[m.load_idx], 0.6 * b * f)
pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx], b)
ps = U(m.dev_p_min[m.setp_idx], m.dev_p_max[m.setp_idx], 1.2 * b * f)
qs = U(m.dev_q_min[m.setp_idx], m.dev_q_max[m.setp_idx], 1.2 * b * f)
soc = U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx], 1.0)
pl[-max(1, E // 1024):] *= 40.0 / f
for _ in range(n):
    sim.soc.copy_(soc)
    sim.transition(pl, pp, ps, qs)
torch.cuda.synchronize()
print("mesh%d: %d transitions per launch, %d launches, lanes per environment %d, converged %.4f, mean iterations %.2f" % (
    n_bus, E, n, sim.lanes_per_env, float(sim.pfe_converged.double().mean()), float(sim.nr_iters.double().mean())))
