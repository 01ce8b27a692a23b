"""Debug aid: per-iteration traces (anm_mpc_opts.trace) of solves that did not converge in an earlier run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + "/tests")
sys.path.insert(0, ROOT + "/oracle")
import numpy as np
import torch

from gym_anm_amd import networks
from gym_anm_amd.agents.mpc import BatchedDCOPF
from gym_anm_amd.simulator import BatchedSimulator
import test_mpc as T

dev = sys.argv[1]
np.set_printoptions(linewidth=220, precision=3)


def sim_for(net):
    return T._host_sim(net) if dev == "cpu" else BatchedSimulator(net, 0.25, 100, num_envs=2, device=dev)


sim = sim_for(networks.anm6_network())
for N in (1,):
    g = np.load(ROOT + "/scripts/_dbg/mpc_unconverged_N%d.npz" % N)
    sol = BatchedDCOPF(sim, 0.995, 0.92, N, keep_trace=True)
    reps = 64  # the same programs in every lane of a wavefront
    sol.solve(np.repeat(g["pl"], reps, 0), np.repeat(g["pg"], reps, 0), np.repeat(g["soc"], reps, 0))
    print(N, dev, "iters", sol.iters.tolist()[::16], "rd", sol.info[::16, 2].cpu().numpy())
    for e in (0, reps):
        print(sol.trace[e, :41].cpu().numpy())
net = T.two_storage_network()
sim = sim_for(net)
m = sim.model
rng = np.random.default_rng(3)
nl, ng, ns = len(m.load_idx), len(m.gen_idx), len(m.des_idx)
n_cases = 16
for N in (1, 3, 8, 16):
    margin, gamma = float(rng.choice([0.8, 0.9, 1.0])), float(rng.choice([0.9, 0.995, 1.0]))
    pl = -rng.uniform(0, 1, (n_cases, nl, N)) * (-m.dev_p_min[m.load_idx])[None, :, None]
    pg = rng.uniform(0, 1, (n_cases, ng, N)) * m.dev_p_max[m.gen_idx][None, :, None] * (rng.random((n_cases, ng, N)) > 0.25)
    lo_, hi_ = m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx]
    pick = rng.integers(0, 4, (n_cases, ns))
    soc = np.where(pick == 0, lo_, np.where(pick == 1, hi_, lo_ + (hi_ - lo_) * rng.random((n_cases, ns))))
    solver = BatchedDCOPF(sim, gamma, margin, N, keep_trace=True)
    solver.solve(pl, pg, soc)
    print("two storage units N", N, solver.iters.tolist(), solver.info[:, 2].cpu().numpy())
    for e in range(n_cases):
        if solver.iters[e] >= 30:
            print("case", e, "soc", soc[e], "pg", pg[e])
            print(solver.trace[e, :41].cpu().numpy())
            break
