"""The MPC kernel on a network beyond its register budget (30-bus feeder, 3 generators, 5 storage units: 123 rows and
42 variables per stage, row arrays spilling to scratch): time per solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gym_anm_amd import networks
from gym_anm_amd.agents.mpc import BatchedDCOPF
from gym_anm_amd.simulator import BatchedSimulator

net = networks.synthetic_radial_network(30, 0)
sim = BatchedSimulator(net, 0.25, 100, num_envs=2, device="cuda:0")
m = sim.model
rng = np.random.default_rng(0)
for N, E in ((1, 16384), (4, 16384), (10, 4096), (10, 16384)):
    s_ = BatchedDCOPF(sim, 0.995, 0.9, N)
    pl = -rng.uniform(0, 1, (E, len(m.load_idx), N)) * (-m.dev_p_min[m.load_idx])[None, :, None]
    pg = rng.uniform(0, 1, (E, len(m.gen_idx), N)) * m.dev_p_max[m.gen_idx][None, :, None]
    soc = rng.uniform(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx], (E, len(m.des_idx)))
    pl, pg, soc = (torch.as_tensor(a, device="cuda:0") for a in (pl, pg, soc))
    s_.solve(pl, pg, soc)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        s_.solve(pl, pg, soc)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 3
    print("case30  N=%2d E=%6d  %d rows x %d variables per stage  solve %.2f ms  %.3e programs/s  iterations mean %.1f max %d  unconverged %d" % (
        N, E, s_.dims.n_stage_rows, s_.dims.n_stage_vars, ms, E / ms * 1e3, float(s_.iters.double().mean()), int(s_.iters.max()),
        int((s_.iters >= s_.max_iter).sum())))
