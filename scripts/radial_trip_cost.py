"""Cost of ONE Newton trip of the tree kernel (config 4's 30-bus feeder), measured by forcing trip counts: tol = 0 makes
every solve run to the cap, so (time at cap b - time at cap a) / (b - a) is a trip -- for the whole batch (16 384 transitions:
throughput, three wavefronts per SIMD) and for a batch of 64 (32 wavefronts on an otherwise idle chip: the latency of the chain a
diverging solve imposes at the reference's cap).  Run from the root of a tree (scripts/ab_trees.sh style)."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from gym_anm_amd import networks
from gym_anm_amd.simulator import BatchedSimulator

dev = torch.device("cuda", 0)
net = networks.synthetic_radial_network(30, 0)


def launch_us(E, cap, n=30):
    sim = BatchedSimulator(net, 0.25, 100, num_envs=E, device=dev, tol=0.0, max_iter=cap)
    m, b = sim.model, sim.model.baseMVA
    g = torch.Generator(device=dev).manual_seed(0)

    def U(lo, hi):
        lo, hi = torch.as_tensor(lo, device=dev), torch.as_tensor(hi, device=dev)
        return lo + (hi - lo) * torch.rand((E, lo.numel()), generator=g, dtype=torch.float64, device=dev)

    pl = U(m.dev_p_min[m.load_idx] * b, 0 * m.dev_p_min[m.load_idx])
    pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * b)
    ps = U(m.dev_p_min[m.setp_idx] * b, m.dev_p_max[m.setp_idx] * b)
    qs = U(m.dev_q_min[m.setp_idx] * b, m.dev_q_max[m.setp_idx] * b)
    soc = U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx])
    for _ in range(3):
        sim.soc.copy_(soc)
        sim.transition(pl, pp, ps, qs)
    torch.cuda.synchronize(dev)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            sim.soc.copy_(soc)
            sim.transition(pl, pp, ps, qs)
        torch.cuda.synchronize(dev)
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6


for E in (16384, 64):
    a, b_ = launch_us(E, 10), launch_us(E, 40)
    print("E %6d: cap 10 %7.1f us, cap 40 %7.1f us -> %.3f us per trip" % (E, a, b_, (b_ - a) / 30.0), flush=True)
