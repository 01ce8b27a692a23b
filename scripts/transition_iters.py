"""Iteration histogram of bench.transition_figure's ANM6 batch (uniformly random inputs), and the launch time by hand-over
point.  usage: python scripts/transition_iters.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_anm_amd import networks
from gym_anm_amd.simulator import BatchedSimulator

dev = torch.device("cuda", 0)
E = 65536
for handoff in (None, 4, 6, 8, 10, 12, -1):
    kw = {} if handoff is None else dict(handoff_after=handoff)
    sim = BatchedSimulator(networks.anm6_network(), 0.25, 100, num_envs=E, device=dev, tol=1e-6, max_iter=100, **kw)
    m, b = sim.model, sim.model.baseMVA
    g = torch.Generator(device=dev).manual_seed(0)
    U = lambda lo, hi: (torch.as_tensor(lo, device=dev) + (torch.as_tensor(hi, device=dev) - torch.as_tensor(lo, device=dev))
                        * torch.rand((E, len(lo)), generator=g, dtype=torch.float64, device=dev))
    pl = U(m.dev_p_min[m.load_idx] * b, 0 * m.dev_p_min[m.load_idx]); pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * b)
    ps = U(m.dev_p_min[m.setp_idx] * b, m.dev_p_max[m.setp_idx] * b); qs = U(m.dev_q_min[m.setp_idx] * b, m.dev_q_max[m.setp_idx] * b)
    soc = U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx])
    for _ in range(3):
        sim.soc.copy_(soc); sim.transition(pl, pp, ps, qs)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(20):
        sim.soc.copy_(soc)
        ev0.record(); sim.transition(pl, pp, ps, qs); ev1.record(); torch.cuda.synchronize()
        tot += ev0.elapsed_time(ev1)
    it = sim.nr_iters.cpu().numpy()
    if handoff is None:
        h = np.bincount(it, minlength=101)
        print("iterations: " + " ".join("%d:%d" % (k, h[k]) for k in range(101) if h[k]))
        per_wave = (it.reshape(-1, 64) > 6).sum(1)
        print("lanes per wavefront still running after 6 iterations: max %d, mean %.2f, wavefronts with > 8: %d" % (per_wave.max(), per_wave.mean(), (per_wave > 8).sum()))
    print("handoff %s: %.1f us per launch (events)" % (handoff, tot / 20 * 1e3), flush=True)
