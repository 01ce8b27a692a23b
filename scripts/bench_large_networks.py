"""Networks above 65 buses (one WORKGROUP per environment, k_mesh<.., WG>): Simulator.transition per launch on
synthetic meshed feeders of 100 ... 513 buses, with the oracle (the reference's algorithm: SciPy sparse Jacobian +
spsolve) timed on the same transitions on one host core beside it.  Loads are scaled with 40 / n_bus (the synthetic
feeders carry the same load per bus whatever their size) and a few hopeless cases run to the iteration cap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import torch
from gym_anm_amd import networks
from gym_anm_amd.simulator import BatchedSimulator
DEV = "cuda:0"


def run(n_bus, seed, n_chords, E, n=10, cpu_sample=6):
    net = networks.synthetic_meshed_network(n_bus, seed, n_chords)
    sim = BatchedSimulator(net, 0.25, 100, num_envs=E, device=DEV)
    m, b, f = sim.model, sim.model.baseMVA, 40.0 / n_bus
    rng = np.random.default_rng(seed)
    U = lambda lo, hi, s: np.asarray(lo) * s + (np.asarray(hi) * s - np.asarray(lo) * s) * rng.uniform(size=(E, len(lo)))  # noqa: E731
    pl = U(m.dev_p_min[m.load_idx], 0 * m.dev_p_min[m.load_idx], 0.6 * b * f)
    pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx], b)
    ps = U(m.dev_p_min[m.setp_idx], m.dev_p_max[m.setp_idx], 1.2 * b * f)
    qs = U(m.dev_q_min[m.setp_idx], m.dev_q_max[m.setp_idx], 1.2 * b * f)
    soc = U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx], 1.0)
    pl[-max(1, E // 1024):] *= 40.0 / f   # the diverging solves every large batch has
    T = lambda a: torch.as_tensor(a, device=DEV)  # noqa: E731
    tpl, tpp, tps, tqs, tsoc = T(pl), T(pp), T(ps), T(qs), T(soc)
    for _ in range(2):
        sim.soc.copy_(tsoc)
        sim.transition(tpl, tpp, tps, tqs)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    dt = 0.0
    for _ in range(n):
        sim.soc.copy_(tsoc)
        ev[0].record()
        sim.transition(tpl, tpp, tps, tqs)
        ev[1].record()
        torch.cuda.synchronize()
        dt += ev[0].elapsed_time(ev[1]) * 1e-3 / n
    import anm_oracle as O
    on = O.parse_network(net, 0.25, 100)
    t0 = time.perf_counter()
    for e in range(cpu_sample):
        O.transition(on, pl[e], pp[e], ps[e], qs[e], soc[e], tol=1e-5, sparse=True)
    cpu = (time.perf_counter() - t0) / cpu_sample
    print("mesh%-4d buses %3d branches %4d devices %3d  lanes/env %3d  E=%-5d %9.1f us/launch  %.3e transitions/s  converged %.4f  mean iters %.2f"
          "  | oracle (1 core, converging cases) %.2f ms/transition = %.0f transitions/s  -> x%.0f" % (
              n_bus, m.N_bus, m.N_branch, m.N_device, sim.lanes_per_env, E, dt * 1e6, E / dt, float(sim.pfe_converged.double().mean()),
              float(sim.nr_iters.double().mean()), cpu * 1e3, 1 / cpu, E / dt * cpu))


if __name__ == "__main__":
    for n_bus, seed, n_chords in [(64, 10, 24), (100, 12, 12), (200, 13, 30), (300, 14, 40), (513, 15, 40)]:
        for E in (4096, 16384):
            run(n_bus, seed, n_chords, E)
