"""BASELINE.json config 4: synthetic 30-bus radial feeder, 16384 environments: one Simulator.transition
launch per step, both kernel families; also ANM6 through the lane-group family for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd import networks
from gym_anm_amd.simulator import BatchedSimulator
DEV = "cuda:0"

def inputs(sim, seed):
    m, b, E = sim.model, sim.model.baseMVA, sim.num_envs
    g = torch.Generator(device=DEV).manual_seed(seed)
    U = lambda lo, hi: (torch.as_tensor(lo, device=DEV) + (torch.as_tensor(hi, device=DEV) - torch.as_tensor(lo, device=DEV))
                        * torch.rand((E, len(lo)), generator=g, dtype=torch.float64, device=DEV))
    return (U(m.dev_p_min[m.load_idx] * b, 0 * m.dev_p_min[m.load_idx]), U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * b),
            U(m.dev_p_min[m.setp_idx] * b, m.dev_p_max[m.setp_idx] * b), U(m.dev_q_min[m.setp_idx] * b, m.dev_q_max[m.setp_idx] * b),
            U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx]))

def run(name, net, E, impl, prec="f64", n=30, full=True):
    sim = BatchedSimulator(net, 0.25, 100, num_envs=E, device=DEV, impl=impl, precision=prec)
    pl, pp, ps, qs, soc = inputs(sim, 0)
    # the SoC is restored before every launch so that each launch does identical work (otherwise the
    # storage units drift to their limits and the share of diverging solves changes from launch to launch)
    for _ in range(3):
        sim.soc.copy_(soc)
        sim.transition(pl, pp, ps, qs)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        sim.soc.copy_(soc)
        sim.transition(pl, pp, ps, qs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    conv = float(sim.pfe_converged.double().mean())
    print("%-8s %-7s %s E=%-7d %9.1f us/launch  %.3e env-steps/s  converged %.4f  mean iters %.2f" % (
        name, impl, prec, E, dt * 1e6, E / dt, conv, float(sim.nr_iters.double().mean())))

if __name__ == '__main__':
  n30 = networks.synthetic_radial_network(30, 0)
  for E in (16384, 65536, 262144):
    run("case30", n30, E, "radial")
  run("case30", n30, 16384, "radial", "f32")
  run("case30", n30, 16384, "thread", n=5)
  a6 = networks.anm6_network()
  for E in (65536, 262144):
    run("anm6", a6, E, "radial")
    run("anm6", a6, E, "thread")
