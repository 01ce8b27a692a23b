"""BASELINE.json config 3: ANM6Easy, 65536 envs, fp64 vs fp32 Newton solve (Jacobian + LU in fp32,
mismatch / stop test / update in fp64) -- deviation and iteration histograms; config 4: the same
for the 30-bus feeder at 16384 envs.  Writes a text report (stdout)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_anm_amd import networks
from gym_anm_amd.simulator import BatchedSimulator

DEV = "cuda:0"


def inputs(sim, seed):
    m, b, E = sim.model, sim.model.baseMVA, sim.num_envs
    g = torch.Generator(device=DEV).manual_seed(seed)
    U = lambda lo, hi: (torch.as_tensor(lo, device=DEV) + (torch.as_tensor(hi, device=DEV) - torch.as_tensor(lo, device=DEV))
                        * torch.rand((E, len(lo)), generator=g, dtype=torch.float64, device=DEV))
    pl = U(m.dev_p_min[m.load_idx] * b, 0 * m.dev_p_min[m.load_idx])
    pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * b)
    ps = U(m.dev_p_min[m.setp_idx] * b, m.dev_p_max[m.setp_idx] * b)
    qs = U(m.dev_q_min[m.setp_idx] * b, m.dev_q_max[m.setp_idx] * b)
    soc = U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx])
    return pl, pp, ps, qs, soc


def run(net, E, tol, name, impl=None):
    out = {}
    for prec in ("f64", "f32"):
        sim = BatchedSimulator(net, 0.25, 100, num_envs=E, device=DEV, tol=tol, precision=prec, impl=impl)
        family = sim.impl
        pl, pp, ps, qs, soc = inputs(sim, 0)
        sim.soc.copy_(soc)
        sim.transition(pl, pp, ps, qs)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            sim.soc.copy_(soc)
            st, r, e, p, conv = sim.transition(pl, pp, ps, qs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 10
        out[prec] = dict(vm=st.tensor("bus_v_magn", "pu").clone(), va=st.tensor("bus_v_ang", "rad").clone(),
                         bs=st.tensor("branch_s", "pu").clone(), bp=st.tensor("branch_p", "pu").clone(),
                         conv=conv.clone(), it=sim.nr_iters.clone(), r=r.clone(), dt=dt)
    a, b = out["f64"], out["f32"]
    both = a["conv"] & b["conv"]
    print("== %s: %d envs, tol %.0e, full-state transition kernel, %s family" % (name, E, tol, family))
    print("   converged: f64 %d  f32 %d  flag mismatches %d" % (int(a["conv"].sum()), int(b["conv"].sum()), int((a["conv"] != b["conv"]).sum())))
    for k, lab in (("vm", "|V| (p.u.)"), ("va", "theta (rad)"), ("bp", "branch P (p.u.)"), ("bs", "branch S (p.u.)")):
        d = (a[k] - b[k]).abs()[both]
        print("   max |f64 - f32| %-16s %.3e   (mean %.3e)" % (lab, float(d.max()), float(d.mean())))
    d = (a["r"] - b["r"]).abs()[both]
    print("   max |d reward| %.3e" % float(d.max()))
    for prec in ("f64", "f32"):
        it = out[prec]["it"][both].cpu().numpy()
        print("   %s iterations histogram (converged): %s   kernel+copy %.1f us/launch" % (prec, dict(zip(*np.unique(it, return_counts=True))), out[prec]["dt"] * 1e6))


for impl in ("thread", "radial", "mesh"):   # the three kernel families (round 2: lane-group continuation, mesh family)
    run(networks.anm6_network(), 65536, 1e-5, "ANM6 (reference stop rule)", impl)
    run(networks.anm6_network(), 65536, 1e-6, "ANM6 (metric stop rule)", impl)
run(networks.anm6_network(), 65536, 1e-9, "ANM6 (tight)", "thread")
for impl in ("radial", "mesh"):
    run(networks.synthetic_radial_network(30, 0), 16384, 1e-5, "30-bus radial feeder", impl)
    run(networks.synthetic_radial_network(30, 0), 16384, 1e-9, "30-bus radial feeder (tight)", impl)
run(networks.synthetic_meshed_network(30, 0, 4), 16384, 1e-5, "meshed 30-bus network", "mesh")
run(networks.synthetic_meshed_network(30, 0, 4), 16384, 1e-9, "meshed 30-bus network (tight)", "mesh")
