"""Cycles per part of a Newton trip of the general lane-group kernel (tuning build with -DANM_PHASE_TIMING):

    ANM_BUILD_TAG=phases ANM_EXTRA_HIPCC_FLAGS=-DANM_PHASE_TIMING python scripts/mesh_trip_times.py [mesh30|mesh200|case30|anm6] [max_iter]

Prints, for the wavefront with most trips (the one that holds a diverging solve: alone on its SIMD at the end) and for a
median wavefront, the shader-clock cycles per trip of: publish V + branch products, bus sums + stop test + diagonal,
product steps, sum steps, the tail step, back substitution, update."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_anm_amd import networks
from gym_anm_amd.simulator import BatchedSimulator
which = sys.argv[1] if len(sys.argv) > 1 else "mesh30"
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 100
net, E, ls = {"mesh30": (networks.synthetic_meshed_network(30, 6, 4), 16384, 1.0),
              "case30": (networks.synthetic_radial_network(30, 0), 16384, 1.0),
              "anm6": (networks.anm6_network(), 65536, 1.0),
              "mesh200": (networks.synthetic_meshed_network(200, 13, 30), 4096, 40.0 / 200)}[which]
dev = "cuda:0"
sim = BatchedSimulator(net, 0.25, 100, num_envs=E, device=dev, tol=1e-6, max_iter=cap, impl="mesh")
m, b = sim.model, sim.model.baseMVA
g = torch.Generator(device=dev).manual_seed(0)
def U(lo, hi, scale=1.0):
    lo, hi = torch.as_tensor(lo, device=dev) * scale, torch.as_tensor(hi, device=dev) * scale
    return lo + (hi - lo) * torch.rand((E, lo.numel()), generator=g, dtype=torch.float64, device=dev)
pl = U(m.dev_p_min[m.load_idx] * b, 0 * m.dev_p_min[m.load_idx], ls)
pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * b)
ps = U(m.dev_p_min[m.setp_idx] * b, m.dev_p_max[m.setp_idx] * b, ls)
qs = U(m.dev_q_min[m.setp_idx] * b, m.dev_q_max[m.setp_idx] * b, ls)
soc = U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx])
if ls != 1.0:
    pl[-max(1, E // 1024):] *= 40.0 / ls
for _ in range(3):
    sim.soc.copy_(soc)
    sim.transition(pl, pp, ps, qs)
torch.cuda.synchronize()
lib = sim.backend.lib
W = 16384
buf = np.zeros((8, W), dtype=np.uint64)
assert lib.anm_debug_phase_times(buf.ctypes.data_as(ctypes.c_void_p), W) == 0
A = buf.astype(np.float64)
trips = A[7]
names = ["publish+branch", "bus sums+test", "product steps", "sum steps", "tail step", "back subst.", "update"]
print("%s cap %d: lanes/env %d, converged %.4f, trips per workgroup: max %d, median %d" %
      (which, cap, sim.lanes_per_env, float(sim.pfe_converged.double().mean()), trips.max(), np.median(trips[trips > 0])))
order = np.argsort(trips)
for label, w in (("most trips", order[-1]), ("second", order[-2]), ("median", order[len(order) // 2])):
    n = max(trips[w], 1.0)
    print("  %-10s (workgroup %d, %d trips): " % (label, w, trips[w]) +
          "  ".join("%s %.0f" % (nm, A[k, w] / n) for k, nm in enumerate(names)) + "  | total/trip %.0f" % (A[:7, w].sum() / n))
