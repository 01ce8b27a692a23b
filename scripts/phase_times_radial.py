"""Per-wavefront time in each phase of k_radial (config 4: case30, 16 384 transitions with the dump), tuning build:

    ANM_BUILD_TAG=phases ANM_EXTRA_HIPCC_FLAGS=-DANM_PHASE_TIMING python scripts/phase_times_radial.py [cap]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench_case30 import inputs
from gym_anm_amd import networks
from gym_anm_amd.simulator import BatchedSimulator
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sim = BatchedSimulator(networks.synthetic_radial_network(30, 0), 0.25, 100, num_envs=16384, device="cuda:0", impl="radial", max_iter=cap)
pl, pp, ps, qs, soc = inputs(sim, 0)
names = ["inputs", "device maps + bus sums", "Newton loop", "slack, flows, reward", "outputs (dump: hypot / atan2, stores)"]
acc = None
for it in range(8):
    sim.soc.copy_(soc)
    sim.transition(pl, pp, ps, qs)
    torch.cuda.synchronize()
    buf = np.zeros((8, 1024), dtype=np.uint64)
    assert sim.backend.lib.anm_debug_phase_times(buf.ctypes.data_as(ctypes.c_void_p), 1024) == 0
    d = np.diff(buf[:6].astype(np.int64), axis=0)
    if it >= 2:
        acc = d if acc is None else acc + d
tot = buf[5].astype(np.int64) - buf[0].astype(np.int64)
print("case30, 16384 transitions, cap %d: clock ticks of __builtin_readcyclecounter per wavefront (first 1024 wavefronts), mean / median / max" % cap)
for k, nm in enumerate(names):
    x = acc[k] / 6
    print("%-40s mean %9.0f  median %9.0f  max %9.0f" % (nm, x.mean(), np.median(x), x.max()))
print("whole wavefront (last launch): mean %.0f  median %.0f  max %.0f; start skew of the first 1024 wavefronts %d" % (
    tot.mean(), np.median(tot), tot.max(), int(buf[0].max() - buf[0].min())))
