"""Average per-wave time spent in each phase of k_step (tuning build with -DANM_PHASE_TIMING).

    ANM_BUILD_TAG=phases ANM_EXTRA_HIPCC_FLAGS=-DANM_PHASE_TIMING python scripts/phase_times.py [max_iter]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_anm_amd.envs import ANM6EasyVec
E = 65536
mi = int(sys.argv[1]) if len(sys.argv) > 1 else 20
env = ANM6EasyVec(num_envs=E, device="cuda:0", seed=1, tol=1e-6, max_iter=mi, autoreset=True)
env.check_actions = False
env.reset(seed=1)
g = torch.Generator(device="cuda:0").manual_seed(0)
lo = torch.as_tensor(env.action_space.low, device="cuda:0"); hi = torch.as_tensor(env.action_space.high, device="cuda:0")
acc = None
names = ["inputs -> exo/action", "device maps+bus sums", "Newton loop", "(loop exit)", "flows+reward", "finish+stores"]
lib = env.simulator.backend.lib
for it in range(12):
    a = lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device="cuda:0")
    env.step(a)
    torch.cuda.synchronize()
    buf = np.zeros((8, 1024), dtype=np.uint64)
    rc = lib.anm_debug_phase_times(buf.ctypes.data_as(ctypes.c_void_p), 1024)
    assert rc == 0
    d = np.diff(buf[:7].astype(np.int64), axis=0)  # [6, waves]
    if it >= 2:
        acc = d if acc is None else acc + d
        pre = (buf[0].astype(np.int64) - buf[7].astype(np.int64))
        acc_pre = pre if it == 2 else acc_pre + pre
        n = it - 1
tot = (buf[6].astype(np.int64) - buf[0].astype(np.int64))
print("max_iter", mi, "clock ticks (s_memtime, 100 MHz?) per wave; mean / median / max over waves, averaged over steps")
x = acc_pre / n
print("%-22s mean %9.0f  median %9.0f  max %9.0f" % ("entry -> rows loaded", x.mean(), np.median(x), x.max()))
for k, nm in enumerate(names):
    x = acc[k] / n
    print("%-22s mean %9.0f  median %9.0f  max %9.0f" % (nm, x.mean(), np.median(x), x.max()))
print("total last step: mean %.0f max %.0f; start skew (max-min of phase 0): %d" % (tot.mean(), tot.max(), int(buf[0].max() - buf[0].min())))
