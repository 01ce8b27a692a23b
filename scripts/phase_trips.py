"""Newton-loop time per wave as a function of the (forced) number of trips: tuning build with
-DANM_PHASE_TIMING, tol = 0 so that every lane runs exactly max_iter updates.

    ANM_BUILD_TAG=phases ANM_EXTRA_HIPCC_FLAGS=-DANM_PHASE_TIMING python scripts/phase_trips.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_anm_amd.envs import ANM6EasyVec
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = ANM6EasyVec(num_envs=E, device="cuda:0", seed=1, tol=1e-6, max_iter=100, autoreset=True)
env.check_actions = False
env.reset(seed=1)
lib = env.simulator.backend.lib
a = torch.zeros((E, 6), dtype=torch.float64, device="cuda:0"); a[:, 0] = 10; a[:, 1] = 20
env.simulator.opts.tol = 0.0
for mi in (1, 2, 3, 4, 6, 8, 12, 16):
    env.simulator.opts.max_iter = mi
    acc = []
    for it in range(6):
        env.step(a); env._term_u8.zero_()
        torch.cuda.synchronize()
        buf = np.zeros((8, 1024), dtype=np.uint64)
        assert lib.anm_debug_phase_times(buf.ctypes.data_as(ctypes.c_void_p), 1024) == 0
        if it >= 2:
            acc.append((buf[3].astype(np.int64) - buf[2].astype(np.int64)))
    x = np.mean(acc, axis=0)
    print("updates %2d: loop cycles per wave  mean %8.0f  median %8.0f  min %8.0f  max %8.0f" % (mi, x.mean(), np.median(x), x.min(), x.max()))
