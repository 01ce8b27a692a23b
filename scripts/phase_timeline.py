"""Timeline of the slowest wavefronts of one headline step (tuning build with -DANM_PHASE_TIMING):

    ANM_BUILD_TAG=phases ANM_EXTRA_HIPCC_FLAGS=-DANM_PHASE_TIMING python scripts/phase_timeline.py [max_iter]

stamps (shader clock of lane 0): 7 kernel entry, 0 rows loaded, 1 inputs formed, 2 Newton loop starts (device maps done), 3 thread-mode
iterations done, 4 lane-group continuation done, 5 flows + reward done, 6 rows stored."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_anm_amd.envs import ANM6EasyVec
E = 65536
mi = int(sys.argv[1]) if len(sys.argv) > 1 else 100
env = ANM6EasyVec(num_envs=E, device="cuda:0", seed=1, tol=1e-6, max_iter=mi, autoreset=True)
env.check_actions = False
env.reset(seed=1)
g = torch.Generator(device="cuda:0").manual_seed(0)
lo = torch.as_tensor(env.action_space.low, device="cuda:0"); hi = torch.as_tensor(env.action_space.high, device="cuda:0")
lib = env.simulator.backend.lib
order = [7, 0, 1, 2, 3, 4, 5, 6]
names = ["entry", "rows loaded", "inputs", "maps done", "thread its", "groups done", "flows", "stored"]
for it in range(6):
    a = lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device="cuda:0")
    env.step(a)
    torch.cuda.synchronize()
    buf = np.zeros((8, 1024), dtype=np.uint64)
    assert lib.anm_debug_phase_times(buf.ctypes.data_as(ctypes.c_void_p), 1024) == 0
    if it < 3:
        continue
    T = buf[order].astype(np.int64)            # [8 stamps, waves]
    # the shader clocks of the XCDs have different origins: durations are per wave, the entry skew is per clock domain
    dom = np.round(T[0] / 1e9).astype(np.int64)
    skew = np.zeros(T.shape[1], dtype=np.int64)
    for d in np.unique(dom):
        m = dom == d
        skew[m] = T[0][m] - T[0][m].min()
    T = T - T[0][None, :]
    end = T[7] + skew
    slow = np.argsort(end)[-3:][::-1]
    med = np.argsort(end)[len(end) // 2]
    print("step %d: longest entry-skew + duration %d ticks; largest skew %d; waves with a continuation: %d; longest continuation %d"
          % (it, end.max(), skew.max(), int((T[5] - T[4] > 2000).sum()), int((T[5] - T[4]).max())))
    for w in list(slow) + [med]:
        print("  wave %4d (skew %d): " % (w, skew[w]) + "  ".join("%s %d" % (nm, T[k, w]) for k, nm in enumerate(names)))
