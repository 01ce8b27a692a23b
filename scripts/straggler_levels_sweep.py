"""Two-launch step: where the first launch hands over (straggler_after) x where the first straggler launch leaves the
long runners to the second (straggler_mid), ANM6Easy, random agent, tol 1e-6, reference cap."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_anm_amd.envs import ANM6EasyVec

DEV = "cuda:0"
for E in (131072, 524288, 1048576):
    for after in (4, 5, 6):
        for mid in (None, 9, 12):
            env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, autoreset=True, tol=1e-6, max_iter=100, straggler_after=after,
                              straggler_mid=mid)
            env.check_actions = False
            env.reset(seed=1)
            g = torch.Generator(device=DEV).manual_seed(0)
            lo, hi = torch.as_tensor(env.action_space.low, device=DEV), torch.as_tensor(env.action_space.high, device=DEV)
            pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=DEV) for _ in range(4)]
            for i in range(8):
                env.step(pool[i % 4])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(40):
                env.step(pool[i % 4])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 40
            print("E %8d  hand-over after %d  second level at %-4s  %.1f us per step  %.3e env-steps/s" % (E, after, mid, dt * 1e6, E / dt), flush=True)
            del env
