"""Two-launch step: where the first launch hands over (straggler_after 5 / 6 / 7 / 8), ANM6Easy, random agent, tol 1e-6, reference cap,
one straggler level below 1 M environments (end of round 6, with the shorter lane-group trip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec
DEV = "cuda:0"
for rep in range(2):
    for E in (131072, 524288):
        out = []
        for after in (5, 6, 7, 8):
            env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, autoreset=True, tol=1e-6, max_iter=100, straggler_after=after, straggler_mid=None)
            env.check_actions = False
            env.reset(seed=1)
            g = torch.Generator(device=DEV).manual_seed(0)
            lo, hi = torch.as_tensor(env.action_space.low, device=DEV), torch.as_tensor(env.action_space.high, device=DEV)
            pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=DEV) for _ in range(4)]
            for i in range(8): env.step(pool[i % 4])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(60): env.step(pool[i % 4])
            torch.cuda.synchronize()
            out.append("@%d %6.1f" % (after, (time.perf_counter() - t0) / 60 * 1e6))
            del env
        print("E %7d  " % E + "  ".join(out), flush=True)
