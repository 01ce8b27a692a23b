import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec
DEV = "cuda:0"
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for mi in (18, 58):
    env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, straggler_after=8)
    env.check_actions = False
    env.reset(seed=1)
    env.simulator.opts.tol = 0.0
    env.simulator.opts.max_iter = mi
    a = torch.zeros((E, 6), dtype=torch.float64, device=DEV); a[:, 0] = 10; a[:, 1] = 20
    for _ in range(3): env.step(a); env._term_u8.zero_()
    torch.cuda.synchronize()
    t = time.perf_counter(); n = 10
    for _ in range(n): env.step(a); env._term_u8.zero_()
    torch.cuda.synchronize()
    print("E", E, "max_iter", mi, "us/step", (time.perf_counter() - t) / n * 1e6)
