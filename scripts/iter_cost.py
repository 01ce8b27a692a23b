"""Per-iteration cost of the NR loop: tol=0 forces exactly max_iter iterations in every lane."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
prec = sys.argv[2] if len(sys.argv) > 2 else "f64"
impl = sys.argv[3] if len(sys.argv) > 3 else None
for mi in (10, 60):
    env = ANM6EasyVec(num_envs=E, device="cuda:0", seed=1, precision=prec, impl=impl)
    env.check_actions = False
    env.reset(seed=1)
    env.simulator.opts.tol = 0.0
    env.simulator.opts.max_iter = mi
    a = torch.zeros((E, 6), dtype=torch.float64, device="cuda:0")
    a[:, 0] = 10; a[:, 1] = 20
    for _ in range(3): env.step(a); env._term_u8.zero_()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 20
    for _ in range(n):
        env.step(a); env._term_u8.zero_()
    torch.cuda.synchronize()
    print(impl, prec, E, "max_iter", mi, "us/step", (time.perf_counter() - t) / n * 1e6, "iters", int(env.simulator.nr_iters.max()))
