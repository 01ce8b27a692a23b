"""Step time of ANM6EasyVec (tol 1e-6, autoreset, random agent) against the in-wave straggler hand-over
point (anm_solver_opts.handoff_after; None = stay in the lane) and against the two-launch mode, for several
batch sizes and both iteration caps.  Usage: python scripts/handoff_sweep.py [E ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec
DEV = "cuda:0"
sizes = [int(x) for x in sys.argv[1:]] or [65536, 4096, 131072, 262144, 1048576]
def run(E, cap, n=80, **kw):
    env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, autoreset=True, tol=1e-6, max_iter=cap, **kw)
    env.check_actions = False
    env.reset(seed=1)
    g = torch.Generator(device=DEV).manual_seed(0)
    lo = torch.as_tensor(env.action_space.low, device=DEV); hi = torch.as_tensor(env.action_space.high, device=DEV)
    pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=DEV) for _ in range(8)]
    for i in range(10): env.step(pool[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): env.step(pool[i % 8])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for E in sizes:
    for cap in (100, 20):
        cols = []
        for name, kw in (("lane-only", dict(handoff_after=None, straggler_after=None)),
                         ("handoff@2", dict(handoff_after=2, straggler_after=None)),
                         ("handoff@4", dict(handoff_after=4, straggler_after=None)),
                         ("handoff@6", dict(handoff_after=6, straggler_after=None)),
                         ("handoff@8", dict(handoff_after=8, straggler_after=None)),
                         ("handoff@10", dict(handoff_after=10, straggler_after=None)),
                         ("two-launch@6 threads", dict(handoff_after=None, straggler_after=6)),
                         ("two-launch@6 groups", dict(handoff_after=6, straggler_after=6)),
                         ("auto", dict())):
            cols.append("%s %7.1f" % (name, run(E, cap, **kw)))
        print("E=%8d cap=%3d us/step:  %s" % (E, cap, "   ".join(cols)), flush=True)
