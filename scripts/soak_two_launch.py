"""Soak of the automatic two-launch step with lane-group stragglers: 262 144 autoresetting ANM6Easy environments, 600
steps under a random agent; observations stay finite and inside the Box, rewards inside the clipping range, every
environment keeps cycling, and the same seeds stepped by the one-launch in-wave path give bit-identical results at
checkpoints."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec
DEV = "cuda:0"
# usage: soak_two_launch.py [num_envs [steps]] -- at 1 048 576 environments the automatic step has TWO straggler levels
E = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 600
a = ANM6EasyVec(num_envs=E, device=DEV, seed=4, autoreset=True, tol=1e-6)                          # auto: two-launch here
b = ANM6EasyVec(num_envs=E, device=DEV, seed=4, autoreset=True, tol=1e-6, straggler_after=None)    # in-wave hand-over
assert a._ws is not None and b._ws is None
print("environments", E, "second straggler level at", a._ws.mid_cap)
for e in (a, b):
    e.check_actions = False
    e.reset(seed=4)
g = torch.Generator(device=DEV).manual_seed(1)
lo = torch.as_tensor(a.action_space.low, device=DEV); hi = torch.as_tensor(a.action_space.high, device=DEV)
olo = torch.as_tensor(a.observation_space.low, device=DEV); ohi = torch.as_tensor(a.observation_space.high, device=DEV)
n_term = 0
for t in range(STEPS):
    act = lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=DEV)
    o, r, term, _, _ = a.step(act)
    o2, r2, term2, _, _ = b.step(act)
    if t % 50 == 49:
        assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(term, term2), t
        assert torch.equal(a.state, b.state) and torch.equal(a._reset_count, b._reset_count), t
        assert bool(torch.isfinite(o).all()) and bool(((o >= olo) & (o <= ohi)).all()), t
        assert float(r.min()) >= -100 / (1 - 0.995) - 1e-6 and float(r.max()) <= 1.0 + 1e-9, t   # e_loss may be negative: |e| <= c1
    n_term += int(term.sum())
print("steps", STEPS, "x", E, "terminations", n_term, "min resets per env", int(a._reset_count.min()), "max", int(a._reset_count.max()))
assert n_term > 0.002 * STEPS * E and int(a._reset_count.max()) >= min(5, STEPS // 100)
print("soak ok")
