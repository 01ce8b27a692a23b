"""When can a Newton solve be declared hopeless?  (evidence for or against a provable early
exit of diverging solves; the headline keeps the reference's cap of 100 either way.)

Random-agent ANM6Easy steps through the CPU oracle's Newton iteration (the reference algorithm,
solve_load_flow.py:176-226, dense LAPACK variant), tol 1e-6, cap 100; for every solve the per-iteration
max |V| and ||F||inf are recorded.  Reported: for each candidate bound B on max|V| (resp. on ||F||inf), the
iteration at which the diverging solves first exceed it, and how many CONVERGING solves ever exceeded it --
a bound is a valid early exit only if that count is provably zero, and the table can only ever refute.

    python scripts/divergence_onset.py [n_steps] > profiles/r02_divergence_onset.txt
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import anm_oracle as O
from gym_anm_amd import networks

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
traces = []  # (converged, n_iter, vmax per iteration, fnorm per iteration)
orig = O.newton_raphson


def traced(p, q, Y, tol=1e-5, max_iter=100, sparse=True):
    m = len(p)
    x = np.array([0.0] * m + [1.0] * m)
    vmax, fn = [], []
    with np.errstate(all="ignore"):
        F = O.nr_f(x, p, q, Y)
        diff = np.linalg.norm(F, np.inf)
        it = 0
        while diff > tol and it < max_iter:
            it += 1
            try:
                x = x - np.linalg.solve(O.nr_jacobian(x, Y, False), F)
            except np.linalg.LinAlgError:
                x = x * np.nan
            F = O.nr_f(x, p, q, Y)
            diff = np.linalg.norm(F, np.inf)
            vmax.append(np.max(np.abs(x[m:])))
            fn.append(diff)
    conv = (not np.isnan(diff)) and diff <= tol
    traces.append((conv, it, np.array(vmax), np.array(fn)))
    return x, it, diff, not np.isnan(diff)


O.newton_raphson = traced
rng = np.random.default_rng(0)
env = O.OracleEnv(networks.anm6_network(), sparse=False, tol=1e-6)
lo = np.array([0, 0, -30, -50, -50, -50.0]); hi = np.array([30, 50, 30, 50, 50, 50.0])
tab = O.anm6easy_tables()


def reset():
    t0 = rng.integers(0, 96)
    s0 = np.zeros(18)
    s0[[1, 3, 5]] = tab[:3, t0]; s0[[2, 4]] = tab[3:, t0]; s0[15:17] = tab[3:, t0]
    s0[9], s0[11], s0[14], s0[17] = rng.uniform(-0.3, 0.3), rng.uniform(-0.5, 0.5), rng.uniform(0, 1), t0
    env.reset_to(s0)


reset()
n_reset = len(traces)
for _ in range(n_steps):
    _, _, term = env.step(rng.uniform(lo, hi))
    if term:
        reset()
conv = [t for t in traces if t[0]]
div = [t for t in traces if not t[0]]
print("# scripts/divergence_onset.py %d: %d Newton solves (tol 1e-6, cap 100, flat start), %d converged, %d not"
      % (n_steps, len(traces), len(conv), len(div)))
its = np.array([t[1] for t in conv])
print("converging solves: iterations  max %d   histogram %s" % (its.max(), np.bincount(its).tolist()))
print("non-converging solves: iterations run  min %d  (cap 100 reached by %d of %d; the others stopped on NaN)"
      % (min(t[1] for t in div), sum(t[1] == 100 for t in div), len(div)))
print()
print("bound on max|V|      converging solves that ever exceed it    diverging solves: first iteration above it (min / median / max / never)")
for B in (1.5, 2.0, 5.0, 10.0, 100.0, 1e4):
    c = sum(bool((t[2] > B).any()) for t in conv)
    first = [int(np.argmax(t[2] > B)) + 1 for t in div if (t[2] > B).any()]
    never = len(div) - len(first)
    print("%10g  %38d    %s" % (B, c, ("%d / %d / %d / %d" % (min(first), np.median(first), max(first), never)) if first else "-"))
print()
print("bound on ||F||inf    converging solves that ever exceed it    diverging solves: first iteration above it (min / median / max / never)")
for B in (1.0, 10.0, 100.0, 1e4, 1e8):
    c = sum(bool((t[3] > B).any()) for t in conv)
    first = [int(np.argmax(t[3] > B)) + 1 for t in div if (t[3] > B).any()]
    never = len(div) - len(first)
    print("%10g  %38d    %s" % (B, c, ("%d / %d / %d / %d" % (min(first), np.median(first), max(first), never)) if first else "-"))
print()
mx = np.array([np.nanmax(t[2]) if len(t[2]) else 0 for t in div])
print("diverging solves: max over the run of max|V|: min %.3g  median %.3g  max %.3g" % (mx.min(), np.median(mx), mx.max()))
back = sum(bool((t[2][-10:] < 2.0).all()) for t in div if len(t[2]) >= 10)
print("diverging solves whose last 10 iterates are all back below max|V| = 2: %d of %d (Newton wanders back and is thrown out again:" % (back, len(div)))
print("a bound on |V| or ||F|| at some iteration says nothing about the iterations that follow, so no early exit is claimed)")
