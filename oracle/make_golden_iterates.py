"""Golden Newton-Raphson ITERATES: the reference's x_k and ||F(x_k)||inf after k = 1 ... 12 iterations.

TEST INFRASTRUCTURE ONLY (dev container; needs /root/reference).  Usage::

    python oracle/make_golden_iterates.py        # (re)writes tests/golden/iterates_<net>.npz

Why: ``transition_<net>.npz`` pins where Newton ENDS (final V, iteration count).  Newton is self-correcting,
so that cannot tell an exact Jacobian / linear solve (``_dfdx`` + ``spsolve``,
``gym_anm/simulator/solve_load_flow.py:123-164, 220``) from one that is accurate to 1e-7.  An iterate after k
steps can: an error e in the k-th step shows up in x_k at first order.

How: every case is one ``Simulator.transition`` of the UNMODIFIED reference on seeded inputs (the arguments its
solver receives -- p, q, Y, x_tol -- are captured at the call, ``solve_load_flow.py:45``), then the reference's own
``_newton_raphson_sparse(v_guess, p, q, Y, x_tol, lim_iter=k)`` (``:176-226``) is called for k = 1 ... min(n_iter, 12):
its return values are ``x_k`` and ``diff_k``.  Nothing of the reference is re-implemented here.

Per net (``iterates_<net>.npz``):
  P_load, P_pot, P_set, Q_set, soc0   the transition's inputs (MW / MVAr / p.u.), as in transition_<net>.npz
  bus_p, bus_q [M, N]                 the injections the solver got (slack entries: 0)
  n_iter [M], converged [M], diff [M] of the full solve (x_tol = 1e-5, cap 100)
  n_k [M]                             number of iterates recorded = min(n_iter, 12)
  vm_k, th_k [M, 12, N-1]             the unknowns of x_k as the reference returns them: |V| (may be negative) and theta
                                      (not wrapped) of the non-slack buses; rows k >= n_k: NaN.  V_k = vm_k e^{j th_k}
                                      (``_construct_v_from_guess``, ``:167-173``) is formed by the tests
  diff_k [M, 13]                      ||F(x_k)||inf for k = 0 ... n_k (k = 0: the flat start)
  diverging [M]                       the full solve ran to the cap or produced a NaN
"""

from __future__ import annotations

import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_harness  # noqa: E402

ref_harness.load_reference()

from gym_anm.simulator import Simulator  # noqa: E402
from gym_anm.simulator import solve_load_flow as slf  # noqa: E402

import make_golden as MG  # noqa: E402  (input sampler, device lists; its NR wrapper records (n_iter, diff))
from gym_anm_amd import networks  # noqa: E402  (network *data* only)

warnings.simplefilter("ignore")
GOLDEN = os.path.join(ROOT, "tests", "golden")
K_MAX = 12

_CALLS = []
_inner = slf._newton_raphson_sparse  # (make_golden's recording wrapper around the reference's function)


def _capture(v_guess, p, q, Y, x_tol=1e-10, lim_iter=100):
    _CALLS.append((np.array(v_guess, copy=True), np.array(p, copy=True), np.array(q, copy=True), Y, x_tol))
    return _inner(v_guess, p, q, Y, x_tol=x_tol, lim_iter=lim_iter)


slf._newton_raphson_sparse = _capture


def record(network, delta_t, lamb, ins):
    P_load, P_pot, P_set, Q_set, soc0 = ins
    sim = Simulator(network, delta_t, lamb)
    loads, gens, des = MG._dev_lists(sim)
    setp = sorted(gens + des)
    M, N = P_load.shape[0], sim.N_bus
    out = {k: [] for k in ["bus_p", "bus_q", "n_iter", "converged", "diff", "n_k", "vm_k", "th_k", "diff_k", "diverging"]}
    for m in range(M):
        for k, i in enumerate(des):
            sim.devices[i].soc = float(soc0[m, k])
        _CALLS.clear()
        sim.transition({i: P_load[m, k] for k, i in enumerate(loads)}, {i: P_pot[m, k] for k, i in enumerate(gens)},
                       {i: P_set[m, k] for k, i in enumerate(setp)}, {i: Q_set[m, k] for k, i in enumerate(setp)})
        v0, p, q, Y, x_tol = _CALLS[-1]
        with np.errstate(all="ignore"):
            x, n_iter, diff, conv = MG._orig_nr(v0.copy(), p, q, Y, x_tol=x_tol)
        n_k = min(n_iter, K_MAX)
        vm = np.full((K_MAX, N - 1), np.nan)
        th = np.full((K_MAX, N - 1), np.nan)
        dk = np.full(K_MAX + 1, np.nan)
        dk[0] = np.linalg.norm(slf._f(v0, p, q, Y), np.inf)
        for k in range(1, n_k + 1):
            with np.errstate(all="ignore"):
                xk, it_k, diff_k, _ = MG._orig_nr(v0.copy(), p, q, Y, x_tol=x_tol, lim_iter=k)
            assert it_k == k
            th[k - 1], vm[k - 1] = xk[: N - 1], xk[N - 1:]
            dk[k] = diff_k
        if n_iter <= K_MAX:   # the last recorded iterate IS the solve's result
            assert np.array_equal(xk, x, equal_nan=True) if n_k else True
        out["bus_p"].append(np.concatenate(([0.0], p)))
        out["bus_q"].append(np.concatenate(([0.0], q)))
        out["n_iter"].append(n_iter)
        out["converged"].append(bool(conv) and bool(diff <= x_tol))
        out["diff"].append(diff)
        out["n_k"].append(n_k)
        out["vm_k"].append(vm)
        out["th_k"].append(th)
        out["diff_k"].append(dk)
        out["diverging"].append(bool(n_iter >= 100 or np.isnan(diff)))
    res = {k: np.array(v) for k, v in out.items()}
    res.update(P_load=P_load, P_pot=P_pot, P_set=P_set, Q_set=Q_set, soc0=soc0, delta_t=delta_t, lamb=lamb, x_tol=1e-5)
    return res


def pick(network, delta_t, lamb, seed, n_conv, n_div, load_scale, wild_frac):
    """A seeded pool of inputs; the first `n_div` diverging and the first `n_conv` converging cases of it.  Overloads
    (heavier load_scale rounds) are added until the pool holds enough diverging solves."""
    keep_c, keep_d = [], []
    scale, rnd = load_scale, 0
    while len(keep_c) < n_conv or len(keep_d) < n_div:
        ins = MG.random_inputs(network, delta_t, lamb, 400, seed + 1000 * rnd, wild_frac, scale)
        res = record(network, delta_t, lamb, ins)
        for m in range(len(res["n_iter"])):
            row = tuple(a[m] for a in ins)
            if res["diverging"][m]:
                if len(keep_d) < n_div:
                    keep_d.append(row)
            elif len(keep_c) < n_conv:
                keep_c.append(row)
        rnd += 1
        if len(keep_d) < n_div:
            scale *= 1.6          # not enough collapses at this loading: the next round overloads the feeders
            wild_frac = max(wild_frac, 0.5)
        assert rnd < 12, "no diverging solves found"
    rows = keep_c + keep_d
    return tuple(np.array([r[j] for r in rows]) for j in range(5))


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    p = np.load(os.path.join(GOLDEN, "3bus_tx_params.npz"))
    jobs = [
        ("anm6", networks.anm6_network(), 0.25, 100, 31, 1.0, 0.25),
        ("case30", networks.synthetic_radial_network(30, 0), 0.25, 100, 32, 1.0, 0.1),
        ("3bus", networks.three_bus_loop_network(gen_max=1.5), 0.5, 100, 33, 0.15, 0.2),
        ("3bus_tx2", networks.three_bus_loop_network(p["tap"][2], p["shift"][2], gen_max=1.0), 0.5, 100, 34, 0.1, 0.2),
        ("3bus_tx7", networks.three_bus_loop_network(p["tap"][7], p["shift"][7], gen_max=1.0), 0.5, 100, 35, 0.1, 0.2),
    ]
    for name, net, dt, lamb, seed, ls, wf in jobs:
        ins = pick(net, dt, lamb, seed, 180, 24, ls, wf)
        res = record(net, dt, lamb, ins)
        np.savez_compressed(os.path.join(GOLDEN, "iterates_%s.npz" % name), **res)
        it = res["n_iter"]
        print("%-10s M=%d  iters: %s  diverging=%d  recorded iterates=%d" % (
            name, len(it), {int(k): int((it == k).sum()) for k in np.unique(it)}, int(res["diverging"].sum()), int(res["n_k"].sum())))


if __name__ == "__main__":
    main()
