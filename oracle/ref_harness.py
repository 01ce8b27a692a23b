"""Import harness for the *unmodified* reference (robinhenry/gym-anm) in the dev container.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``gym_anm_amd``) may import this
file.  It exists so that ``oracle/make_golden.py`` can run the reference's own Python from
``/root/reference`` (read-only, never copied) and record golden input/output vectors under
``tests/golden/``.  ``/root/reference`` does not exist on the GPU box, so this module is
only ever used here, and only by the fixture generator and by the "reference present"
cross-checks in ``tests/`` (which skip themselves when the reference is absent).

The reference needs three third-party packages that are not installed in this image
(``gymnasium``, ``cvxpy``, ``websocket``/``websocket_server``).  They are replaced by
minimal stand-ins *of the third-party API only* -- no reference code is re-implemented:

* ``gymnasium``: ``Env`` (only ``reset(seed=)`` seeding ``np_random`` the way
  ``gymnasium.utils.seeding.np_random`` does: ``Generator(PCG64(SeedSequence(seed)))``),
  ``spaces.Box`` (low/high/shape/contains/sample) and a no-op ``register``.
* ``cvxpy``: the reference only ever builds ``Problem(Minimize(sum_squares(x - point)),
  [G @ x <= h]).solve()`` with ``x`` in R^2 (``gym_anm/simulator/components/devices.py:299-301``
  and ``:517-519``).  The stand-in solves exactly that problem -- the Euclidean projection of
  ``point`` onto the polygon ``{x : G x <= h}`` -- by exhaustive active-set enumeration
  (candidates: the point, its projection on every edge line, every pairwise vertex; the nearest
  feasible candidate wins).  Real cvxpy hands the QP to an iterative solver with ~1e-5
  accuracy, so golden vectors at this boundary are pinned to the reference's own known-answer
  tests at 1e-5 (``tests/simulator/test_devices.py:269-295,523-562``), not bit-pinned.
* ``websocket`` / ``websocket_server``: rendering only; empty attribute holders.
"""

from __future__ import annotations

import itertools
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("ANM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "gym_anm"))


# --------------------------------------------------------------------------------------
# gymnasium stand-in
# --------------------------------------------------------------------------------------
class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float64, seed=None):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)
        self._rng = np.random.default_rng(seed)

    def contains(self, x):
        x = np.asarray(x)
        if x.shape != self.shape:
            return False
        return bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        return self._rng.uniform(self.low, self.high)


class _Env:
    metadata = {}
    np_random = None

    def reset(self, *, seed=None, options=None):
        if seed is not None or self.np_random is None:
            self.np_random = np.random.default_rng(seed)


def _install_gymnasium():
    gym = types.ModuleType("gymnasium")
    gym.Env = _Env
    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box = _Box
    gym.spaces = spaces
    envs = types.ModuleType("gymnasium.envs")
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register = lambda *a, **k: None
    envs.registration = reg
    gym.envs = envs
    sys.modules.update(
        {
            "gymnasium": gym,
            "gymnasium.spaces": spaces,
            "gymnasium.envs": envs,
            "gymnasium.envs.registration": reg,
        }
    )


# --------------------------------------------------------------------------------------
# cvxpy stand-in (exact 2-D polygon projection)
# --------------------------------------------------------------------------------------
class _Variable:
    __array_ufunc__ = None  # make ``ndarray @ Variable`` defer to __rmatmul__

    def __init__(self, n):
        assert n == 2
        self.value = None

    def __sub__(self, point):
        return _Diff(self, np.asarray(point, dtype=np.float64))

    def __rmatmul__(self, G):
        return _Lin(self, np.asarray(G, dtype=np.float64))


class _Diff:
    def __init__(self, var, point):
        self.var, self.point = var, point


class _Lin:
    def __init__(self, var, G):
        self.var, self.G = var, G

    def __le__(self, h):
        return ("le", self.var, self.G, np.asarray(h, dtype=np.float64))


def exact_projection(point, G, h, slack=1e-12):
    """argmin ||x - point||^2  s.t.  G x <= h   (x in R^2), by active-set enumeration."""
    point = np.asarray(point, dtype=np.float64)
    keep = np.isfinite(h)
    G, h = G[keep], h[keep]

    def feasible(x):
        return bool(np.all(G @ x <= h + slack))

    if feasible(point):
        return point.copy()
    cands = []
    m = len(h)
    for i in range(m):
        g = G[i]
        x = point - g * ((g @ point - h[i]) / (g @ g))
        cands.append(x)
    for i, j in itertools.combinations(range(m), 2):
        A = np.array([G[i], G[j]])
        det = A[0, 0] * A[1, 1] - A[0, 1] * A[1, 0]
        if abs(det) < 1e-14:
            continue
        cands.append(np.linalg.solve(A, np.array([h[i], h[j]])))
    best, best_d = None, np.inf
    for x in cands:
        if feasible(x):
            d = float(np.sum((x - point) ** 2))
            if d < best_d:
                best, best_d = x, d
    if best is None:  # empty polygon: cannot happen for valid device specs
        raise RuntimeError("projection onto an empty polygon")
    return best


class _Problem:
    def __init__(self, objective, constraints):
        self.diff = objective
        (_, self.var, self.G, self.h) = constraints[0]

    def solve(self, *a, **k):
        self.var.value = exact_projection(self.diff.point, self.G, self.h)
        return 0.0


def _install_cvxpy():
    cp = types.ModuleType("cvxpy")
    cp.Variable = _Variable
    cp.sum_squares = lambda d: d
    cp.Minimize = lambda d: d
    cp.Problem = _Problem
    cp.Parameter = lambda *a, **k: None  # only touched by agents/mpc.py at import time
    sys.modules["cvxpy"] = cp


def _install_websocket():
    ws = types.ModuleType("websocket")
    ws.create_connection = None
    wss = types.ModuleType("websocket_server")
    wss.WebsocketServer = object
    sys.modules["websocket"] = ws
    sys.modules["websocket_server"] = wss


_LOADED = False


def load_reference():
    """Install the stand-ins, put the reference on sys.path and import ``gym_anm``."""
    global _LOADED
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    if not _LOADED:
        for name in ("gymnasium", "cvxpy", "websocket", "websocket_server"):
            try:
                __import__(name)
            except Exception:
                {
                    "gymnasium": _install_gymnasium,
                    "cvxpy": _install_cvxpy,
                    "websocket": _install_websocket,
                    "websocket_server": _install_websocket,
                }[name]()
        if REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, REFERENCE_ROOT)
        import logging

        logging.disable(logging.WARNING)
        _LOADED = True
    import gym_anm  # noqa: F401

    return sys.modules["gym_anm"]
