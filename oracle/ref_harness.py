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
* ``cvxpy``: in the simulator the reference only ever builds ``Problem(Minimize(sum_squares(x - point)),
  [G @ x <= h]).solve()`` with ``x`` in R^2 (``gym_anm/simulator/components/devices.py:299-301``
  and ``:517-519``).  The stand-in solves exactly that problem -- the Euclidean projection of
  ``point`` onto the polygon ``{x : G x <= h}`` -- by exhaustive active-set enumeration
  (candidates: the point, its projection on every edge line, every pairwise vertex; the nearest
  feasible candidate wins).  Real cvxpy hands the QP to an iterative solver with ~1e-5
  accuracy, so golden vectors at this boundary are pinned to the reference's own known-answer
  tests at 1e-5 (``tests/simulator/test_devices.py:269-295,523-562``), not bit-pinned.
  The MPC agents (``gym_anm/agents/mpc.py:105-107, 163-319``) state a linear program with cvxpy's modelling
  operators; the stand-in offers exactly those operators (affine expressions of variables and parameters,
  ``==``, ``<=``, ``abs``, ``maximum(0, .)``) and hands the program the reference builds to scipy's HiGHS, with
  the epigraph reformulation cvxpy itself applies.  An LP's optimal VALUE is solver-independent; its minimiser
  need not be unique, so golden vectors of this path pin the value, feasibility, and the minimiser only where
  it is unique (``oracle/make_golden.py``: ``mpc_anm6.npz``).
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
# cvxpy stand-in
#   (a) exact 2-D polygon projection for the device maps (devices.py:299-301, 517-519);
#   (b) a small affine modelling layer for the linear program of agents/mpc.py:105-107, 163-319: Variable,
#       Parameter, +, -, scalar *, ==, <=, abs, maximum(0, .), Minimize, Problem.solve() -> scipy's HiGHS.
#       cvxpy itself reduces maximum(0, abs(e) - c) to epigraph form before calling an LP/conic solver; so does
#       this layer.  Only the third-party API is stood in for: the problem is built by the reference's own code.
# --------------------------------------------------------------------------------------
def _num(x):
    return isinstance(x, (int, float, np.integer, np.floating))


class _Aff:
    """sum of coef * (variable element | parameter element) + const"""

    __array_ufunc__ = None  # numpy scalars defer to the reflected operators

    def __init__(self, vt=None, pt=None, const=0.0):
        self.vt, self.pt, self.const = dict(vt or {}), dict(pt or {}), float(const)

    @staticmethod
    def of(x):
        if isinstance(x, _Aff):
            return x
        if isinstance(x, _Variable):
            assert x.shape == (), "only scalar variables enter expressions whole"
            return x[()]
        if _num(x):
            return _Aff(const=x)
        raise TypeError("cvxpy stand-in: unsupported operand %r" % (x,))

    def _comb(self, other, sign):
        o = _Aff.of(other)
        r = _Aff(self.vt, self.pt, self.const + sign * o.const)
        for k, c in o.vt.items():
            r.vt[k] = r.vt.get(k, 0.0) + sign * c
        for k, c in o.pt.items():
            r.pt[k] = r.pt.get(k, 0.0) + sign * c
        return r

    def __add__(self, o):
        return o.__radd__(self) if isinstance(o, (_Obj, _Hinge)) else self._comb(o, 1.0)

    __radd__ = lambda self, o: self._comb(o, 1.0)
    __sub__ = lambda self, o: self._comb(o, -1.0)
    __rsub__ = lambda self, o: (-self)._comb(o, 1.0)
    __neg__ = lambda self: self * -1.0

    def __mul__(self, k):
        assert _num(k), "cvxpy stand-in: only scalar * expression"
        k = float(k)
        return _Aff({a: c * k for a, c in self.vt.items()}, {a: c * k for a, c in self.pt.items()}, self.const * k)

    __rmul__ = __mul__
    __truediv__ = lambda self, k: self * (1.0 / float(k))
    __eq__ = lambda self, o: _Con("eq", self - o)       # noqa: E731  (constraints, as in cvxpy)
    __le__ = lambda self, o: _Con("le", self - o)
    __ge__ = lambda self, o: _Con("le", _Aff.of(o) - self)
    __hash__ = None

    def numeric(self):  # (variable terms, constant) with the parameters' current values put in
        c = self.const
        for (par, idx), k in self.pt.items():
            c += k * float(np.asarray(par.value, dtype=np.float64)[idx])
        return self.vt, c

    @property
    def value(self):
        vt, c = self.numeric()
        for (var, idx), k in vt.items():
            c += k * float(np.asarray(var.value, dtype=np.float64)[idx])
        return c


class _Con:
    def __init__(self, kind, aff):  # aff == 0  /  aff <= 0
        self.kind, self.aff = kind, aff


class _Abs:
    def __init__(self, aff):
        self.aff = _Aff.of(aff)

    def __sub__(self, c):
        assert _num(c)
        return _AbsMinus(self.aff, float(c))


class _AbsMinus:
    def __init__(self, aff, c):
        self.aff, self.c = aff, c


class _Hinge:
    """max(0, |aff| - c)"""

    __array_ufunc__ = None

    def __init__(self, aff, c):
        self.aff, self.c = aff, c

    def _obj(self):
        return _Obj(_Aff(), [(1.0, self.aff, self.c)])

    __add__ = lambda self, o: self._obj() + o
    __radd__ = __add__
    __mul__ = lambda self, k: self._obj() * k
    __rmul__ = __mul__


class _Obj:
    """affine part + sum of w * max(0, |aff| - c), w >= 0 (convex: minimised through epigraph variables)"""

    __array_ufunc__ = None

    def __init__(self, aff, hinges):
        self.aff, self.hinges = aff, list(hinges)

    def __add__(self, o):
        if isinstance(o, _Hinge):
            o = o._obj()
        if isinstance(o, _Obj):
            return _Obj(self.aff + o.aff, self.hinges + o.hinges)
        return _Obj(self.aff + o, self.hinges)

    __radd__ = __add__

    def __mul__(self, k):
        assert _num(k) and k >= 0
        return _Obj(self.aff * k, [(w * float(k), a, c) for w, a, c in self.hinges])

    __rmul__ = __mul__


def _maximum(a, b):
    assert _num(a) and float(a) == 0.0 and isinstance(b, _AbsMinus), "cvxpy stand-in: maximum(0, abs(e) - c) only"
    return _Hinge(b.aff, b.c)


class _Variable:
    __array_ufunc__ = None  # make ``ndarray @ Variable`` defer to __rmatmul__

    def __init__(self, shape=(), nonneg=False, **kw):
        self.shape = () if shape == () else ((int(shape),) if _num(shape) else tuple(shape))
        self.nonneg = bool(nonneg)
        self.value = None

    def __getitem__(self, idx):
        return _Aff({(self, idx): 1.0})

    # --- (a) the device maps: Variable(2) - point, G @ Variable(2)
    def __sub__(self, o):
        if self.shape == (2,) and isinstance(o, (np.ndarray, list, tuple)):
            return _Diff(self, np.asarray(o, dtype=np.float64))
        return _Aff.of(self) - o

    def __rmatmul__(self, G):
        return _Lin(self, np.asarray(G, dtype=np.float64))

    # --- (b) scalar variables in affine expressions
    __add__ = lambda self, o: _Aff.of(self) + o
    __radd__ = __add__
    __rsub__ = lambda self, o: o - _Aff.of(self)
    __neg__ = lambda self: -_Aff.of(self)
    __mul__ = lambda self, k: _Aff.of(self) * k
    __rmul__ = __mul__
    __truediv__ = lambda self, k: _Aff.of(self) / k


class _Parameter:
    __array_ufunc__ = None

    def __init__(self, shape=(), nonneg=False, **kw):
        self.shape = () if shape == () else ((int(shape),) if _num(shape) else tuple(shape))
        self.value = None

    def __getitem__(self, idx):
        if isinstance(idx, tuple) and any(isinstance(i, slice) for i in idx):
            return _ParamView(self, idx)
        return _Aff(pt={(self, idx): 1.0})


class _ParamView:
    """parameter[:, i]: indexable by the free axis"""

    def __init__(self, par, idx):
        self.par, self.idx = par, idx

    def __getitem__(self, j):
        return _Aff(pt={(self.par, tuple(j if isinstance(i, slice) else i for i in self.idx)): 1.0})


class _Diff:
    def __init__(self, var, point):
        self.var, self.point = var, point


class _Lin:
    def __init__(self, var, G):
        self.var, self.G = var, G

    def __le__(self, h):
        return ("le", self.var, self.G, np.asarray(h, dtype=np.float64))


def solve_lp(objective, constraints):
    """min objective s.t. constraints with scipy.optimize.linprog (HiGHS).  Returns (status, value); the
    variables' .value are set."""
    from scipy.optimize import linprog

    obj = objective if isinstance(objective, _Obj) else (objective._obj() if isinstance(objective, _Hinge) else _Obj(_Aff.of(objective), []))
    cols, lo, hi = {}, [], []

    def col(var, idx):
        k = (id(var), idx)
        if k not in cols:
            cols[k] = (len(lo), var, idx)
            lo.append(0.0 if var.nonneg else -np.inf)
            hi.append(np.inf)
        return cols[k][0]

    rows = []  # (kind, {col: coef}, rhs)
    for con in constraints:
        vt, c = con.aff.numeric()
        rows.append((con.kind, {col(v, i): k for (v, i), k in vt.items()}, -c))
    cvt, c0 = obj.aff.numeric()
    cost = {col(v, i): k for (v, i), k in cvt.items()}
    n_var = len(lo)
    for w, aff, cc in obj.hinges:  # t >= aff - c, t >= -aff - c, t >= 0
        vt, c = aff.numeric()
        a = {col(v, i): k for (v, i), k in vt.items()}
        t = len(lo)
        lo.append(0.0)
        hi.append(np.inf)
        cost[t] = cost.get(t, 0.0) + w
        rows.append(("le", {**a, t: -1.0}, cc - c))
        rows.append(("le", {**{j: -k for j, k in a.items()}, t: -1.0}, cc + c))
    n = len(lo)
    cvec = np.zeros(n)
    for j, k in cost.items():
        cvec[j] = k

    def mat(kind):
        sel = [r for r in rows if r[0] == kind]
        A, bvec = np.zeros((len(sel), n)), np.zeros(len(sel))
        for r, (_, a, rhs) in enumerate(sel):
            for j, k in a.items():
                A[r, j] += k
            bvec[r] = rhs
        return (A, bvec) if len(sel) else (None, None)

    A_ub, b_ub = mat("le")
    A_eq, b_eq = mat("eq")
    res = linprog(cvec, A_ub=A_ub, b_ub=b_ub, A_eq=A_eq, b_eq=b_eq, bounds=list(zip(lo, hi)), method="highs")
    if res.status != 0:
        return {2: "infeasible", 3: "unbounded"}.get(res.status, "solver_error"), None
    vals = {}
    for (_, idx), (j, var, _i) in cols.items():
        vals.setdefault(id(var), (var, {}))[1][idx] = res.x[j]
    for var, d in vals.values():
        if var.shape == ():
            var.value = float(d[()])
        else:
            v = np.zeros(var.shape)
            for idx, x in d.items():
                v[idx] = x
            var.value = v
    solve_lp.last = dict(n_var=n_var, n_aux=n - n_var, n_le=0 if A_ub is None else len(b_ub), n_eq=0 if A_eq is None else len(b_eq))
    return "optimal", float(res.fun + c0)


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
        self.objective, self.constraints = objective, constraints
        self.status, self.value = None, None

    def solve(self, *a, **k):
        if isinstance(self.objective, _Diff):  # (a) projection of a point onto a polygon
            (_, var, G, h) = self.constraints[0]
            var.value = exact_projection(self.objective.point, G, h)
            self.status, self.value = "optimal", 0.0
            return 0.0
        self.status, self.value = solve_lp(self.objective, self.constraints)  # (b)
        return self.value


def _install_cvxpy():
    cp = types.ModuleType("cvxpy")
    cp.Variable = _Variable
    cp.Parameter = _Parameter
    cp.sum_squares = lambda d: d
    cp.Minimize = lambda d: d
    cp.Problem = _Problem
    cp.abs = _Abs
    cp.maximum = _maximum
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
