"""CPU oracle: a NumPy/SciPy restatement of gym-anm's simulator step, one environment at a time.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import this module; the product package never does, and it is never the
thing being measured as the product.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against golden
vectors recorded from the unmodified reference (``oracle/make_golden.py`` -> ``tests/golden/``) and
against the reference's own known-answer tests (Y_bus, reward, map_pq tables, SoC update).

Every function cites the reference lines (relative to the reference repo root) it restates.
The per-device QP (``cvxpy`` in the reference, a third-party dependency that is absent here:
cvxpy 1.6.0 / OSQP 0.6.7.post3 per the reference's poetry.lock) is restated as the exact
Euclidean projection onto the constraint polygon; the reference's own tests pin that boundary to
1e-5 (``tests/simulator/test_devices.py:269-295,523-562``).
"""

from __future__ import annotations

import itertools

import numpy as np
from scipy.sparse import csc_matrix, csr_matrix, hstack as shstack, vstack as svstack
from scipy.sparse.linalg import spsolve

LOAD, SLACK, CLASSICAL, RENEWABLE, STORAGE = -1, 0, 1, 2, 3


# ======================================================================================
# Network parsing  (simulator.py:113-181, components/{bus,branch,devices}.py)
# ======================================================================================
class Net:
    """Flat p.u. constants of one network, in the reference's ordering rules: buses and devices by
    ascending ID, branches in file order (simulator.py:148,152,179)."""


def _none(v):
    return v is None or (isinstance(v, float) and np.isnan(v))


def parse_network(network: dict, delta_t: float, lamb: float) -> Net:
    n = Net()
    base = n.baseMVA = network["baseMVA"]
    n.delta_t, n.lamb = delta_t, lamb

    bus = sorted([list(r) for r in network["bus"]], key=lambda r: int(r[0]))
    n.bus_ids = [int(r[0]) for r in bus]
    n.N = len(bus)
    n.bus_slack = np.array([int(r[1]) == 0 for r in bus])
    n.baseKV = np.array([float(r[2]) for r in bus])
    n.v_max = np.array([float(r[3]) for r in bus])  # bus.py:49-51
    n.v_min = np.array([float(r[4]) for r in bus])

    # branches (branch.py:70-151)
    n.br_f, n.br_t, series, shunt, tap, rate = [], [], [], [], [], []
    for r in network["branch"]:
        f, t = int(r[0]), int(r[1])
        rr = np.float64(0.0 if _none(r[2]) else r[2])  # numpy scalars: same complex-division path as the reference
        xx = np.float64(0.0 if _none(r[3]) else r[3])
        bb = np.float64(0.0 if _none(r[4]) else r[4])
        rt = np.inf if _none(r[5]) else float(r[5]) / base
        a = np.float64(1.0 if _none(r[6]) else r[6])
        sh = np.float64(0.0) if _none(r[7]) else np.float64(r[7]) * np.pi / 180
        n.br_f.append(f)
        n.br_t.append(t)
        series.append(1.0 / (rr + 1.0j * xx))  # branch.py:145
        shunt.append(1.0j * bb / 2.0)  # branch.py:148
        tap.append(a * np.exp(1.0j * sh))  # branch.py:151
        rate.append(rt)
    n.series, n.shunt, n.tap, n.rate = map(np.array, (series, shunt, tap, rate))
    n.B = len(n.br_f)

    # nodal admittance matrix (simulator.py:183-199)
    Y = np.zeros((max(n.bus_ids) + 1,) * 2, dtype=np.complex128)
    for k in range(n.B):
        f, t = n.br_f[k], n.br_t[k]
        Y[f, t] = -n.series[k] / np.conjugate(n.tap[k])
        Y[t, f] = -n.series[k] / n.tap[k]
        Y[f, f] += (n.series[k] + n.shunt[k]) / (np.abs(n.tap[k]) ** 2)
        Y[t, t] += n.series[k] + n.shunt[k]
    n.Y = Y

    # devices (devices.py:57-470)
    dev = sorted([list(r) for r in network["device"]], key=lambda r: int(r[0]))
    n.D = len(dev)
    n.dev_ids = [int(r[0]) for r in dev]
    n.dev_bus = [int(r[1]) for r in dev]
    n.dev_type = [int(r[2]) for r in dev]
    f = {k: np.full(n.D, np.nan) for k in
         ["qp", "p_min", "p_max", "q_min", "q_max", "soc_min", "soc_max", "eff"]}  # fmt: skip
    n.tau = np.zeros((n.D, 4))
    n.rho = np.zeros((n.D, 4))
    for k, r in enumerate(dev):
        typ = n.dev_type[k]
        qp, pmax, pmin, qmax, qmin, pplus, pminus, qplus, qminus, socmax, socmin, eff = r[3:15]
        if typ == LOAD:  # devices.py:125-154
            f["qp"][k] = qp
            f["p_max"][k] = 0.0 if _none(pmax) else pmax  # NB: the reference does not rescale PMAX
            f["p_min"][k] = -np.inf if _none(pmin) else pmin / base
            f["q_max"][k] = f["p_max"][k] * qp
            f["q_min"][k] = f["p_min"][k] * qp
            continue
        is_slack = typ == SLACK
        p_max = np.inf if _none(pmax) else pmax / base
        if typ == STORAGE:
            p_min = -np.inf if _none(pmin) else pmin / base
        else:
            p_min = (-np.inf if is_slack else 0.0) if _none(pmin) else pmin / base
        q_max = np.inf if _none(qmax) else qmax / base
        q_min = -np.inf if _none(qmin) else qmin / base
        p_plus = p_max if _none(pplus) else pplus / base
        q_plus = q_max if _none(qplus) else qplus / base
        q_minus = q_min if _none(qminus) else qminus / base
        # devices.py:270-278 / 424-431
        if p_max == p_plus:
            t1 = t2 = 0.0
        else:
            t1 = (q_plus - q_max) / (p_max - p_plus)
            t2 = (q_minus - q_min) / (p_max - p_plus)
        r1 = q_max - t1 * p_plus
        r2 = q_min - t2 * p_plus
        t3 = t4 = r3 = r4 = 0.0
        if typ == STORAGE:  # devices.py:433-442
            p_minus = p_min if _none(pminus) else pminus / base
            if p_min == p_minus:
                t3 = t4 = 0.0
            else:
                t3 = (q_min - q_minus) / (p_minus - p_min)
                t4 = (q_max - q_plus) / (p_minus - p_min)
            r3 = q_min - t3 * p_minus
            r4 = q_max - t4 * p_minus
            f["soc_min"][k] = 0.0 if _none(socmin) else socmin / base
            f["soc_max"][k] = socmax / base
            f["eff"][k] = 1.0 if _none(eff) else eff
        f["p_min"][k], f["p_max"][k], f["q_min"][k], f["q_max"][k] = p_min, p_max, q_min, q_max
        n.tau[k] = [t1, t2, t3, t4]
        n.rho[k] = [r1, r2, r3, r4]
    for k, v in f.items():
        setattr(n, k, v)
    n.loads = [k for k in range(n.D) if n.dev_type[k] == LOAD]
    n.gens = [k for k in range(n.D) if n.dev_type[k] in (CLASSICAL, RENEWABLE)]
    n.des = [k for k in range(n.D) if n.dev_type[k] == STORAGE]
    n.setp = sorted(n.gens + n.des)
    return n


# ======================================================================================
# Device maps  (devices.py:156-167, 181-187, 280-304, 472-545)
# ======================================================================================
def project_polygon(point, G, h, slack=1e-12):
    """argmin ||x-point||^2 s.t. G x <= h in R^2: the problem devices.py:299-301 / 517-519 hands to
    cvxpy, solved exactly by active-set enumeration."""
    point = np.asarray(point, dtype=np.float64)
    keep = np.isfinite(h)
    G, h = G[keep], h[keep]
    if np.all(G @ point <= h + slack):
        return point.copy()
    cands = []
    for i in range(len(h)):
        g = G[i]
        cands.append(point - g * ((g @ point - h[i]) / (g @ g)))
    for i, j in itertools.combinations(range(len(h)), 2):
        A = np.array([G[i], G[j]])
        if abs(A[0, 0] * A[1, 1] - A[0, 1] * A[1, 0]) < 1e-14:
            continue
        cands.append(np.linalg.solve(A, np.array([h[i], h[j]])))
    best, best_d = None, np.inf
    for x in cands:
        if np.all(G @ x <= h + slack):
            d = float(np.sum((x - point) ** 2))
            if d < best_d:
                best, best_d = x, d
    assert best is not None, "empty feasibility polygon"
    return best


def gen_map_pq(n: Net, k, p, q, p_pot):  # devices.py:280-304
    t1, t2 = n.tau[k, 0], n.tau[k, 1]
    G = np.array([[-1, 0], [1, 0], [1, 0], [0, -1], [0, 1], [-t1, 1], [t2, -1]], dtype=float)
    h = np.array([-n.p_min[k], n.p_max[k], p_pot, -n.q_min[k], n.q_max[k], n.rho[k, 0], -n.rho[k, 1]])
    return project_polygon([p, q], G, h)


def des_map_pq(n: Net, k, p, q, soc):  # devices.py:472-522
    t1, t2, t3, t4 = n.tau[k]
    G = np.array(
        [[-1, 0], [1, 0], [0, -1], [0, 1], [-t1, 1], [t2, -1], [t3, -1], [-t4, 1], [-1, 0], [1, 0]], dtype=float
    )
    h = np.array(
        [
            -n.p_min[k], n.p_max[k], -n.q_min[k], n.q_max[k], n.rho[k, 0], -n.rho[k, 1], -n.rho[k, 2], n.rho[k, 3],
            -(soc - n.soc_max[k]) / (n.delta_t * n.eff[k]),
            n.eff[k] * (soc - n.soc_min[k]) / n.delta_t,
        ]
    )  # fmt: skip
    return project_polygon([p, q], G, h)


def update_soc(n: Net, k, soc, p):  # devices.py:524-545
    if p <= 0:
        soc = soc - n.delta_t * n.eff[k] * p
    else:
        soc = soc - n.delta_t * p / n.eff[k]
    return float(np.clip(soc, n.soc_min[k], n.soc_max[k]))


# ======================================================================================
# Newton-Raphson power flow  (solve_load_flow.py:84-226)
# ======================================================================================
def _v_from_x(x):  # solve_load_flow.py:167-173
    m = len(x) // 2
    return np.concatenate(([1 + 0j], x[m:] * np.exp(1j * x[:m])))


def nr_f(x, p, q, Y):  # solve_load_flow.py:84-120
    v = _v_from_x(x)
    mis = (v * np.conj(Y @ v))[1:] - (p + 1j * q)
    return np.concatenate((mis.real, mis.imag))


def nr_jacobian(x, Y, sparse):  # solve_load_flow.py:123-164
    v = _v_from_x(x)
    if sparse:
        idx = np.arange(len(v))
        vd = csr_matrix((v, (idx, idx)))
        vn = csr_matrix((v / abs(v), (idx, idx)))
        idg = csr_matrix((Y * v, (idx, idx)))
        dva = 1j * vd * np.conj(idg - Y * vd)
        dvm = vn * np.conj(idg) + vd * np.conj(Y * vn)
        a, b = dva[1:, 1:], dvm[1:, 1:]
        return svstack([shstack([a.real, b.real]), shstack([a.imag, b.imag])], format="csr")
    Yd = Y
    i = Yd @ v
    dva = 1j * np.diag(v) @ np.conj(np.diag(i) - Yd @ np.diag(v))
    dvm = np.diag(v / abs(v)) @ np.conj(np.diag(i)) + np.diag(v) @ np.conj(Yd @ np.diag(v / abs(v)))
    a, b = dva[1:, 1:], dvm[1:, 1:]
    return np.block([[a.real, b.real], [a.imag, b.imag]])


def newton_raphson(p, q, Y, tol=1e-5, max_iter=100, sparse=True):
    """solve_load_flow.py:176-226 from the flat start of :42-43.  ``sparse=True`` follows the
    reference literally (scipy.sparse Jacobian + spsolve); ``sparse=False`` is the same algorithm
    on dense arrays with LAPACK partial pivoting (fast checker)."""
    m = len(p)
    x = np.array([0.0] * m + [1.0] * m)
    Yop = csc_matrix(Y) if sparse else Y
    with np.errstate(all="ignore"):
        F = nr_f(x, p, q, Yop)
        diff = np.linalg.norm(F, np.inf)
        it = 0
        while diff > tol and it < max_iter:
            it += 1
            J = nr_jacobian(x, Yop, sparse)
            if sparse:
                x = x - spsolve(J, F)
            else:
                try:
                    x = x - np.linalg.solve(J, F)
                except np.linalg.LinAlgError:
                    x = x * np.nan
            F = nr_f(x, p, q, Yop)
            diff = np.linalg.norm(F, np.inf)
    converged = not np.isnan(diff)
    return x, it, diff, converged


# ======================================================================================
# Simulator.transition / reset  (simulator.py:225-293, 464-683)
# ======================================================================================
def compute_reward(n: Net, dev_p, p_pot, v_magn, br_s):
    """``Simulator._compute_reward`` (simulator.py:638-683): (e_loss, penalty) in p.u. energy.
    ``dev_p`` per device, ``p_pot`` per non-slack generator, ``v_magn`` per bus, ``br_s`` per branch."""
    e_loss = 0.0
    for k in range(n.D):
        if n.dev_type[k] in (LOAD, SLACK, CLASSICAL, RENEWABLE):
            e_loss += dev_p[k]
        if n.dev_type[k] == RENEWABLE:
            e_loss += np.maximum(0, p_pot[n.gens.index(k)] - dev_p[k])
    e_loss *= n.delta_t
    pen = 0.0
    for i in range(n.N):
        pen += np.maximum(0, v_magn[i] - n.v_max[i]) + np.maximum(0, n.v_min[i] - v_magn[i])
    for b in range(n.B):
        pen += np.maximum(0, np.abs(br_s[b]) - n.rate[b])
    pen *= n.delta_t * n.lamb
    return e_loss, pen


def transition(n: Net, P_load, P_pot, P_set, Q_set, soc, tol=1e-5, max_iter=100, sparse=True):
    """One ``Simulator.transition``.  Inputs in MW/MVAr ordered like n.loads / n.gens / n.setp,
    ``soc`` (p.u., ordered like n.des) is the SoC before the step.  Returns a dict of p.u. outputs."""
    import warnings

    base = n.baseMVA
    dev_p = np.zeros(n.D)
    dev_q = np.zeros(n.D)
    p_pot = np.zeros(len(n.gens))
    soc_new = np.array(soc, dtype=float).copy()
    for k in range(n.D):  # simulator.py:503-523
        typ = n.dev_type[k]
        if typ == LOAD:
            pp = np.clip(P_load[n.loads.index(k)] / base, n.p_min[k], n.p_max[k])
            dev_p[k], dev_q[k] = pp, pp * n.qp[k]
        elif typ in (CLASSICAL, RENEWABLE):
            g = n.gens.index(k)
            p_pot[g] = np.clip(P_pot[g] / base, n.p_min[k], n.p_max[k])
            s = n.setp.index(k)
            dev_p[k], dev_q[k] = gen_map_pq(n, k, P_set[s] / base, Q_set[s] / base, p_pot[g])
        elif typ == STORAGE:
            s, d = n.setp.index(k), n.des.index(k)
            dev_p[k], dev_q[k] = des_map_pq(n, k, P_set[s] / base, Q_set[s] / base, soc_new[d])
            soc_new[d] = update_soc(n, k, soc_new[d], dev_p[k])
    # simulator.py:539-549
    bus_p = np.zeros(n.N)
    bus_q = np.zeros(n.N)
    for k in range(n.D):
        bus_p[n.dev_bus[k]] += dev_p[k]
        bus_q[n.dev_bus[k]] += dev_q[k]
    # solve_load_flow.py:31-72
    mask = ~n.bus_slack
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, it, diff, conv = newton_raphson(bus_p[mask], bus_q[mask], n.Y, tol, max_iter, sparse)
    stable = bool(conv and diff <= tol)
    with np.errstate(all="ignore"):
        V = _v_from_x(x)
        I = n.Y @ V
        s0 = V[0] * np.conj(I[0])
        slack_bus = int(np.where(n.bus_slack)[0][0])
        bus_p[slack_bus] = s0.real if not np.isnan(s0.real) else np.inf
        bus_q[slack_bus] = s0.imag if not np.isnan(s0.imag) else np.inf
        for k in range(n.D):
            if n.dev_type[k] == SLACK:
                dev_p[k], dev_q[k] = bus_p[n.dev_bus[k]], bus_q[n.dev_bus[k]]
        # branch.py:153-198
        vf, vt = V[n.br_f], V[n.br_t]
        i_from = (n.series + n.shunt) * vf / (np.absolute(n.tap) ** 2) + -n.series * vt / np.conjugate(n.tap)
        i_to = (n.series + n.shunt) * vt + -n.series * vf / n.tap
        s_from = vf * np.conj(i_from)
        s_to = vt * np.conj(i_to)
        br_s = np.sign(s_from.real) * np.maximum(np.abs(s_from), np.abs(s_to))
        e_loss, pen = compute_reward(n, dev_p, p_pot, np.abs(V), br_s)
    return dict(
        dev_p=dev_p, dev_q=dev_q, soc_after=soc_new, p_pot=p_pot, bus_p=bus_p, bus_q=bus_q, V=V, I=I,
        n_iter=it, diff=diff, converged=stable, br_p_from=s_from.real, br_q_from=s_from.imag,
        br_p_to=s_to.real, br_q_to=s_to.imag, br_s=br_s, br_i_from=i_from, br_i_to=i_to,
        reward=-(e_loss + pen), e_loss=e_loss, penalty=pen,
    )  # fmt: skip


def sim_reset(n: Net, init_state, **kw):
    """``Simulator.reset`` (simulator.py:225-293): returns (transition outputs, soc after reset)."""
    D, nd, ng = n.D, len(n.des), len(n.gens)
    P_dev, Q_dev = init_state[:D], init_state[D : 2 * D]
    soc = init_state[2 * D : 2 * D + nd]
    P_max = init_state[2 * D + nd : 2 * D + nd + ng]
    P_load = [P_dev[k] for k in n.loads]
    P_set = [P_dev[k] for k in n.setp]
    Q_set = [Q_dev[k] for k in n.setp]
    soc_pre = [n.soc_min[k] if P_dev[k] <= 0 else n.soc_max[k] for k in n.des]
    out = transition(n, P_load, list(P_max), P_set, Q_set, soc_pre, **kw)
    out["soc_after"] = np.array([soc[d] / n.baseMVA for d in range(nd)])
    return out


# ======================================================================================
# ANMEnv / ANM6Easy  (envs/anm_env.py:235-453, envs/anm6_env/anm6_easy.py)
# ======================================================================================
def anm6easy_tables():
    """The fixed 96-step daily series (anm6_easy.py:77-132): rows P1,P3,P5 (loads) then P2,P4."""

    def day(s1, a12, b12, s2, a23, b23, s3):
        l12 = np.linspace(a12, b12, 7)
        l23 = np.linspace(a23, b23, 7)
        return np.concatenate(
            (s1 * np.ones(25), l12, s2 * np.ones(13), l23, s3 * np.ones(13), l23[::-1], s2 * np.ones(13), l12[::-1],
             s1 * np.ones(4))
        )  # fmt: skip

    P1 = day(-1, -1.5, -4.5, -5, -4.625, -2.375, -2)
    P3 = day(-4, -4.75, -9.25, -10, -11.25, -18.75, -20)
    P5 = day(0, -3.125, -21.875, -25, -21.875, -3.125, 0)
    P2 = day(0, 0.5, 3.5, 4, 7.25, 36.75, 30)
    P4 = day(40, 36.375, 14.625, 11, 14.725, 36.375, 40)
    return np.vstack((P1, P3, P5, P2, P4))


class OracleEnv:
    """``ANMEnv`` + ``ANM6Easy`` semantics on top of :func:`transition` (one environment).

    ``state`` layout (anm_env.py:139-147): [dev_p MW (D), dev_q MVAr (D), des_soc MWh, gen_p_max MW, aux].
    """

    def __init__(self, network, delta_t=0.25, gamma=0.995, lamb=100, costs_clipping=(1, 100),
                 aux_bounds=((0, 95),), tables=None, sparse=True, tol=1e-5, max_iter=100, next_vars=None):  # fmt: skip
        # next_vars: the task hook of anm_env.py:176-191, state -> [P_load (MW).., P_pot (MW).., aux (K)..];
        # default: ANM6Easy's table lookup (anm6_easy.py:54-65), K = 1
        self.next_vars = next_vars
        self.K = len(aux_bounds)
        self.n = parse_network(network, delta_t, lamb)
        self.gamma, self.c1, self.c2 = gamma, costs_clipping[0], costs_clipping[1]
        self.tables = anm6easy_tables() if tables is None else tables
        self.T = self.tables.shape[1]
        self.kw = dict(sparse=sparse, tol=tol, max_iter=max_iter)
        n, b = self.n, self.n.baseMVA
        lo = [n.p_min[k] * b for k in range(n.D)] + [n.q_min[k] * b for k in range(n.D)]
        hi = [n.p_max[k] * b for k in range(n.D)] + [n.q_max[k] * b for k in range(n.D)]
        lo += [n.soc_min[k] * b for k in n.des] + [n.p_min[k] * b for k in n.gens]
        hi += [n.soc_max[k] * b for k in n.des] + [n.q_max[k] * b for k in n.gens]  # simulator.py:430 (sic)
        lo += [a[0] for a in aux_bounds]
        hi += [a[1] for a in aux_bounds]
        self.obs_low, self.obs_high = np.array(lo, dtype=float), np.array(hi, dtype=float)
        self.soc = None

    def _state_vec(self, out, aux):
        b = self.n.baseMVA
        return np.concatenate((out["dev_p"] * b, out["dev_q"] * b, self.soc * b, out["p_pot"] * b, np.atleast_1d(aux)))

    def reset_to(self, init_state):
        """anm_env.py:266-311 for one given init_state draw; returns (obs, converged)."""
        out = sim_reset(self.n, np.asarray(init_state, dtype=float), **self.kw)
        self.soc = out["soc_after"]
        self.terminated = False
        self.state = self._state_vec(out, init_state[len(init_state) - self.K :])
        self.last = out
        return np.clip(self.state, self.obs_low, self.obs_high), out["converged"]

    def load_state(self, state, soc_pu):
        """Continue from a given (non-terminal) state vector and storage SoC in p.u. (test helper: lets a
        test replay a few steps of a long device run; the reference has no counterpart)."""
        self.state = np.asarray(state, dtype=float).copy()
        self.soc = np.atleast_1d(np.asarray(soc_pu, dtype=float)).copy()
        self.terminated = False

    def step(self, action):
        """anm_env.py:333-453 with ANM6Easy.next_vars (anm6_easy.py:54-65)."""
        n = self.n
        if self.terminated:
            return np.zeros_like(self.state), 0.0, True
        nl, ng = len(n.loads), len(n.gens)
        if self.next_vars is None:
            aux = int((self.state[-1] + 1) % self.T)
            P_load, P_pot = self.tables[:nl, aux], self.tables[nl : nl + ng, aux]
        else:
            v = np.asarray(self.next_vars(self.state), dtype=float)
            assert v.size == nl + ng + self.K  # anm_env.py:372-374
            P_load, P_pot, aux = v[:nl], v[nl : nl + ng], v[nl + ng :]
        gen_ids, des_ids = n.gens, n.des
        P_set, Q_set = {}, {}
        for a, k in zip(action[:ng], gen_ids):
            P_set[k] = a
        for a, k in zip(action[ng : 2 * ng], gen_ids):
            Q_set[k] = a
        for a, k in zip(action[2 * ng : 2 * ng + len(des_ids)], des_ids):
            P_set[k] = a
        for a, k in zip(action[2 * ng + len(des_ids) :], des_ids):
            Q_set[k] = a
        out = transition(n, P_load, P_pot, [P_set[k] for k in n.setp], [Q_set[k] for k in n.setp], self.soc, **self.kw)
        self.last = out
        self.soc = out["soc_after"]
        self.terminated = not out["converged"]
        if not self.terminated:
            self.e_loss = np.sign(out["e_loss"]) * np.clip(np.abs(out["e_loss"]), 0, self.c1)
            self.penalty = np.clip(out["penalty"], 0, self.c2)
            r = -(self.e_loss + self.penalty)
            self.state = self._state_vec(out, aux)
            obs = np.clip(self.state, self.obs_low, self.obs_high)
        else:
            r = -self.c2 / (1 - self.gamma)
            self.e_loss, self.penalty = self.c1, self.c2
            self.state = np.zeros_like(self.state)
            obs = np.zeros_like(self.state)
        return obs, r, self.terminated
