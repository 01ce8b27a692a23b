"""Batched MPC DC-OPF policy: the reference's ``gym_anm/agents/mpc.py`` for ``num_envs`` environments at once.

The reference builds, per environment and per call of ``act()``, an N-stage DC optimal power flow with
cvxpy (``mpc.py:163-319``) and hands it to a generic convex solver.  The problem is a linear program whose
constraint MATRIX depends only on the network and the horizon; what changes from call to call and from
environment to environment are right-hand sides: the load / generation forecasts and the storage state
of charge (``mpc.py:390-417``).  So here

* :class:`DCOPFProgram` assembles the LP once (NumPy), in the form  min q.x  s.t.  l <= A x <= u  with
  ``l, u`` affine in the per-environment parameters;
* :class:`BatchedADMM` solves all environments together with an OSQP-style ADMM whose linear system
  ``(sigma I + A' diag(rho) A)`` is inverted ONCE on the host: an iteration is two dense fp64 GEMMs over the
  whole batch (rocBLAS on the MI355X: ``[num_envs, n+m] x [n+m, n]`` and ``[num_envs, n] x [n, m]``) plus an
  element-wise projection / dual update; iterates are warm-started from the previous call;
* :class:`MPCAgent` / :class:`MPCAgentConstant` / :class:`MPCAgentPerfect` keep the reference's names,
  constructor arguments and ``act(env)`` contract (``mpc.py:321-346``, ``mpc_constant.py``, ``mpc_perfect.py``)
  on a :class:`~gym_anm_amd.envs.anm_env.BatchedANMEnv`.

All quantities are per-unit, as in the reference.  An LP optimum is not unique in general (curtailment vs
storage, charge vs discharge splits), so two exact solvers may return different optimal actions; what is
comparable is the optimal objective and the feasibility of the action (tests/test_mpc.py, against
``scipy.optimize.linprog`` on an independently assembled copy of the reference's program).
"""

from __future__ import annotations

import numpy as np
import torch

from ..model import CLASSICAL, LOAD, RENEWABLE, SLACK, STORAGE

INF = np.inf


class DCOPFProgram:
    """The N-stage DC-OPF of ``mpc.py:163-319`` as  min q.x  s.t.  l0 + Lp p <= A x <= u0 + Up p.

    Variables per stage i (in this order): bus angles theta (n_bus), device injections P_dev (n_dev),
    storage charge p_c and discharge p_d (n_des each, both >= 0, ``mpc.py:282-291``) and one epigraph variable
    s per branch for ``max(0, |P_ij| - beta rate)`` (``mpc.py:308-311``).  Parameters p (per environment):
    load forecasts [n_load x N] (stage-major), generation forecasts [n_gen x N], initial SoC [n_des].
    """

    def __init__(self, model, gamma, safety_margin, planning_steps):
        m, N = model, int(planning_steps)
        self.N = N
        nb, nd, nbr = m.N_bus, m.N_device, m.N_branch
        loads, gens, des = list(m.load_idx), list(m.gen_idx), list(m.des_idx)
        nl, ng, ns = len(loads), len(gens), len(des)
        self.nl, self.ng, self.ns = nl, ng, ns
        B = np.asarray(m.Y_bus).imag  # mpc.py:113: B_bus = Im(Y_bus)
        per = nb + nd + 2 * ns + nbr
        self.per, self.n = per, per * N
        TH, PD, PC, PDIS, SB = 0, nb, nb + nd, nb + nd + ns, nb + nd + 2 * ns
        self.off = dict(theta=TH, p_dev=PD, p_c=PC, p_d=PDIS, s=SB)
        self.n_param = (nl + ng) * N + ns
        rows, lo, up, Lp, Up = [], [], [], [], []

        def add(coefs, l, u, lp=None, up_=None):
            r = np.zeros(self.n)
            for k, v in coefs:
                r[k] += v
            rows.append(r)
            lo.append(l)
            up.append(u)
            Lp.append(np.zeros(self.n_param) if lp is None else lp)
            Up.append(np.zeros(self.n_param) if up_ is None else up_)

        def pvec(idx, val=1.0):
            v = np.zeros(self.n_param)
            v[idx] = val
            return v

        q = np.zeros(self.n)
        dt = m.delta_t
        slack_theta = m.slack_dev  # mpc.py:302 indexes the ANGLES with the slack DEVICE's position (sic)
        for i in range(N):
            o = i * per
            # P_bus[i] = sum over the branches at i of B_ij (theta_i - theta_j)   (mpc.py:232-245)
            for b in range(nb):
                c = []
                for f, t in zip(m.br_f, m.br_t):
                    if f == b:
                        c += [(o + TH + f, B[f, t]), (o + TH + t, -B[f, t])]
                    elif t == b:
                        c += [(o + TH + t, B[t, f]), (o + TH + f, -B[t, f])]
                c += [(o + PD + k, -1.0) for k in range(nd) if m.dev_bus[k] == b]
                add(c, 0.0, 0.0)
            # loads follow their forecast (mpc.py:247-251)
            for j, k in enumerate(loads):
                pv = pvec(i * nl + j)
                add([(o + PD + k, 1.0)], 0.0, 0.0, pv, pv)
            # non-slack generators: P_min <= P <= min(P_max, forecast)  (mpc.py:253-258 and 267-271: two upper
            # bounds, kept as two rows)
            for j, k in enumerate(gens):
                add([(o + PD + k, 1.0)], m.dev_p_min[k], m.dev_p_max[k])
                add([(o + PD + k, 1.0)], -INF, 0.0, None, pvec(nl * N + i * ng + j))
            # storage: box, split into charge / discharge, SoC window (mpc.py:260-265, 273-291)
            for j, k in enumerate(des):
                add([(o + PD + k, 1.0)], m.dev_p_min[k], m.dev_p_max[k])
                add([(o + PD + k, 1.0), (o + PDIS + j, -1.0), (o + PC + j, 1.0)], 0.0, 0.0)
                c = []
                for ii in range(i + 1):  # soc_i = soc_0 + sum_{stages <= i} (p_c dt eff - p_d dt / eff)
                    c += [(ii * per + PC + j, dt * m.dev_eff[k]), (ii * per + PDIS + j, -dt / m.dev_eff[k])]
                ps = pvec((nl + ng) * N + j, -1.0)
                add(c, m.dev_soc_min[k], m.dev_soc_max[k], ps, ps)
                add([(o + PC + j, 1.0)], 0.0, INF)
                add([(o + PDIS + j, 1.0)], 0.0, INF)
            # angles (mpc.py:293-302)
            for b in range(nb):
                add([(o + TH + b, 1.0)], -np.pi, np.pi)
            add([(o + TH + slack_theta, 1.0)], 0.0, 0.0)
            # epigraph of max(0, |B_ft (theta_f - theta_t)| - beta rate)  (mpc.py:308-311, 148-155)
            for e, (f, t) in enumerate(zip(m.br_f, m.br_t)):
                lim = safety_margin * m.br_rate[e]
                add([(o + TH + f, B[f, t]), (o + TH + t, -B[f, t]), (o + SB + e, -1.0)], -INF, lim)
                add([(o + TH + f, -B[f, t]), (o + TH + t, B[f, t]), (o + SB + e, -1.0)], -INF, lim)
                add([(o + SB + e, 1.0)], 0.0, INF)
            # objective (mpc.py:304-313): generation of every non-renewable generator (slack included) + lambda x overloads
            w = gamma**i
            for k in range(nd):
                if m.dev_type[k] in (SLACK, CLASSICAL):
                    q[o + PD + k] += w
            for e in range(nbr):
                q[o + SB + e] += w * m.lamb
        self.A = np.array(rows)
        self.l0, self.u0 = np.array(lo, float), np.array(up, float)
        self.Lp, self.Up = np.array(Lp), np.array(Up)
        self.q = q
        self.m = len(rows)
        self.gens, self.des, self.loads = gens, des, loads
        self.base = float(m.baseMVA)

    def bounds(self, params):
        """[E, n_param] -> (l, u), each [E, m]."""
        P = torch.as_tensor(params, dtype=torch.float64)
        dev = P.device
        l = torch.as_tensor(self.l0, device=dev) + P @ torch.as_tensor(self.Lp.T, device=dev)
        u = torch.as_tensor(self.u0, device=dev) + P @ torch.as_tensor(self.Up.T, device=dev)
        return l, u

    def objective(self, x):
        return x @ torch.as_tensor(self.q, dtype=torch.float64, device=x.device)


class BatchedADMM:
    """OSQP-style ADMM for  min q.x, l <= A x <= u  over a batch sharing q and A (Stellato et al., OSQP, alg. 1 with
    P = 0).  Equality rows get a 1e3 times larger penalty; the problem is Ruiz-equilibrated; rho is re-tuned from
    the residual ratio every ``adapt_every`` iterations (a refactorisation = one small host inverse)."""

    def __init__(self, A, q, eq_mask, device, sigma=1e-6, rho=0.1, alpha=1.6, backend=None):
        self.device = torch.device(device)
        self.backend = backend  # library with anm_admm_update_f64 (the fused non-GEMM part of an iteration)
        A = np.asarray(A, float)
        self.m, self.n = A.shape
        # Ruiz equilibration of [0 A'; A 0]
        D, Ee = np.ones(self.n), np.ones(self.m)
        As = A.copy()
        for _ in range(15):
            cn = np.sqrt(np.maximum(np.abs(As).max(axis=0), 1e-8))
            rn = np.sqrt(np.maximum(np.abs(As).max(axis=1), 1e-8))
            As = As / rn[:, None] / cn[None, :]
            D, Ee = D / cn, Ee / rn
        self.D, self.E = D, Ee
        qs = D * q
        self.c = 1.0 / max(np.abs(qs).max(), 1e-8)
        self.As, self.qs = As, qs * self.c
        self.sigma, self.alpha = sigma, alpha
        self.eq = np.asarray(eq_mask, bool)
        self.rho0 = rho
        self._factor(rho)
        t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=self.device)
        self.tD, self.tE, self.tq = t(D), t(Ee), t(self.qs)
        self.x = self.z = self.y = None

    def _factor(self, rho):
        self.rho = rho
        rv = np.where(self.eq, 1e3 * rho, rho)
        M = self.sigma * np.eye(self.n) + self.As.T @ (rv[:, None] * self.As)
        Minv = np.linalg.inv(M)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=self.device)
        # x~ = [x, w] @ K1 - const,  w = rho z - y:   x~ = Minv (sigma x - q + A' w)
        self.K1 = t(np.vstack([self.sigma * Minv, self.As @ Minv]))     # [(n + m), n]
        self.kq = t(Minv @ self.qs)                                       # [n]
        self.At = t(self.As.T)                                            # [n, m]
        self.rv = t(rv)

    def solve(self, l, u, max_iter=4000, eps=1e-6, check_every=25, adapt_every=100, warm=True):
        """l, u: [E, m] (unscaled).  Returns x [E, n] (unscaled), info."""
        E_ = l.shape[0]
        ls, us = l * self.tE, u * self.tE
        if not warm or self.x is None or self.x.shape[0] != E_:
            self.x = torch.zeros((E_, self.n), dtype=torch.float64, device=self.device)
            self.z = torch.zeros((E_, self.m), dtype=torch.float64, device=self.device)
            self.y = torch.zeros_like(self.z)
        x, z, y = self.x, self.z, self.y
        z = torch.minimum(torch.maximum(z, ls), us)
        info = {"iters": max_iter, "refactor": 0}
        fused = self.backend is not None
        if fused:
            import ctypes as C

            x, z, y = x.contiguous().clone(), z.contiguous().clone(), y.contiguous().clone()
            ls, us = ls.contiguous(), us.contiguous()
            xw = torch.cat((x, self.rv * z - y), dim=1).contiguous()
            stream = None if self.device.type != "cuda" else C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            upd = self.backend.lib.anm_admm_update_f64
        for it in range(1, max_iter + 1):
            if fused:  # two GEMMs (rocBLAS) + one fused HIP kernel per iteration
                xt = torch.addmm(-self.kq.expand(E_, -1), xw, self.K1)
                zt = xt @ self.At
                rc = upd(E_, self.n, self.m, self.alpha, xt.data_ptr(), zt.data_ptr(), ls.data_ptr(), us.data_ptr(),
                         self.rv.data_ptr(), x.data_ptr(), z.data_ptr(), y.data_ptr(), xw.data_ptr(), stream)
                if rc != 0:
                    self.backend.check(rc, "anm_admm_update_f64")
            else:
                w = self.rv * z - y
                xt = torch.cat((x, w), dim=1) @ self.K1 - self.kq
                zt = xt @ self.At
                x = self.alpha * xt + (1 - self.alpha) * x
                zh = self.alpha * zt + (1 - self.alpha) * z
                zn = torch.minimum(torch.maximum(zh + y / self.rv, ls), us)
                y = y + self.rv * (zh - zn)
                z = zn
            if it % check_every == 0 or it == max_iter:
                Ax = x @ self.At
                rp = ((Ax - z).abs() / self.tE).amax(dim=1)                       # unscaled primal residual
                rd = ((y @ self.At.T + self.tq).abs() / self.tD).amax(dim=1) / self.c
                sp = torch.maximum((Ax.abs() / self.tE).amax(dim=1), (z.abs() / self.tE).amax(dim=1))
                sd = torch.maximum(((y @ self.At.T).abs() / self.tD).amax(dim=1) / self.c, torch.as_tensor(np.abs(self.qs / self.D).max() / self.c, device=self.device))
                ok = (rp <= eps * (1 + sp)) & (rd <= eps * (1 + sd))
                info.update(r_prim=float(rp.max()), r_dual=float(rd.max()), iters=it)
                if bool(ok.all()):
                    break
                if it % adapt_every == 0 and info["refactor"] < 12:
                    ratio = float(torch.sqrt((rp / (sp + 1e-12)).median() / ((rd / (sd + 1e-12)).median() + 1e-18)))
                    if ratio > 5 or ratio < 0.2:
                        new_rho = float(np.clip(self.rho * ratio, 1e-6, 1e6))
                        # (y is the unscaled-by-rho dual: keep it; z/w follow from rv at the next iteration)
                        self._factor(new_rho)
                        info["refactor"] += 1
                        if fused:
                            xw = torch.cat((x, self.rv * z - y), dim=1).contiguous()
        self.x, self.z, self.y = x, z, y
        return x * self.tD, info


class MPCAgent:
    """``MPCAgent(simulator, action_space, gamma, safety_margin=0.9, planning_steps=1)`` (``mpc.py:33-122``):
    ``act(env)`` returns the ``[num_envs, action_dim]`` tensor of the first-stage set-points (MW; reactive
    set-points 0, ``mpc.py:383-388``), clipped to the action space."""

    def __init__(self, simulator, action_space, gamma, safety_margin=0.9, planning_steps=1, eps=1e-6, max_iter=4000):
        self.simulator = simulator
        self.action_space = action_space
        self.gamma, self.safety_margin, self.planning_steps = gamma, safety_margin, int(planning_steps)
        self.baseMVA, self.lamb, self.delta_t = simulator.baseMVA, simulator.lamb, simulator.delta_t
        m = simulator.model
        self.program = DCOPFProgram(m, gamma, safety_margin, planning_steps)
        self.load_ids = [m.dev_ids[k] for k in m.load_idx]
        self.non_slack_gen_ids = [m.dev_ids[k] for k in m.gen_idx]
        self.des_ids = [m.dev_ids[k] for k in m.des_idx]
        self.device = simulator.device
        pr = self.program
        self.solver = BatchedADMM(pr.A, pr.q, pr.l0 == pr.u0, self.device, backend=simulator.backend)
        self.eps, self.max_iter = eps, max_iter
        self.last_info = None
        self._lo = torch.as_tensor(np.asarray(action_space.low, float), device=self.device)
        self._hi = torch.as_tensor(np.asarray(action_space.high, float), device=self.device)

    def forecast(self, env):
        """-> (P_load_forecast [E, n_load, N], P_gen_forecast [E, n_gen, N]) in p.u.  (``mpc.py:348-372``)."""
        raise NotImplementedError()

    def _soc(self, env):
        return env.simulator.soc  # p.u., [E, n_des]  (mpc.py:417 reads des_soc in pu)

    def solve(self, P_load_forecast, P_gen_forecast, soc):
        pr = self.program
        E_ = soc.shape[0]
        N = pr.N
        pl = torch.as_tensor(P_load_forecast, dtype=torch.float64, device=self.device).reshape(E_, pr.nl, N)
        pg = torch.as_tensor(P_gen_forecast, dtype=torch.float64, device=self.device).reshape(E_, pr.ng, N)
        params = torch.cat((pl.permute(0, 2, 1).reshape(E_, -1), pg.permute(0, 2, 1).reshape(E_, -1),
                            torch.as_tensor(soc, dtype=torch.float64, device=self.device).reshape(E_, pr.ns)), dim=1)
        l, u = pr.bounds(params)
        x, info = self.solver.solve(l, u, max_iter=self.max_iter, eps=self.eps)
        self.last_info, self.last_x, self.last_bounds = info, x, (l, u)
        return x

    def act(self, env):
        """``env``: a batched environment -> ``[num_envs, action_dim]`` tensor; or one of the NumPy-facing
        single-environment classes (``ANMEnv``, ``ANM6``, ``ANM6Easy``) -> 1-D NumPy action, as in the reference's
        ``examples/mpc_*.py``."""
        if hasattr(env, "vec"):
            return self.act(env.vec)[0].cpu().numpy()
        pl, pg = self.forecast(env)
        x = self.solve(pl, pg, self._soc(env))
        pr = self.program
        pd0 = x[:, pr.off["p_dev"] : pr.off["p_dev"] + env.simulator.N_device] * self.baseMVA
        P_gen = pd0[:, pr.gens]
        P_des = pd0[:, pr.des]
        a = torch.cat((P_gen, torch.zeros_like(P_gen), P_des, torch.zeros_like(P_des)), dim=1)
        return torch.minimum(torch.maximum(a, self._lo), self._hi)  # mpc.py:341-344


class MPCAgentConstant(MPCAgent):
    """Constant forecasts: the current loads and generation potentials persist over the horizon
    (``mpc_constant.py:24-35``).  They are read from the state vector: dev_p of the loads and gen_p_max, MW."""

    def forecast(self, env):
        m = env.simulator.model
        s = env.state
        D, nd = m.N_device, m.N_des
        pl = s[:, list(m.load_idx)] / self.baseMVA
        pg = s[:, 2 * D + nd : 2 * D + nd + m.N_non_slack_gen] / self.baseMVA
        N = self.planning_steps
        return pl.unsqueeze(2).expand(-1, -1, N), pg.unsqueeze(2).expand(-1, -1, N)


class MPCAgentPerfect(MPCAgent):
    """Perfect forecasts for series-mode tasks (ANM6Easy): the next N columns of the task's periodic tables
    (``mpc_perfect.py:24-40``)."""

    def forecast(self, env):
        tab = torch.as_tensor(env._series, dtype=torch.float64, device=self.device)  # [n_load + n_gen, period]
        period = tab.shape[1]
        t0 = env.state[:, -1].long() + 1
        idx = (t0.unsqueeze(1) + torch.arange(self.planning_steps, device=self.device).unsqueeze(0)) % period  # [E, N]
        cols = tab[:, idx]  # [n, E, N]
        nl = env.simulator.N_load
        return cols[:nl].permute(1, 0, 2) / self.baseMVA, cols[nl:].permute(1, 0, 2) / self.baseMVA
