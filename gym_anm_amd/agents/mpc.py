"""Batched MPC DC-OPF policy: the reference's ``gym_anm/agents/mpc.py`` for ``num_envs`` environments at once.

The reference builds, per environment and per call of ``act()``, an N-stage DC optimal power flow with
cvxpy (``mpc.py:163-319``) and hands it to a generic convex solver.  The problem is a linear program whose
constraint MATRIX depends only on the network and the horizon; what changes from call to call and from
environment to environment are right-hand sides: the load / generation forecasts and the storage state
of charge (``mpc.py:390-417``).  So here

* :class:`BatchedDCOPF` solves all environments in ONE launch of the hand-written interior-point kernel
  (``csrc/anm_mpc.hpp``; the reduced, stage-structured form of the program is documented in ``dcopf.py``);
* :class:`MPCAgent` / :class:`MPCAgentConstant` / :class:`MPCAgentPerfect` keep the reference's names,
  constructor arguments and ``act(env)`` contract (``mpc.py:321-346``, ``mpc_constant.py``, ``mpc_perfect.py``)
  on a :class:`~gym_anm_amd.envs.anm_env.BatchedANMEnv`;
* :class:`DCOPFProgram` spells the reference's program out row by row (``min q.x, l <= A x <= u``); nothing in
  the solve uses it -- it is what the tests check a solution's feasibility against.

All quantities are per-unit, as in the reference.  An LP optimum is not unique in general (curtailment vs
storage, charge vs discharge splits), so two exact solvers may return different optimal actions; what is
comparable is the optimal objective, the feasibility of the action, and the action itself where the minimiser
is unique (tests/test_mpc.py).
"""

from __future__ import annotations

import ctypes as C

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


class BatchedDCOPF:
    """The N-stage DC-OPF of every environment of a batch, solved on the device by ONE launch of the
    interior-point kernel (``anm_mpc_solve_f64``, ``csrc/anm_mpc.hpp``): one lane per stage, ``64 / 2^ceil(log2 N)``
    environments per wavefront, the network tables of the reduced program (``dcopf.py``) as scalar loads.

    ``solve(P_load_forecast [E, n_load, N], P_gen_forecast [E, n_gen, N], soc [E, n_des])`` -> the first-stage
    ``[P_gen.., P_des..]`` (p.u.); ``objective``, ``iters``, ``info`` and, on request, the whole primal solution stay
    available as tensors.  ``info[:, 0]`` is the final complementarity ``mu``, ``info[:, 1]`` a STATUS (0 ok, 1 a stage has
    no interior start, 2 an angle row left out of the solve is violated -- ``include/anm_mi355x.h``), ``info[:, 2]`` the
    largest dual residual.  ``converged`` folds them into the verdict every caller should apply."""

    TOL_DEFAULT, MAX_ITER_DEFAULT = 1e-11, 40   # the library's defaults (anm_mpc_capi.inc), always passed explicitly

    def __init__(self, simulator, gamma, safety_margin, planning_steps, tol=None, max_iter=None, keep_solution=False,
                 keep_trace=False, angle_rows=None, size_class=None):
        from .. import _lib

        self.backend, self.device = simulator.backend, simulator.device
        if getattr(self.backend, "generic", False) or size_class:
            # the simulator steps on the table-driven lane-group kernels of a library compiled for another topology; the
            # MPC kernel depends on the network's SIZES: a precompiled size class the network is padded into (no compile),
            # or a small library of its own (size_class=True forces a class: tests)
            self.backend = _lib.load_mpc_for_topology(simulator.model.topology(), size_class=size_class)
        self._device_ctx = simulator._device_ctx
        self._stream_ptr = lambda: _lib_stream(self.device)
        self.N = int(planning_steps)
        desc, self._keep = _lib.network_desc(simulator.model)
        self._handle = C.c_void_p()
        lib = self.backend.lib
        with self._device_ctx():
            self.backend.check(lib.anm_mpc_create(C.byref(desc), float(gamma), float(safety_margin), self.N,
                                                  C.byref(self._handle)), "anm_mpc_create")
        d = _lib.MpcDims()
        self.backend.check(lib.anm_mpc_dims_of(self._handle, C.byref(d)), "anm_mpc_dims_of")
        self.dims = d
        # angle_rows: None = the library's choice (dims.angle_rows: the rows |theta| <= pi ride through the solve only
        # where an angle can come near pi within the device limits); True / False force them in / out (left out, the
        # solution's angles are checked: info[:, 1] == 2 where one exceeds pi)
        # tolerance and iteration cap are always sent explicitly (0 would mean "the library's default": the trace buffer
        # below is sized by max_iter, and a default that changed under it would be an out-of-bounds write)
        self.tol = self.TOL_DEFAULT if tol is None else float(tol)
        self.max_iter = self.MAX_ITER_DEFAULT if max_iter is None else int(max_iter)
        if not (self.tol > 0.0 and self.max_iter > 0):
            raise ValueError("tol and max_iter must be positive")
        self.opts = _lib.MpcOpts(tol=self.tol, max_iter=self.max_iter, angle_rows=0 if angle_rows is None else (1 if angle_rows else 2))
        self.dual_tol = 1e-6 * (1.0 + float(simulator.lamb))   # against costs of 1 ... lamb per p.u.
        self.keep_solution, self.keep_trace = keep_solution, keep_trace
        self._E = None

    def tables(self):
        """host copy of the constant table the kernel reads (layout: ``mpc::Sz`` in csrc/anm_mpc.hpp)"""
        out = np.zeros(self.dims.table_doubles)
        self.backend.check(self.backend.lib.anm_mpc_get_tables(self._handle, out.ctypes.data_as(C.POINTER(C.c_double))),
                           "anm_mpc_get_tables")
        return out

    def _buffers(self, E):
        if self._E != E:
            d, f = self.dims, dict(dtype=torch.float64, device=self.device)
            self.u0 = torch.zeros((E, d.n_ctrl), **f)
            self.objective = torch.zeros(E, **f)
            self.iters = torch.zeros(E, dtype=torch.int32, device=self.device)
            self.info = torch.zeros((E, 3), **f)
            self.solution = torch.zeros((E, self.N, d.n_stage_vars), **f) if self.keep_solution else None
            self.trace = torch.full((E, self.max_iter + 1, 12), float("nan"), **f) if self.keep_trace else None
            self.opts.trace = None if self.trace is None else self.trace.data_ptr()
            self._E = E

    def solve(self, P_load_forecast, P_gen_forecast, soc):
        d = self.dims
        E = int(soc.shape[0])
        f = dict(dtype=torch.float64, device=self.device)
        # [E, n, N] (the reference's forecast layout, mpc.py:348-372) -> stage-major [E, N, n]
        pl = torch.as_tensor(P_load_forecast, **f).reshape(E, d.n_load, self.N).permute(0, 2, 1).contiguous()
        pg = torch.as_tensor(P_gen_forecast, **f).reshape(E, d.n_gen, self.N).permute(0, 2, 1).contiguous()
        sc = torch.as_tensor(soc, **f).reshape(E, d.n_des).contiguous()
        self._buffers(E)
        with self._device_ctx():
            rc = self.backend.lib.anm_mpc_solve_f64(
                self._handle, E, pl.data_ptr(), pg.data_ptr(), sc.data_ptr(), self.u0.data_ptr(), self.objective.data_ptr(),
                self.iters.data_ptr(), self.info.data_ptr(), 0 if self.solution is None else self.solution.data_ptr(),
                C.byref(self.opts), self._stream_ptr())
        self.backend.check(rc, "anm_mpc_solve_f64")
        return self.u0

    def act(self, forecast, env, act_low, act_high):
        """``MPCAgent.act`` as ONE launch (``anm_mpc_act_f64``): the forecasts are gathered inside the kernel -- ``forecast``
        1: constant, from the state rows of ``env``; 2: perfect, from its periodic tables -- and the first stage's
        set-points come back as the clipped MW action rows ``[E, 2 n_gen + 2 n_des]`` (a persistent buffer)."""
        d = self.dims
        E = int(env.num_envs)
        self._buffers(E)
        if getattr(self, "_action", None) is None or self._action.shape[0] != E:
            self._action = torch.zeros((E, 2 * d.n_gen + 2 * d.n_des), dtype=torch.float64, device=self.device)
        series_ptr, period = None, 0
        if forecast == 2:
            if getattr(self, "_series_src", None) is not env._series:
                self._series_src = env._series
                self._series_dev = torch.as_tensor(env._series, dtype=torch.float64, device=self.device).contiguous()
            series_ptr, period = self._series_dev.data_ptr(), int(self._series_dev.shape[1])
        same = env._state_same
        aux = None   # the time index is read from the last column of the state row (what `forecast()` does)
        soc = env.simulator.soc
        with self._device_ctx():
            rc = self.backend.lib.anm_mpc_act_f64(
                self._handle, E, int(forecast), env._state_buf.data_ptr(), None if same is None else env._state_obs.data_ptr(),
                None if same is None else same.data_ptr(), int(env._state_buf.shape[1]), None if aux is None else aux.data_ptr(),
                series_ptr, period, soc.data_ptr(), act_low.data_ptr(), act_high.data_ptr(), self._action.data_ptr(),
                self.u0.data_ptr(), self.objective.data_ptr(), self.iters.data_ptr(), self.info.data_ptr(), C.byref(self.opts),
                self._stream_ptr())
        self.backend.check(rc, "anm_mpc_act_f64")
        return self._action

    @property
    def converged(self):
        """``[E]`` bool: the last solve reached the complementarity tolerance ``mu <= tol (1 + |objective|)`` (the kernel's
        own stop test, read from ``info[:, 0]``: a solve that meets it in its last allowed iteration counts), status 0,
        and a dual residual that is small against the costs (the kernel's stop test does not look at it)."""
        mu_ok = self.info[:, 0] <= self.tol * (1.0 + self.objective.abs())
        return mu_ok & (self.info[:, 1] == 0) & (self.info[:, 2] <= self.dual_tol)

    def __del__(self):
        try:
            if getattr(self, "_handle", None) and self._handle.value:
                self.backend.lib.anm_mpc_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass


def _lib_stream(device):
    if device.type != "cuda":
        return None
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class MPCAgent:
    """``MPCAgent(simulator, action_space, gamma, safety_margin=0.9, planning_steps=1)`` (``mpc.py:33-122``):
    ``act(env)`` returns the ``[num_envs, action_dim]`` tensor of the first-stage set-points (MW; reactive
    set-points 0, ``mpc.py:383-388``), clipped to the action space.

    ``last_converged`` is the per-environment mask of solves that reached the tolerance; like the reference, which
    prints ``OPF problem is <status>`` and goes on (``mpc.py:377-379``), ``act`` warns when it is not all true."""

    def __init__(self, simulator, action_space, gamma, safety_margin=0.9, planning_steps=1, tol=None, max_iter=None,
                 size_class=None):
        self.simulator = simulator
        self.action_space = action_space
        self.gamma, self.safety_margin, self.planning_steps = gamma, safety_margin, int(planning_steps)
        self.baseMVA, self.lamb, self.delta_t = simulator.baseMVA, simulator.lamb, simulator.delta_t
        m = simulator.model
        self.load_ids = [m.dev_ids[k] for k in m.load_idx]
        self.non_slack_gen_ids = [m.dev_ids[k] for k in m.gen_idx]
        self.des_ids = [m.dev_ids[k] for k in m.des_idx]
        self.device = simulator.device
        self.solver = BatchedDCOPF(simulator, gamma, safety_margin, planning_steps, tol=tol, max_iter=max_iter, size_class=size_class)
        self._acted, self._conv = False, None
        # `tensor / python scalar` is a multiplication by the reciprocal on the GPU; the reference divides (`/ self.baseMVA`):
        # the forecasts are divided by this 0-dim tensor (a true division), like the fused kernel does
        self._base_t = torch.tensor(float(self.baseMVA), dtype=torch.float64, device=self.device)
        self._lo = torch.as_tensor(np.asarray(action_space.low, float), device=self.device)
        self._hi = torch.as_tensor(np.asarray(action_space.high, float), device=self.device)

    def forecast(self, env):
        """-> (P_load_forecast [E, n_load, N], P_gen_forecast [E, n_gen, N]) in p.u.  (``mpc.py:348-372``)."""
        raise NotImplementedError()

    def _soc(self, env):
        return env.simulator.soc  # p.u., [E, n_des]  (mpc.py:417 reads des_soc in pu)

    def solve(self, P_load_forecast, P_gen_forecast, soc):
        """-> first-stage [P_gen.., P_des..] in p.u., [E, n_gen + n_des]"""
        return self.solver.solve(P_load_forecast, P_gen_forecast, soc)

    FUSED_FORECAST = 0   # the stock agents: which forecast anm_mpc_act_f64 gathers inside the kernel (0: none, use forecast())

    def _fused(self, env):
        """The whole of act() can run as one launch: a stock forecast (not overridden by a subclass), the stock solve and
        state of charge, a batched environment of this package that has what the kernel reads."""
        cls = type(self)
        stock = {1: MPCAgentConstant, 2: MPCAgentPerfect}.get(self.FUSED_FORECAST)
        return (stock is not None and cls.forecast is stock.forecast and cls.solve is MPCAgent.solve and cls._soc is MPCAgent._soc
                and hasattr(self.solver.backend.lib, "anm_mpc_act_f64") and hasattr(env, "_state_buf")
                and (self.FUSED_FORECAST == 1 or getattr(env, "_series", None) is not None)
                and env._state_buf.shape[1] >= 2 * env.simulator.model.N_device + env.simulator.model.N_des
                + env.simulator.model.N_non_slack_gen + (1 if self.FUSED_FORECAST == 2 else 0))

    warn_unconverged = True   # False: act() neither synchronises nor launches anything but the solve (last_converged stays lazy)
    # False (default): act() returns a tensor of its own, like the reference's fresh array and like the unfused path --
    # actions kept across steps (rollout lists, replay buffers) stay what they were.  True: the fused path returns the
    # solver's persistent buffer, which the next act() overwrites (no allocation, no copy kernel: what bench.py times)
    reuse_action_buffer = False

    @property
    def last_converged(self):
        """per-environment mask of the solves of the last ``act`` that reached the tolerance (``BatchedDCOPF.converged``);
        None before the first ``act``"""
        if self._conv is None and self._acted:
            self._conv = self.solver.converged
        return self._conv

    def _check_converged(self, n):
        self._acted, self._conv = True, None
        if self.warn_unconverged and not bool(self.last_converged.all()):
            import warnings

            warnings.warn("OPF problem did not reach the tolerance in %d of %d environments (see last_converged)"
                          % (int((~self.last_converged).sum()), n))

    def act(self, env):
        """``env``: a batched environment -> ``[num_envs, action_dim]`` tensor; or one of the NumPy-facing
        single-environment classes (``ANMEnv``, ``ANM6``, ``ANM6Easy``) -> 1-D NumPy action, as in the reference's
        ``examples/mpc_*.py``.  With a stock forecast the whole call is ONE kernel launch (forecast gather, solve, scaling
        to MW, clipping: ``anm_mpc_act_f64``); the tensor returned is the caller's own unless ``reuse_action_buffer`` is set
        (then it is the solver's buffer, overwritten by the next call)."""
        if hasattr(env, "vec"):
            return self.act(env.vec)[0].cpu().numpy()
        if self._fused(env):
            a = self.solver.act(self.FUSED_FORECAST, env, self._lo, self._hi)
            self._check_converged(a.shape[0])
            return a if self.reuse_action_buffer else a.clone()
        pl, pg = self.forecast(env)
        u0 = self.solve(pl, pg, self._soc(env)) * self.baseMVA
        ng = self.solver.dims.n_gen
        P_gen, P_des = u0[:, :ng], u0[:, ng:]
        a = torch.cat((P_gen, torch.zeros_like(P_gen), P_des, torch.zeros_like(P_des)), dim=1)
        # reached the tolerance with status 0 and a dual residual that is small against the costs (BatchedDCOPF.converged)
        self._check_converged(a.shape[0])
        return torch.minimum(torch.maximum(a, self._lo), self._hi)  # mpc.py:341-344


class MPCAgentConstant(MPCAgent):
    """Constant forecasts: the current loads and generation potentials persist over the horizon
    (``mpc_constant.py:24-35``).  They are read from the state vector: dev_p of the loads and gen_p_max, MW."""

    FUSED_FORECAST = 1

    def forecast(self, env):
        m = env.simulator.model
        s = env.state
        D, nd = m.N_device, m.N_des
        pl = s[:, list(m.load_idx)] / self._base_t
        pg = s[:, 2 * D + nd : 2 * D + nd + m.N_non_slack_gen] / self._base_t
        N = self.planning_steps
        return pl.unsqueeze(2).expand(-1, -1, N), pg.unsqueeze(2).expand(-1, -1, N)


class MPCAgentPerfect(MPCAgent):
    """Perfect forecasts for series-mode tasks (ANM6Easy): the next N columns of the task's periodic tables
    (``mpc_perfect.py:24-40``)."""

    FUSED_FORECAST = 2

    def forecast(self, env):
        tab = torch.as_tensor(env._series, dtype=torch.float64, device=self.device)  # [n_load + n_gen, period]
        period = tab.shape[1]
        t0 = env.state[:, -1].long() + 1
        idx = (t0.unsqueeze(1) + torch.arange(self.planning_steps, device=self.device).unsqueeze(0)) % period  # [E, N]
        cols = tab[:, idx]  # [n, E, N]
        nl = env.simulator.N_load
        return cols[:nl].permute(1, 0, 2) / self._base_t, cols[nl:].permute(1, 0, 2) / self._base_t
