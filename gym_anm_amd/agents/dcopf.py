"""The N-stage DC optimal power flow of ``gym_anm/agents/mpc.py:163-319`` in the reduced, stage-structured form the
HIP interior-point kernel (``csrc/anm_mpc.hpp``) works on.

The reference states the program with one angle per bus, one injection per device, a charge / discharge pair per
storage unit and (through cvxpy's ``maximum(0, abs(.))``) one epigraph variable per branch, per stage; most of
that is pinned by equalities:

* the DC balance (``mpc.py:232-245``) with the pinned angle (``:302``) determines every angle and the slack
  device's injection as LINEAR functions of the other injections:  ``theta = Th_l p_load + Th_c u``,
  ``flow = Ph_l p_load + Ph_c u``, ``P_slack = sg_l . p_load + sg_c . u`` with ``u = [P_gen.., P_des..]`` (one small
  inverse on the host, once per network -- also for meshed networks and phase shifters, where Im(Y) is not symmetric);
* loads equal their forecast (``:247-251``).

What is left per stage i are ``n_gen + 2 n_des + n_branch`` variables

    xi_g in [0, 1]      P_g = P_min + xi_g * max(0, min(P_max, forecast_g,i) - P_min)       (``:253-258, 267-271``)
    soc_i, d_i          state of charge AFTER the stage and discharge power: the charge power is
                        p_c = (soc_i - soc_{i-1}) / (dt eff) + d_i / eff^2,  P_des = d_i - p_c  (``:273-291``)
    t_e >= 0            epigraph of max(0, |flow_e| - beta rate_e)                           (``:308-311``)

and only inequality rows ``g.v <= h`` (boxes, p_c >= 0, d >= 0, SoC window, |theta| <= pi, the three epigraph rows
per branch).  A stage's rows touch its own variables and ``soc_{i-1}`` only, so the normal matrix of an
interior-point step is block tridiagonal with the storage units as the only coupling -- the kernel maps one stage
to one lane.  Scaling a generator by its own interval (``xi``) keeps the program well posed when a forecast closes
the interval (0 <= P <= 0 at night): ``xi`` then simply does not matter.

Everything here is one-time host work (NumPy).  ``dense_program`` spells the reduced program out as dense arrays
for tests; the kernel receives ``tables()``.
"""

from __future__ import annotations

import numpy as np

from .. import errors as E
from ..model import CLASSICAL, SLACK


class ReducedDCOPF:
    def __init__(self, model, gamma, safety_margin, planning_steps):
        m, N = model, int(planning_steps)
        if N < 1:
            raise ValueError("planning_steps must be at least 1")
        self.N, self.gamma, self.margin = N, float(gamma), float(safety_margin)
        nb, nd, nbr = m.N_bus, m.N_device, m.N_branch
        self.loads, self.gens, self.des = list(m.load_idx), list(m.gen_idx), list(m.des_idx)
        nl, ng, ns = len(self.loads), len(self.gens), len(self.des)
        self.nb, self.nbr, self.nl, self.ng, self.ns, self.nc = nb, nbr, nl, ng, ns, ng + ns
        self.dt, self.lamb = float(m.delta_t), float(m.lamb)
        B = np.asarray(m.Y_bus.todense() if hasattr(m.Y_bus, "todense") else m.Y_bus).imag  # mpc.py:113
        pin = int(m.slack_dev)  # mpc.py:302 pins the angle at the slack DEVICE's position (sic)
        if not 0 <= pin < nb:
            raise E.UnsupportedNetworkError("the reference's DC-OPF pins the angle with index %d (the slack device's "
                                            "position), which is not a bus of this %d-bus network" % (pin, nb))
        self.pin = pin
        Lap = np.zeros((nb, nb))
        for f, t in zip(m.br_f, m.br_t):  # mpc.py:232-245
            Lap[f, f] += B[f, t]
            Lap[f, t] -= B[f, t]
            Lap[t, t] += B[t, f]
            Lap[t, f] -= B[t, f]
        slack_bus = int(m.dev_bus[m.slack_dev])
        keep = [b for b in range(nb) if b != pin]
        M = np.zeros((nb, nb))
        M[:, : nb - 1] = Lap[:, keep]
        M[slack_bus, nb - 1] = -1.0  # ... - P_slack at the slack device's bus
        try:
            K = np.linalg.inv(M)
        except np.linalg.LinAlgError as ex:
            raise E.UnsupportedNetworkError("the DC balance of this network is singular (disconnected network?)") from ex
        D = np.zeros((nb, nd))  # injections of the devices into the buses
        for k in range(nd):
            D[m.dev_bus[k], k] = 1.0
        Th = np.zeros((nb, nd))
        Th[keep, :] = K[: nb - 1] @ D
        sg = K[nb - 1] @ D
        Ph = np.array([B[f, t] * (Th[f] - Th[t]) for f, t in zip(m.br_f, m.br_t)]).reshape(nbr, nd)
        ctrl = self.gens + self.des
        self.Th_l, self.Th_c = Th[:, self.loads], Th[:, ctrl]
        self.Ph_l, self.Ph_c = Ph[:, self.loads], Ph[:, ctrl]
        self.sg_l, self.sg_c = sg[self.loads], sg[ctrl]
        # stage cost of u (before the discount): the slack's injection + the classical generators (mpc.py:304-306)
        self.cost_c = self.sg_c.copy()
        for j, k in enumerate(self.gens):
            if m.dev_type[k] in (SLACK, CLASSICAL):
                self.cost_c[j] += 1.0
        self.lim = self.margin * np.asarray(m.br_rate, float)
        self.g_pmin, self.g_pmax = np.asarray(m.dev_p_min, float)[self.gens], np.asarray(m.dev_p_max, float)[self.gens]
        self.s_pmin, self.s_pmax = np.asarray(m.dev_p_min, float)[self.des], np.asarray(m.dev_p_max, float)[self.des]
        self.soc_min, self.soc_max = np.asarray(m.dev_soc_min, float)[self.des], np.asarray(m.dev_soc_max, float)[self.des]
        self.eff = np.asarray(m.dev_eff, float)[self.des]
        self.theta_rows = keep
        self.nv = ng + 2 * ns + nbr                      # variables per stage
        self.nr = 2 * ng + 6 * ns + 2 * (nb - 1) + 3 * nbr  # rows per stage
        self.base = float(m.baseMVA)
        self.slack_dev, self.n_dev = int(m.slack_dev), nd

    # ------------------------------------------------------------------------------------------------
    def widths(self, pg):
        """[ng, N] forecasts -> generator interval widths max(0, min(P_max, forecast) - P_min), [ng, N]."""
        return np.maximum(np.minimum(self.g_pmax[:, None], pg) - self.g_pmin[:, None], 0.0)

    def dense_program(self, pl, pg, soc0):
        """One environment: (G [N nr, N nv], h, c, c0) of  min c.v  s.t.  G v <= h;  the objective of the reference's
        program is ``c.v + c0``.  Variable order per stage: xi (ng), soc (ns), d (ns), t (nbr)."""
        N, ng, ns, nbr, nv = self.N, self.ng, self.ns, self.nbr, self.nv
        pl, pg, soc0 = np.asarray(pl, float).reshape(self.nl, N), np.asarray(pg, float).reshape(ng, N), np.asarray(soc0, float)
        wd = self.widths(pg)
        XI, SOC, DD, TT = 0, ng, ng + ns, ng + 2 * ns
        rows, hs = [], []
        c, const = np.zeros(N * nv), 0.0
        a1, a2 = 1.0 / (self.dt * self.eff), 1.0 / self.eff**2  # p_c = a1 (soc_i - soc_prev) + a2 d
        for i in range(N):
            o, op = i * nv, (i - 1) * nv

            def lin_u(coef):
                """row over the stage variables (+ soc_{i-1}) and constant of  coef . u  with u = [P_g, P_des]"""
                r = np.zeros(N * nv)
                k0 = 0.0
                for j in range(ng):
                    r[o + XI + j] += coef[j] * wd[j, i]
                    k0 += coef[j] * self.g_pmin[j]
                for j in range(ns):  # P_des = d - p_c = (1 - a2) d - a1 soc_i + a1 soc_prev
                    cj = coef[ng + j]
                    r[o + DD + j] += cj * (1.0 - a2[j])
                    r[o + SOC + j] -= cj * a1[j]
                    if i > 0:
                        r[op + SOC + j] += cj * a1[j]
                    else:
                        k0 += cj * a1[j] * soc0[j]
                return r, k0

            def add(r, k0, hi):  # r.v + k0 <= hi
                rows.append(r)
                hs.append(hi - k0)

            for j in range(ng):
                r = np.zeros(N * nv)
                r[o + XI + j] = 1.0
                add(r, 0.0, 1.0)
                add(-r, 0.0, 0.0)
            for j in range(ns):
                e = np.zeros(self.nc)
                e[ng + j] = 1.0
                r, k0 = lin_u(e)
                add(r, k0, self.s_pmax[j])
                add(-r, -k0, -self.s_pmin[j])
                rc = np.zeros(N * nv)  # p_c >= 0
                rc[o + SOC + j] = a1[j]
                rc[o + DD + j] = a2[j]
                kc = 0.0
                if i > 0:
                    rc[op + SOC + j] = -a1[j]
                else:
                    kc = -a1[j] * soc0[j]
                add(-rc, -kc, 0.0)
                rd = np.zeros(N * nv)
                rd[o + DD + j] = 1.0
                add(-rd, 0.0, 0.0)
                rs = np.zeros(N * nv)
                rs[o + SOC + j] = 1.0
                add(rs, 0.0, self.soc_max[j])
                add(-rs, 0.0, -self.soc_min[j])
            for b in self.theta_rows:
                r, k0 = lin_u(self.Th_c[b])
                k0 += self.Th_l[b] @ pl[:, i]
                add(r, k0, np.pi)
                add(-r, -k0, np.pi)
            for e_ in range(nbr):
                r, k0 = lin_u(self.Ph_c[e_])
                k0 += self.Ph_l[e_] @ pl[:, i]
                rt = np.zeros(N * nv)
                rt[o + TT + e_] = 1.0
                add(r - rt, k0, self.lim[e_])
                add(-r - rt, -k0, self.lim[e_])
                add(-rt, 0.0, 0.0)
            r, k0 = lin_u(self.cost_c)
            w = self.gamma**i
            c += w * r
            c[o + TT : o + TT + nbr] += w * self.lamb
            const += w * (k0 + self.sg_l @ pl[:, i])
        return np.array(rows), np.array(hs), c, const

    def start_point(self, pl, pg, soc0):
        N, ng, ns, nbr, nv = self.N, self.ng, self.ns, self.nbr, self.nv
        v = np.zeros(N * nv)
        soc0 = np.asarray(soc0, float)
        mid = np.clip(soc0, self.soc_min + 0.05 * (self.soc_max - self.soc_min), self.soc_max - 0.05 * (self.soc_max - self.soc_min))
        for i in range(N):
            o = i * nv
            v[o : o + ng] = 0.5
            v[o + ng : o + ng + ns] = mid
            v[o + ng + ns : o + ng + 2 * ns] = 0.1 * self.s_pmax
            v[o + ng + 2 * ns : o + nv] = 0.1
        return v

    def start_duals(self):
        """multipliers of the rows at the start: dual feasible for the epigraph variables (their three rows share
        the stage's weight of the overload term), 1 elsewhere"""
        z = np.ones(self.N * self.nr)
        o3 = self.nr - 3 * self.nbr
        for i in range(self.N):
            z[i * self.nr + o3 : (i + 1) * self.nr] = self.gamma**i * self.lamb / 3.0
        return z

    def first_stage_p_dev(self, v, pl, pg, soc0):
        """first-stage injections of every device, [n_dev] p.u., from a solution vector"""
        N, ng, ns = self.N, self.ng, self.ns
        pl, pg = np.asarray(pl, float).reshape(self.nl, N), np.asarray(pg, float).reshape(ng, N)
        wd = self.widths(pg)[:, 0]
        a1, a2 = 1.0 / (self.dt * self.eff), 1.0 / self.eff**2
        p = np.zeros(self.n_dev)
        p[self.loads] = pl[:, 0]
        pgv = self.g_pmin + wd * v[:ng]
        d = v[ng + ns : ng + 2 * ns]
        pc = a1 * (v[ng : ng + ns] - np.asarray(soc0, float)) + a2 * d
        p[self.gens] = pgv
        p[self.des] = d - pc
        u = np.concatenate((pgv, d - pc))
        p[self.slack_dev] = self.sg_l @ pl[:, 0] + self.sg_c @ u
        return p
