"""NumPy prototype of the STRUCTURED interior-point iteration of csrc/anm_mpc.hpp (development aid): one array
slot per stage where the kernel has one lane per stage.

Per stage the INPUTS are a = (xi [ng], p_c [ns], d [ns]) and the STATE is the state of charge after the stage,
sig_i = sig_{i-1} + dt eff p_c - dt/eff d.  Every row touches inputs only (boxes, theta, flows with their epigraph
variables) or the state only (SoC window), so a Newton step is a linear-quadratic control problem: solved by a
Riccati sweep in the cancellation-free form  P' = P (I + M P)^-1,  M = B R^-1 B'.

    python scripts/ipm_struct_proto.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from gym_anm_amd import networks  # noqa: E402
from gym_anm_amd.agents.dcopf import ReducedDCOPF  # noqa: E402
from gym_anm_amd.model import NetworkModel  # noqa: E402


def ipm_struct(red, pl, pg, soc0, tol=1e-9, max_iter=40, verbose=False, n_ref=0):
    N, ng, ns, nbr = red.N, red.ng, red.ns, red.nbr
    nc, na = ng + ns, ng + 2 * ns
    pl, pg, soc0 = np.asarray(pl, float).reshape(red.nl, N), np.asarray(pg, float).reshape(ng, N), np.asarray(soc0, float)
    wd = red.widths(pg).T                       # [N, ng]
    bc, bd = red.dt * red.eff, red.dt / red.eff  # sig_i = sig_{i-1} + bc p_c - bd d
    Thc, Phc = red.Th_c[red.theta_rows], red.Ph_c  # [nb1, nc], [nbr, nc]
    th0 = (red.Th_l[red.theta_rows] @ pl).T      # [N, nb1]
    f0 = (red.Ph_l @ pl).T                       # [N, nbr]
    wgt = red.gamma ** np.arange(N)
    cu = wgt[:, None] * red.cost_c[None, :]      # [N, nc]
    ct = wgt * red.lamb                          # [N]
    lim = red.lim
    Bm = np.zeros((ns, na))                      # state <- inputs
    Jm = np.zeros((N, nc, na))                   # u <- inputs
    for j in range(ng):
        Jm[:, j, j] = wd[:, j]
    for j in range(ns):
        Bm[j, ng + j], Bm[j, ng + ns + j] = bc[j], -bd[j]
        Jm[:, ng + j, ng + j], Jm[:, ng + j, ng + ns + j] = -1.0, 1.0
    # start
    xi = np.full((N, ng), 0.5)
    pc = np.tile(0.05 * (red.s_pmax - red.s_pmin), (N, 1))
    d = pc * (bc / bd)[None, :]                  # keeps the state of charge where it is
    t = np.full((N, nbr), 0.1)

    def state(pc, d):
        return soc0[None, :] + np.cumsum(bc * pc - bd * d, axis=0)

    def row_values(xi, pc, d, t):
        """g.v - h of every row class (<= 0 when feasible)"""
        sig = state(pc, d)
        u = np.hstack((red.g_pmin + wd * xi, d - pc))
        th = th0 + u @ Thc.T
        fl = f0 + u @ Phc.T
        pdes = d - pc
        return dict(xi=(xi - 1.0, -xi), pdes=(pdes - red.s_pmax, red.s_pmin - pdes), pcd=(-pc, -d),
                    soc=(sig - red.soc_max, red.soc_min - sig), th=(th - np.pi, -th - np.pi),
                    fl=(fl - t - lim, -fl - t - lim, -t)), u

    rv, u = row_values(xi, pc, d, t)
    S = {k: [np.maximum(-x, 1e-2) for x in v] for k, v in rv.items()}
    Z = {k: [np.ones_like(x) for x in v] for k, v in rv.items()}
    Z["fl"] = [np.tile((ct / 3.0)[:, None], (1, nbr)) for _ in range(3)]
    m_rows = N * red.nr
    for it in range(max_iter + 1):
        rv, u = row_values(xi, pc, d, t)
        RP = {k: [x + s for x, s in zip(rv[k], S[k])] for k in rv}
        mu = sum((s * z).sum() for k in S for s, z in zip(S[k], Z[k])) / m_rows
        W = {k: [z / s for s, z in zip(S[k], Z[k])] for k in S}
        w1, w2, w3 = W["fl"]
        htt = w1 + w2 + w3
        hat = (4 * w1 * w2 + w3 * (w1 + w2)) / htt   # weight on (a.du)^2 once the epigraph variable is eliminated
        hut = (w2 - w1) / htt

        def grad(Zh, eliminate_t):
            """gradient of the Lagrangian for multipliers Zh: (inputs [N, na], state [N, ns], epigraph [N, nbr])"""
            gt = ct[:, None] - Zh["fl"][0] - Zh["fl"][1] - Zh["fl"][2]
            ga = Zh["fl"][0] - Zh["fl"][1]
            if eliminate_t:
                ga = ga - hut * gt
            gu = cu + (Zh["th"][0] - Zh["th"][1]) @ Thc + ga @ Phc
            gu[:, ng:] += Zh["pdes"][0] - Zh["pdes"][1]
            g_in = np.einsum("nia,ni->na", Jm, gu)
            g_in[:, :ng] += Zh["xi"][0] - Zh["xi"][1]
            g_in[:, ng:ng + ns] -= Zh["pcd"][0]
            g_in[:, ng + ns:] -= Zh["pcd"][1]
            g_st = Zh["soc"][0] - Zh["soc"][1]
            return g_in, g_st, gt

        g_in, g_st, rd_t = grad(Z, False)
        lam = np.cumsum(g_st[::-1], axis=0)[::-1]            # d/d a_i of the state rows = B' sum_{k >= i} g_st_k
        rd_in = g_in + lam @ Bm
        rpmax = max(np.abs(x).max() for k in RP for x in RP[k])
        rdmax = max(np.abs(rd_in).max(), np.abs(rd_t).max())
        obj = float((cu * u).sum() + (ct[:, None] * t).sum() + (wgt * (red.sg_l @ pl)).sum())
        if verbose:
            print(it, "mu %.2e rp %.2e rd %.2e obj %.10f" % (mu, rpmax, rdmax, obj))
        final = False
        if mu < tol * (1 + abs(obj)) and rpmax < 1e-9 and rdmax < float(os.environ.get('RDTOL', '1e30')) * (1 + ct.max()):
            break
        # ---- factorisation (once per iteration) ----
        Hu = np.einsum("nb,bi,bj->nij", W["th"][0] + W["th"][1], Thc, Thc) + np.einsum("ne,ei,ej->nij", hat, Phc, Phc)
        for j in range(ns):
            Hu[:, ng + j, ng + j] += W["pdes"][0][:, j] + W["pdes"][1][:, j]
        R = np.einsum("nia,nij,njb->nab", Jm, Hu, Jm)
        for j in range(ng):
            R[:, j, j] += W["xi"][0][:, j] + W["xi"][1][:, j]
        for j in range(ns):
            R[:, ng + j, ng + j] += W["pcd"][0][:, j]
            R[:, ng + ns + j, ng + ns + j] += W["pcd"][1][:, j]
        RHO_REL = float(os.environ.get("RHO", "0"))
        if RHO_REL > 0:
            if os.environ.get("LOCALRHO"):
                dmax = np.max(np.diagonal(R, axis1=1, axis2=2), axis=1)
                R = R + (RHO_REL * dmax + 1e-14)[:, None, None] * np.eye(na)[None]
            else:
                wm = max(x.max() for k in W for x in W[k])
                R = R + (RHO_REL * wm + 1e-12) * np.eye(na)[None]
        Lc = np.linalg.cholesky(R)
        Rinv = lambda X: np.linalg.solve(Lc.transpose(0, 2, 1), np.linalg.solve(Lc, X))  # noqa: E731
        RiBt = Rinv(np.tile(Bm.T[None], (N, 1, 1)))          # [N, na, ns]
        M = np.einsum("sa,nat->nst", Bm, RiBt)               # B R^-1 B'
        Q = W["soc"][0] + W["soc"][1]                        # [N, ns] (diagonal)
        P = np.zeros((N, ns, ns))                            # value function of the state AFTER stage i
        Pn = np.zeros((ns, ns))
        IMP = np.zeros((N, ns, ns))
        for i in range(N - 1, -1, -1):
            P[i] = np.diag(Q[i]) + Pn
            IMP[i] = np.eye(ns) + M[i] @ P[i]
            Pn = P[i] @ np.linalg.inv(IMP[i])                # P (I + M P)^-1, symmetric

        if verbose:
            print("   Q", Q.ravel()[:4], "M", M.ravel()[:4], "wmax by class", {k: "%.1e" % max(x.max() for x in W[k]) for k in W})

        def backsolve(g_in, g_st, gt):
            """Newton step for the gradient pieces (epigraph variables eliminated) -> steps and g.dv per row"""
            rt = -gt
            mvec = np.einsum("sa,na->ns", Bm, Rinv(g_in[:, :, None])[:, :, 0])     # B R^-1 r
            p = np.zeros((N, ns))
            pn = np.zeros(ns)
            for i in range(N - 1, -1, -1):
                p[i] = g_st[i] + pn
                pn = np.linalg.solve(IMP[i].T, p[i] - P[i] @ mvec[i])             # (I + P M)^-1 (p - P B R^-1 r)
            x = np.zeros((N, ns))
            xp = np.zeros(ns)
            for i in range(N):
                x[i] = np.linalg.solve(IMP[i], xp - mvec[i] - M[i] @ p[i])
                xp = x[i]
            co = np.einsum("nst,nt->ns", P, x) + p                                 # costate
            da = -Rinv((g_in + co @ Bm)[:, :, None])[:, :, 0]
            # the state step from the inputs themselves (what the rows will see)
            dsg = x if os.environ.get('USEX') else np.cumsum(da @ Bm.T, axis=0)
            dxi, dpc, dd = da[:, :ng], da[:, ng:ng + ns], da[:, ng + ns:]
            du = np.einsum("nia,na->ni", Jm, da)
            al = du @ Phc.T
            dt = rt / htt - hut * al
            dth = du @ Thc.T
            dpdes = dd - dpc
            # g.dv of the three epigraph rows without the cancellation of  al - dt  when one weight dominates
            e1 = (al * (2 * w2 + w3) - rt) / htt
            e2 = (-al * (2 * w1 + w3) - rt) / htt
            GV = dict(xi=(dxi, -dxi), pdes=(dpdes, -dpdes), pcd=(-dpc, -dd), soc=(dsg, -dsg), th=(dth, -dth),
                      fl=(e1, e2, -dt))
            return (dxi, dpc, dd, dt), GV

        def solve(RC, n_ref=0):
            Zh = {k: [z + rc / s + w * rp for z, rc, s, w, rp in zip(Z[k], RC[k], S[k], W[k], RP[k])] for k in S}
            Qd = {k: [zh - z for zh, z in zip(Zh[k], Z[k])] for k in S}
            (dxi, dpc, dd, dt), GV = backsolve(*grad(Zh, True))
            for _ in range(n_ref):
                Zf = {k: [zh + w * gv for zh, w, gv in zip(Zh[k], W[k], GV[k])] for k in S}
                (exi, epc, ed, et), GV2 = backsolve(*grad(Zf, True))
                dxi, dpc, dd, dt = dxi + exi, dpc + epc, dd + ed, dt + et
                GV = {k: [a + b for a, b in zip(GV[k], GV2[k])] for k in S}
            dS = {k: [-rp - gv for rp, gv in zip(RP[k], GV[k])] for k in S}
            dZ = {k: [q + w * gv for q, w, gv in zip(Qd[k], W[k], GV[k])] for k in S}
            if verbose:
                Zf = {k: [zh + w * gv for zh, w, gv in zip(Zh[k], W[k], GV[k])] for k in S}
                gi, gs, gt_ = grad(Zf, False)
                lam_ = np.cumsum(gs[::-1], axis=0)[::-1]
                res = gi + lam_ @ Bm
                print("   newton residual: inputs", np.abs(res).max(axis=0), "t %.2e" % np.abs(gt_).max(),
                      " |rhs| %.2e" % np.abs(grad(Zh, False)[0]).max(), "da", np.abs(np.hstack((dxi,dpc,dd))).max(axis=0), "x-cumsum", np.abs(GV["soc"][0]).max(), "wmax %.1e" % max(x.max() for k in W for x in W[k]))
            return (dxi, dpc, dd, dt), dS, dZ

        def max_step(X, dX):
            a = np.inf
            for k in X:
                for x, dx in zip(X[k], dX[k]):
                    neg = dx < 0
                    if neg.any():
                        a = min(a, (-x[neg] / dx[neg]).min())
            return a

        RC = {k: [-s * z for s, z in zip(S[k], Z[k])] for k in S}
        (dxi, dpc, dd, dt), dS, dZ = solve(RC)
        ap, ad = min(1.0, max_step(S, dS)), min(1.0, max_step(Z, dZ))
        if final:   # last step: the affine-scaling (pure Newton) step as far as the rows allow
            ap = min(1.0, 0.99999 * max_step(S, dS))
            xi, pc, d, t = xi + ap * dxi, pc + ap * dpc, d + ap * dd, t + ap * dt
            rv, u = row_values(xi, pc, d, t)
            obj = float((cu * u).sum() + (ct[:, None] * t).sum() + (wgt * (red.sg_l @ pl)).sum())
            break
        mu_aff = sum(((s + ap * ds) * (z + ad * dz)).sum() for k in S for s, ds, z, dz in zip(S[k], dS[k], Z[k], dZ[k])) / m_rows
        sg_ = min(1.0, max((mu_aff / mu) ** 3, 0.1 * tol * (1 + abs(obj)) / mu))
        RC = {k: [sg_ * mu - s * z - ds * dz for s, z, ds, dz in zip(S[k], Z[k], dS[k], dZ[k])] for k in S}
        (dxi, dpc, dd, dt), dS, dZ = solve(RC, n_ref)
        eta = float(os.environ.get("ETA", "0.99"))
        ap, ad = min(1.0, eta * max_step(S, dS)), min(1.0, eta * max_step(Z, dZ))
        xi, pc, d, t = xi + ap * dxi, pc + ap * dpc, d + ap * dd, t + ap * dt
        S = {k: [s + ap * ds for s, ds in zip(S[k], dS[k])] for k in S}
        Z = {k: [z + ad * dz for z, dz in zip(Z[k], dZ[k])] for k in Z}
    p_dev0 = np.zeros(red.n_dev)
    p_dev0[red.loads], p_dev0[red.gens], p_dev0[red.des] = pl[:, 0], u[0, :ng], u[0, ng:]
    p_dev0[red.slack_dev] = red.sg_l @ pl[:, 0] + red.sg_c @ u[0]
    ipm_struct.last = dict(mu=mu, rp=rpmax, rd=rdmax)
    return obj, p_dev0, it


TOL = float(os.environ.get("TOL", "1e-9"))
NREF = int(os.environ.get("NREF", "0"))


def main():
    m = NetworkModel(networks.anm6_network(), 0.25, 100)
    g = np.load(os.path.join(ROOT, "tests", "golden", "mpc_anm6.npz"))
    worst = 0
    for k in range(len(g["N"])):
        N, margin, gamma = int(g["N"][k]), float(g["safety_margin"][k]), float(g["gamma"])
        red = ReducedDCOPF(m, gamma, margin, N)
        obj = g["c%d_objective" % k]
        its, errs, aerr = [], [], []
        for e in range(len(obj)):
            pl, pg, soc = g["c%d_load" % k][e], g["c%d_gen" % k][e], g["c%d_soc" % k][e]
            f, pd, it = ipm_struct(red, pl, pg, soc, tol=TOL, n_ref=NREF)
            its.append(it)
            errs.append(abs(f - obj[e]) / (1 + abs(obj[e])))
            aerr.append(np.abs(pd - g["c%d_p_dev" % k][e]).max())
        print("config %d N=%2d: iterations %d..%d, value err max %.2e, first-stage p_dev err max %.2e (median %.1e)"
              % (k, N, min(its), max(its), max(errs), max(aerr), np.median(aerr)))
        worst = max(worst, max(errs))
    print("worst relative value error", worst)


if __name__ == "__main__":
    main()
