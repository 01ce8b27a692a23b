"""Batched MPC DC-OPF policy (gym_anm/agents/mpc.py:163-388).

Golden vectors (tests/golden/mpc_anm6.npz, oracle/make_golden_mpc.py): the UNMODIFIED reference agents
(MPCAgentPerfect / MPCAgentConstant, N = 1, 3, 10, 20) in closed loop on the reference's ANM6Easy, their
linear program -- built by the reference's own code -- solved by HiGHS behind a stand-in for cvxpy's modelling
API (cvxpy is not in the image; the reference's own tests of this path pass under the stand-in).  Pinned: the
optimal VALUE of every recorded program (solver-independent) and the optimality, in the oracle's program, of
the first-stage minimiser the reference's run returned.

The product path under test is ``anm_mpc_solve_f64`` (the hand-written interior-point solver of
gym_anm_amd/csrc/anm_mpc.hpp): in the CPU tier the very same solver source compiled for the host (tests/hostsim,
one thread per lane), in the GPU tier the gfx950 kernel.  For all 300 recorded programs: the value within 1e-7
(relative) of the reference's, the returned primal solution feasible for the reference's rows (1e-8 p.u.), and the
first-stage set-point of every device equal to the oracle's (1e-8 p.u. = 1e-6 MW) WHEREVER THE MINIMISER IS UNIQUE
(two more linear programs per device on the optimal face decide that)."""
import os

import numpy as np
import pytest
import torch

import anm_oracle as O
import mpc_oracle as MO
from gym_anm_amd import networks
from gym_anm_amd.agents import MPCAgentConstant, MPCAgentPerfect
from gym_anm_amd.agents.dcopf import ReducedDCOPF
from gym_anm_amd.agents.mpc import BatchedDCOPF, DCOPFProgram
from gym_anm_amd.envs import ANM6EasyVec
from gym_anm_amd.model import NetworkModel
from gym_anm_amd.simulator import BatchedSimulator

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_anm6.npz")


def _golden_configs():
    g = np.load(GOLDEN)
    for k in range(len(g["N"])):
        yield k, str(g["kind"][k]), int(g["N"][k]), float(g["safety_margin"][k]), float(g["gamma"]), g


def _host_sim(net, num_envs=2):
    from hostsim_backend import hostsim_backend

    be = hostsim_backend(NetworkModel(net, 0.25, 100).topology())
    return BatchedSimulator(net, 0.25, 100, num_envs=num_envs, device="cpu", _backend=be)


def _gpu_sim(net, num_envs=2):
    return BatchedSimulator(net, 0.25, 100, num_envs=num_envs, device="cuda:0")


# ---------------------------------------------------------------------------------------------------------------
# the oracle and the reduction, without any solver of ours
# ---------------------------------------------------------------------------------------------------------------
def test_oracle_program_has_the_value_of_the_reference_program():
    """oracle/mpc_oracle.py (the restatement the kernel is checked against) vs the program the reference's own
    code builds: same optimal value on all 300 recorded cases, and the reference run's first-stage minimiser is
    optimal for the oracle's program as well (fixing it does not cost anything)."""
    n = O.parse_network(networks.anm6_network(), 0.25, 100)
    for k, kind, N, margin, gamma, g in _golden_configs():
        obj = g["c%d_objective" % k]
        for e in range(len(obj)):
            pl, pg, soc = g["c%d_load" % k][e], g["c%d_gen" % k][e], g["c%d_soc" % k][e]
            ref = MO.solve_dcopf(n, pl, pg, soc, gamma, margin, N)
            assert ref["status"] == 0
            assert abs(ref["objective"] - obj[e]) <= 1e-9 * (1 + abs(obj[e])), (kind, N, e, ref["objective"], obj[e])
            if e % 6 == 0:
                fixed = MO.solve_dcopf(n, pl, pg, soc, gamma, margin, N, first_stage_p_dev=g["c%d_p_dev" % k][e])
                assert fixed["status"] == 0
                assert abs(fixed["objective"] - obj[e]) <= 1e-7 * (1 + abs(obj[e])), (kind, N, e)


def test_golden_actions_are_the_clipped_first_stage_set_points():
    """mpc.py:385-391, 338-341: action = clip([P_gen.., 0.., P_des.., 0..] * baseMVA, Box)."""
    m = NetworkModel(networks.anm6_network(), 0.25, 100)
    for k, kind, N, margin, gamma, g in _golden_configs():
        p = g["c%d_p_dev" % k]
        a = g["c%d_action" % k]
        raw = np.concatenate((p[:, m.gen_idx] * m.baseMVA, np.zeros((len(p), len(m.gen_idx))),
                              p[:, m.des_idx] * m.baseMVA, np.zeros((len(p), len(m.des_idx)))), 1)
        assert np.all((a == raw) | (np.abs(a - raw) < 1e-6))  # clipping only trims solver noise


def test_reduced_program_is_the_reference_program():
    """gym_anm_amd/agents/dcopf.py (balance and loads eliminated, generators scaled by their interval, state of
    charge as the state): spelled out densely and handed to HiGHS it has the reference's optimal value."""
    from scipy.optimize import linprog

    m = NetworkModel(networks.anm6_network(), 0.25, 100)
    for k, kind, N, margin, gamma, g in _golden_configs():
        red = ReducedDCOPF(m, gamma, margin, N)
        for e in range(0, 60, 7):
            G, h, c, c0 = red.dense_program(g["c%d_load" % k][e], g["c%d_gen" % k][e], g["c%d_soc" % k][e])
            res = linprog(c, A_ub=G, b_ub=h, bounds=[(None, None)] * len(c), method="highs")
            ref = g["c%d_objective" % k][e]
            assert res.status == 0 and abs(res.fun + c0 - ref) <= 1e-8 * (1 + abs(ref)), (kind, N, e, res.fun + c0, ref)


def test_program_rows_follow_the_reference_constraints():
    """Row counts of mpc.py:232-313 for ANM6, N = 2: per stage 6 balances, 3 loads, 2 x 2 generator rows,
    5 storage rows, 6 + 1 angle rows, 3 x 5 branch rows."""
    pr = DCOPFProgram(NetworkModel(networks.anm6_network(), 0.25, 100), 0.995, 0.9, 2)
    assert pr.n == 2 * (6 + 7 + 1 + 1 + 5) and pr.m == 2 * (6 + 3 + 4 + 5 + 7 + 15)
    assert pr.n_param == 2 * 5 + 1
    # objective: slack + nothing else is a classical generator in ANM6; overload epigraphs weighted by lamb
    assert pr.q[6 + 0] == 1.0 and pr.q[6 + 2] == 0.0 and pr.q[6 + 7 + 2] == 100.0 and pr.q[20 + 6] == 0.995


# ---------------------------------------------------------------------------------------------------------------
# the solver behind the C ABI
# ---------------------------------------------------------------------------------------------------------------
def _table_slices(red):
    """the layout of mpc::Sz<Topo> (csrc/anm_mpc.hpp) restated"""
    p1 = lambda n: max(n, 1)  # noqa: E731
    nb1, nc, nl, nbr, ng, ns = red.nb - 1, red.nc, red.nl, red.nbr, red.ng, red.ns
    sizes = [("thc", nb1 * nc), ("thl", nb1 * nl), ("phc", nbr * nc), ("phl", nbr * nl), ("cost", p1(nc)), ("sgl", p1(nl)),
             ("lim", p1(nbr)), ("gpmin", p1(ng)), ("gpmax", p1(ng)), ("spmin", p1(ns)), ("spmax", p1(ns)),
             ("socmin", p1(ns)), ("socmax", p1(ns)), ("bc", p1(ns)), ("bd", p1(ns)), ("lamb", 1), ("wgt", 64)]
    out, o = {}, 0
    for name, n in sizes:
        out[name] = slice(o, o + n)
        o += n
    return out, o


def check_tables(sim, gamma=0.97, margin=0.93, N=5):
    """anm_mpc_create's DC reduction (C++, from the network description) == gym_anm_amd/agents/dcopf.py (NumPy)"""
    sol = BatchedDCOPF(sim, gamma, margin, N)
    red = ReducedDCOPF(sim.model, gamma, margin, N)
    tab = sol.tables()
    sl, total = _table_slices(red)
    assert total == sol.dims.table_doubles == len(tab)
    m = sim.model
    ctrl = red.gens + red.des
    pmin, pmax = np.asarray(m.dev_p_min, float), np.asarray(m.dev_p_max, float)
    bound = np.max(np.abs(red.Th_c[red.theta_rows]) @ np.maximum(np.abs(pmin[ctrl]), np.abs(pmax[ctrl]))
                   + np.abs(red.Th_l[red.theta_rows]) @ np.abs(pmin[red.loads])) if len(red.theta_rows) else 0.0
    assert abs(sol.dims.angle_bound - bound) <= 1e-12 * (1 + bound) and sol.dims.angle_rows == int(bound >= 0.98 * np.pi)
    assert sol.dims.n_stage_rows == red.nr - (0 if sol.dims.angle_rows else 2 * len(red.theta_rows))
    assert sol.dims.n_stage_vars == red.nv and sol.dims.n_ctrl == red.nc
    cmp = lambda name, ref: np.testing.assert_allclose(tab[sl[name]][: np.size(ref)], np.ravel(ref), rtol=1e-12, atol=1e-14, err_msg=name)  # noqa: E731
    cmp("thc", red.Th_c[red.theta_rows])
    cmp("thl", red.Th_l[red.theta_rows])
    cmp("phc", red.Ph_c)
    cmp("phl", red.Ph_l)
    cmp("cost", red.cost_c)
    cmp("sgl", red.sg_l)
    cmp("lim", red.lim)
    cmp("gpmin", red.g_pmin)
    cmp("gpmax", red.g_pmax)
    cmp("spmin", red.s_pmin)
    cmp("spmax", red.s_pmax)
    cmp("socmin", red.soc_min)
    cmp("socmax", red.soc_max)
    cmp("bc", red.dt * red.eff)
    cmp("bd", red.dt / red.eff)
    cmp("lamb", [max(red.lamb, 1e-30)])
    cmp("wgt", np.maximum(gamma ** np.arange(64), 1e-30))


def full_solution(red, pr, pl, pg, sol):
    """[E, N, nv] (P_g, p_c, d, t per stage) -> the variables of the reference's program, [E, pr.n]"""
    E, N = sol.shape[0], red.N
    ng, ns, nbr = red.ng, red.ns, red.nbr
    x = np.zeros((E, pr.n))
    for i in range(N):
        o = i * pr.per
        P_g, p_c, d, t = sol[:, i, :ng], sol[:, i, ng : ng + ns], sol[:, i, ng + ns : ng + 2 * ns], sol[:, i, ng + 2 * ns :]
        u = np.concatenate((P_g, d - p_c), 1)
        x[:, o + pr.off["theta"] : o + pr.off["theta"] + red.nb] = pl[:, :, i] @ red.Th_l.T + u @ red.Th_c.T
        pd = np.zeros((E, red.n_dev))
        pd[:, red.loads], pd[:, red.gens], pd[:, red.des] = pl[:, :, i], P_g, d - p_c
        pd[:, red.slack_dev] = pl[:, :, i] @ red.sg_l + u @ red.sg_c
        x[:, o + pr.off["p_dev"] : o + pr.off["p_dev"] + red.n_dev] = pd
        x[:, o + pr.off["p_c"] : o + pr.off["p_c"] + ns] = p_c
        x[:, o + pr.off["p_d"] : o + pr.off["p_d"] + ns] = d
        x[:, o + pr.off["s"] : o + pr.off["s"] + nbr] = t
    return x


def check_golden(sim, every=1):
    """all recorded programs through anm_mpc_solve_f64: value, feasibility, and the action where it is unique"""
    net = networks.anm6_network()
    m = sim.model
    n = O.parse_network(net, 0.25, 100)
    ctrl = list(m.gen_idx) + list(m.des_idx)
    n_unique = n_checked = 0
    for k, kind, N, margin, gamma, g in _golden_configs():
        pl, pg, soc = g["c%d_load" % k][::every], g["c%d_gen" % k][::every], g["c%d_soc" % k][::every]
        ref = g["c%d_objective" % k][::every]
        solver = BatchedDCOPF(sim, gamma, margin, N, keep_solution=True)
        u0 = solver.solve(pl, pg, soc).cpu().numpy()
        obj, iters = solver.objective.cpu().numpy(), solver.iters.cpu().numpy()
        assert iters.max() < solver.max_iter, (kind, N, iters.max())
        assert float(solver.info[:, 2].max()) <= 1e-7 * 101, (kind, N, solver.info.max(0))  # dual residual vs the costs (1, lambda)
        err = np.abs(obj - ref) / (1 + np.abs(ref))
        assert err.max() <= 1e-7, (kind, N, err.max(), int(err.argmax()))
        # the primal solution satisfies the rows of the reference's program
        pr, red = DCOPFProgram(m, gamma, margin, N), ReducedDCOPF(m, gamma, margin, N)
        x = full_solution(red, pr, pl, pg, solver.solution.cpu().numpy())
        E = len(ref)
        params = np.concatenate((pl.transpose(0, 2, 1).reshape(E, -1), pg.transpose(0, 2, 1).reshape(E, -1), soc), 1)
        l, u = (b.numpy() for b in pr.bounds(torch.as_tensor(params)))
        Ax = x @ pr.A.T
        assert (l - Ax).max() <= 1e-8 and (Ax - u).max() <= 1e-8, (kind, N, (l - Ax).max(), (Ax - u).max())
        np.testing.assert_allclose(x @ pr.q, obj, rtol=0, atol=1e-9 * (1 + np.abs(obj).max()))
        # the first-stage set-points, wherever the oracle says they are unique
        for e in range(E):
            val, rng = MO.first_stage_range(n, pl[e], pg[e], soc[e], gamma, margin, N, ctrl)
            for c, (lo, hi) in enumerate(rng):
                n_checked += 1
                if hi - lo <= 1e-9:
                    n_unique += 1
                    assert abs(u0[e, c] - 0.5 * (lo + hi)) <= 1e-8, (kind, N, e, c, u0[e, c], lo, hi)
                else:  # anywhere on the optimal face is right
                    assert lo - 1e-7 <= u0[e, c] <= hi + 1e-7, (kind, N, e, c, u0[e, c], lo, hi)
    assert n_unique > n_checked // 8, (n_unique, n_checked)  # (a good part of the recorded set-points is unique)


def check_random_programs(sim, net, n_cases, seed, horizons=(1, 2, 5, 12, 33), **solver_kw):
    """random forecasts, horizons, margins, discounts and states of charge (also exactly at the bounds) vs HiGHS"""
    n = O.parse_network(net, 0.25, 100)
    m = sim.model
    rng = np.random.default_rng(seed)
    nl, ng, ns = len(m.load_idx), len(m.gen_idx), len(m.des_idx)
    for N in horizons:
        margin, gamma = float(rng.choice([0.8, 0.9, 1.0])), float(rng.choice([0.9, 0.995, 1.0]))
        solver = BatchedDCOPF(sim, gamma, margin, N, **solver_kw)
        pl = -rng.uniform(0, 1, (n_cases, nl, N)) * (-m.dev_p_min[m.load_idx])[None, :, None]
        pg = rng.uniform(0, 1, (n_cases, ng, N)) * m.dev_p_max[m.gen_idx][None, :, None] * (rng.random((n_cases, ng, N)) > 0.25)
        lo_, hi_ = m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx]
        pick = rng.integers(0, 4, (n_cases, ns))
        soc = np.where(pick == 0, lo_, np.where(pick == 1, hi_, lo_ + (hi_ - lo_) * rng.random((n_cases, ns))))
        solver.solve(pl, pg, soc)
        obj, iters = solver.objective.cpu().numpy(), solver.iters.cpu().numpy()
        for e in range(n_cases):
            ref = MO.solve_dcopf(n, pl[e], pg[e], soc[e], gamma, margin, N)
            assert ref["status"] == 0
            assert iters[e] < solver.max_iter and float(solver.info[e, 2]) <= 1e-5
            assert abs(obj[e] - ref["objective"]) <= 2e-7 * (1 + abs(ref["objective"])), (N, e, obj[e], ref["objective"], iters[e])


def two_storage_network():
    """5 buses, a loop, a classical generator (it costs), a renewable one and TWO storage units of different
    efficiency: the n_des x n_des Riccati blocks and the classical generator's cost term"""
    return {
        "baseMVA": 100,
        "bus": np.array([[0, 0, 132, 1.04, 1.04], [1, 1, 33, 1.1, 0.9], [2, 1, 33, 1.1, 0.9], [3, 1, 33, 1.1, 0.9],
                         [4, 1, 33, 1.1, 0.9]], dtype=float),
        "device": np.array([
            [0, 0, 0, None, 200, -200, 200, -200, None, None, None, None, None, None, None],
            [1, 1, -1, 0.2, 0, -20, None, None, None, None, None, None, None, None, None],
            [2, 2, 1, None, 25, 0, 20, -20, 15, None, 10, -10, None, None, None],
            [3, 3, 2, None, 40, 0, 30, -30, 25, None, 12, -12, None, None, None],
            [4, 3, 3, None, 20, -20, 20, -20, 12, -12, 8, -8, 60, 0, 0.92],
            [5, 4, -1, 0.25, 0, -30, None, None, None, None, None, None, None, None, None],
            [6, 4, 3, None, 15, -15, 15, -15, 10, -10, 6, -6, 40, 5, 0.85],
        ], dtype=object),
        "branch": np.array([[0, 1, 0.01, 0.08, 0.0, 45, 1, 0], [1, 2, 0.03, 0.06, 0.0, 18, 1, 0],
                            [1, 3, 0.04, 0.07, 0.0, 22, 1, 0], [2, 4, 0.05, 0.09, 0.0, 15, 1, 0],
                            [3, 4, 0.04, 0.08, 0.0, 12, 1, 0]], dtype=float),
    }


def check_closed_loop(env, agent_cls, N, steps, n_env):
    ag = agent_cls(env.simulator, env.action_space, env.gamma, safety_margin=0.92, planning_steps=N)
    tot = torch.zeros(n_env, dtype=torch.float64, device=env.device)
    for _ in range(steps):
        a = ag.act(env)
        assert a.shape == (n_env, 6) and bool((a[:, 2:4] == 0).all()) and bool((a[:, 5] == 0).all())  # Q set-points are 0
        assert bool(ag.last_converged.all())
        assert env.action_space.contains(a[0].cpu().numpy())
        _, r, term, _, _ = env.step(a)
        tot += r
    assert not bool(env.terminated.any()) and float(tot.mean()) / steps > -5.0  # random agent: about -175 per step


def check_fused_act(env, n_env, steps=5, horizons=(1, 3)):
    """MPCAgent.act() as ONE launch (anm_mpc_act_f64: forecasts gathered inside the kernel, the clipped MW action row
    written by it; mpc.py:321-346, mpc_constant.py:24-35, mpc_perfect.py:24-40) against the same agent going through
    forecast() -> anm_mpc_solve_f64 -> scaling / clipping in torch: identical bits, step after step in closed loop."""
    for cls in (MPCAgentConstant, MPCAgentPerfect):
        class Unfused(cls):
            def _fused(self, env):
                return False

        for N in horizons:
            fused = cls(env.simulator, env.action_space, env.gamma, safety_margin=0.92, planning_steps=N)
            plain = Unfused(env.simulator, env.action_space, env.gamma, safety_margin=0.92, planning_steps=N)
            assert fused._fused(env) and not plain._fused(env)
            kept = []   # actions a caller keeps across steps stay what they were (the reference returns a fresh array)
            for _ in range(steps):
                a = fused.act(env)
                kept.append((a, a.clone()))
                b = plain.act(env)
                assert a.shape == b.shape == (n_env, env.action_space.shape[0])
                assert torch.equal(a, b), (cls.__name__, N, float((a - b).abs().max()))
                for name in ("u0", "objective", "iters", "info"):
                    assert torch.equal(getattr(fused.solver, name), getattr(plain.solver, name)), name
                assert bool(fused.last_converged.all())
                env.step(a.clone())
            assert all(torch.equal(x, y) for x, y in kept) and len({x.data_ptr() for x, _ in kept}) == len(kept)
            fused.reuse_action_buffer = True    # opt-in: the solver's own buffer, overwritten by the next call
            assert fused.act(env).data_ptr() == fused.act(env).data_ptr()


# ---- CPU tier: the solver source compiled for the host --------------------------------------------------------
def test_fused_act_equals_the_unfused_path_on_the_host_double():
    from hostsim_backend import hostsim_backend

    be = hostsim_backend(NetworkModel(networks.anm6_network(), 0.25, 100).topology())
    env = ANM6EasyVec(num_envs=6, device="cpu", seed=5, _backend=be)
    env.reset(seed=5)
    check_fused_act(env, 6, steps=3)


def test_dc_reduction_tables_host():
    check_tables(_host_sim(networks.anm6_network()))
    check_tables(_host_sim(two_storage_network()), N=3)


def test_solver_on_every_golden_program_host():
    check_golden(_host_sim(networks.anm6_network()), every=2)


def test_solver_vs_highs_random_programs_host():
    check_random_programs(_host_sim(networks.anm6_network()), networks.anm6_network(), 6, 11)


def test_two_storage_units_and_a_classical_generator_host():
    net = two_storage_network()
    check_random_programs(_host_sim(net), net, 5, 3, horizons=(1, 3, 8))


def check_other_stock_networks(make_sim, n_cases):
    """the other networks the package ships that fit the kernel's register budget.  The 3-bus loop in its own
    per-unit base (base_mva = 1: injections of tens of p.u.) is the case whose ANGLE rows |theta| <= pi bind at the
    optimum and are violated at the centred start: the start is scaled back until it is interior"""
    for net, horizons in [(networks.two_bus_network(), (1, 4)), (networks.three_bus_loop_network(), (1, 2, 5, 12)),
                          (networks.three_bus_loop_network(base_mva=10), (1, 5)), (networks.synthetic_radial_network(8, seed=1), (1, 3))]:
        check_random_programs(make_sim(net), net, n_cases, 3, horizons=horizons)


def test_other_stock_networks_host():
    check_other_stock_networks(_host_sim, 4)


def check_angle_rows(make_sim):
    """the rows |theta| <= pi (mpc.py:291-292) ride through the solve only where an angle can come near pi within the
    device limits; left out, the solution's angles are checked (a solution of the relaxed program that satisfies the
    dropped rows solves the full one)"""
    rng = np.random.default_rng(8)
    # ANM6: no angle beyond 0.3 rad whatever the devices do -> relaxed by default; carrying the rows changes nothing
    sim = make_sim(networks.anm6_network())
    m = sim.model
    N, E = 4, 6
    pl = -rng.uniform(0, 1, (E, len(m.load_idx), N)) * (-m.dev_p_min[m.load_idx])[None, :, None]
    pg = rng.uniform(0, 1, (E, len(m.gen_idx), N)) * m.dev_p_max[m.gen_idx][None, :, None]
    soc = rng.uniform(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx], (E, len(m.des_idx)))
    sols = {}
    for rows in (None, True):
        s_ = BatchedDCOPF(sim, 0.995, 0.9, N, keep_solution=True, angle_rows=rows)
        assert s_.dims.angle_rows == 0 and s_.dims.angle_bound < 1.0
        s_.solve(pl, pg, soc)
        assert int(s_.iters.max()) < s_.max_iter and float(s_.info[:, 1].max()) == 0.0
        sols[rows] = (s_.objective.cpu().numpy().copy(), s_.u0.cpu().numpy().copy())
    np.testing.assert_allclose(sols[None][0], sols[True][0], rtol=1e-8, atol=1e-8)
    # the 3-bus loop in its own per-unit base: angles reach pi -> the rows are carried by default; forced out, the
    # programs whose optimum sits AT an angle limit are reported (info 2, not converged), the others agree
    net = networks.three_bus_loop_network()
    sim = make_sim(net)
    m = sim.model
    n = O.parse_network(net, 0.25, 100)
    N, E = 2, 8
    pl = -rng.uniform(0, 1, (E, len(m.load_idx), N)) * (-m.dev_p_min[m.load_idx])[None, :, None]
    pg = rng.uniform(0, 1, (E, len(m.gen_idx), N)) * m.dev_p_max[m.gen_idx][None, :, None]
    soc = rng.uniform(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx], (E, len(m.des_idx)))
    full = BatchedDCOPF(sim, 0.995, 0.9, N)
    assert full.dims.angle_rows == 1 and full.dims.angle_bound > np.pi
    full.solve(pl, pg, soc)
    relaxed = BatchedDCOPF(sim, 0.995, 0.9, N, angle_rows=False)
    relaxed.solve(pl, pg, soc)
    flagged = relaxed.info[:, 1].cpu().numpy() == 2.0
    assert flagged.any() and (relaxed.iters.cpu().numpy()[flagged] == relaxed.max_iter).all()
    for e in range(E):
        ref = MO.solve_dcopf(n, pl[e], pg[e], soc[e], 0.995, 0.9, N)
        assert abs(float(full.objective[e]) - ref["objective"]) <= 2e-7 * (1 + abs(ref["objective"]))
        if not flagged[e]:
            assert abs(float(relaxed.objective[e]) - ref["objective"]) <= 2e-7 * (1 + abs(ref["objective"]))
        else:
            assert float(relaxed.objective[e]) < ref["objective"] - 1e-6   # the relaxed program's value: below the full one's


def test_network_beyond_the_register_budget_host():
    net = networks.synthetic_radial_network(30, 0)
    check_random_programs(_host_sim(net), net, 3, 3, horizons=(1, 3, 6))


def check_discount_extremes(make_sim):
    """gamma = 0 (the reference accepts [0, 1]: later stages then carry no cost at all) and discounts that underflow
    over the horizon: the value of the program is still the reference's"""
    net = networks.anm6_network()
    sim, n = make_sim(net), O.parse_network(net, 0.25, 100)
    m, rng = sim.model, np.random.default_rng(4)
    for gamma, N in ((0.0, 1), (0.0, 4), (0.0, 10), (1e-20, 5), (0.01, 12)):
        E = 4
        s_ = BatchedDCOPF(sim, gamma, 0.9, N)
        pl = -rng.uniform(0, 1, (E, len(m.load_idx), N)) * (-m.dev_p_min[m.load_idx])[None, :, None]
        pg = rng.uniform(0, 1, (E, len(m.gen_idx), N)) * m.dev_p_max[m.gen_idx][None, :, None]
        soc = rng.uniform(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx], (E, len(m.des_idx)))
        s_.solve(pl, pg, soc)
        assert int(s_.iters.max()) < s_.max_iter and float(s_.info[:, 1].max()) == 0.0, (gamma, N)
        for e in range(E):
            ref = MO.solve_dcopf(n, pl[e], pg[e], soc[e], gamma, 0.9, N)
            assert abs(float(s_.objective[e]) - ref["objective"]) <= 2e-7 * (1 + abs(ref["objective"])), (gamma, N, e)


def test_discount_extremes_host():
    check_discount_extremes(_host_sim)


def test_angle_rows_host():
    check_angle_rows(_host_sim)


def test_closed_loop_on_the_host_double():
    from hostsim_backend import hostsim_backend

    be = hostsim_backend(NetworkModel(networks.anm6_network(), 0.25, 100).topology())
    env = ANM6EasyVec(num_envs=8, device="cpu", seed=3, _backend=be)
    env.reset(seed=3)
    check_closed_loop(env, MPCAgentConstant, 1, 12, 8)
    check_closed_loop(env, MPCAgentPerfect, 6, 6, 8)


def test_reference_example_mpc_constant_pattern_on_the_host_double():
    """examples/mpc_constant.py of the reference: `agent = MPCAgentConstant(env.simulator, env.action_space, env.gamma,
    safety_margin=0.96, planning_steps=N); a = agent.act(env); env.step(a)` on the NumPy-facing ANM6Easy."""
    from hostsim_backend import hostsim_backend

    from gym_anm_amd import MPCAgentConstant as TopLevel
    from gym_anm_amd.envs import ANM6Easy

    assert TopLevel is MPCAgentConstant
    net = networks.anm6_network()
    env = ANM6Easy(device="cpu", _backend=hostsim_backend(NetworkModel(net, 0.25, 100).topology()))
    o, _ = env.reset(seed=3)
    agent = MPCAgentConstant(env.simulator, env.action_space, env.gamma, safety_margin=0.96, planning_steps=10)
    total = 0.0
    for t in range(4):
        a = agent.act(env)
        assert isinstance(a, np.ndarray) and a.shape == (6,) and env.action_space.contains(a)
        o, r, terminated, _, _ = env.step(a)
        assert not terminated and isinstance(r, float)
        total += r
    assert total > -20.0  # the random agent loses ~175 per step on this task; the MPC policy a fraction of one


def test_unconverged_solves_are_reported():
    """mpc.py:377-379 prints a non-optimal status and goes on; here: a mask and a warning"""
    sim = _host_sim(networks.anm6_network())
    env_like = type("E", (), {})()
    ag = MPCAgentConstant(sim, type("B", (), {"low": -np.ones(6) * 1e3, "high": np.ones(6) * 1e3})(), 0.995, planning_steps=2,
                          max_iter=3)
    env_like.state = torch.tensor([[0, -5, 10, -10, 20, -15, 0] + [0] * 7 + [50, 30, 35, 0.0]] * 2, dtype=torch.float64)
    env_like.simulator = sim
    sim.soc.fill_(0.5)
    with pytest.warns(UserWarning, match="did not reach the tolerance"):
        ag.act(env_like)
    assert not bool(ag.last_converged.any())


# ---- GPU tier: the gfx950 kernel ------------------------------------------------------------------------------
@pytest.mark.gpu
def test_dc_reduction_tables_gpu():
    check_tables(_gpu_sim(networks.anm6_network()))


@pytest.mark.gpu
def test_solver_on_every_golden_program_gpu():
    check_golden(_gpu_sim(networks.anm6_network()))


@pytest.mark.gpu
def test_solver_vs_highs_random_programs_gpu():
    check_random_programs(_gpu_sim(networks.anm6_network()), networks.anm6_network(), 24, 11, horizons=(1, 2, 3, 5, 8, 12, 20, 33, 64))


@pytest.mark.gpu
def test_two_storage_units_and_a_classical_generator_gpu():
    net = two_storage_network()
    check_random_programs(_gpu_sim(net), net, 16, 3, horizons=(1, 3, 8, 16))


def test_mpc_only_library_builds_and_exports_the_mpc_entry_points():
    """networks above 12 buses step on the table-driven kernels of some other library; their MPC kernel comes from a
    small library of its own (csrc/anm_mpc_only.hip), cross-compiled here, loaded, symbols and signature checked"""
    import ctypes as C

    from gym_anm_amd import _lib, codegen

    topo = NetworkModel(networks.synthetic_meshed_network(14, 5, 3), 0.25, 100).topology()
    path = codegen.build_library(topo, mpc_only=True)
    assert os.path.basename(path).startswith("libmpc_")
    cdll = C.CDLL(path)
    for name in _lib.MPC_ABI:
        assert hasattr(cdll, name), name
    cdll.anm_topology_signature.restype = C.c_char_p
    assert cdll.anm_topology_signature().decode() == codegen.topology_signature(topo)
    assert not hasattr(cdll, "anm_step_f64")


@pytest.mark.gpu
def test_mpc_of_a_network_that_steps_in_generic_mode_gpu():
    """a meshed 20-bus network: the simulator runs on the general lane-group kernel of whatever library is built
    (generic mode), the MPC agent on the MPC-only library of the network's own topology"""
    net = networks.synthetic_meshed_network(20, 3, 6)
    sim = _gpu_sim(net)
    assert sim.backend.generic and sim.impl == "mesh"
    check_random_programs(sim, net, 6, 3, horizons=(1, 4, 10))


@pytest.mark.gpu
@pytest.mark.parametrize("n_bus,seed,n_chords,cls", [(14, 0, 3, "s1"), (14, 2, 0, "s0"), (16, 6, 3, "s2"), (18, 4, 3, "l1"), (24, 7, 3, "l0"),
                                                     (24, 2, 0, "l2")])
def test_mpc_size_classes_without_a_compiler_gpu(monkeypatch, n_bus, seed, n_chords, cls):
    """`k_mpc` depends on a network through its SIZES only (the topology lives in the tables the host computes): a network
    that steps in generic mode gets its MPC agent from a precompiled size class it is padded into (extra loads with zero
    coefficients, generators with an empty range, branches without flow) -- with hipcc hidden.  Random programs on
    networks nobody compiled anything for, against HiGHS (`agents/mpc.py:163-319` handles any network through cvxpy)."""
    from gym_anm_amd import _lib, codegen

    monkeypatch.setattr(codegen, "hipcc_path", lambda: None)
    net = networks.synthetic_meshed_network(n_bus, seed, n_chords) if n_chords else networks.synthetic_radial_network(n_bus, seed)
    sim = _gpu_sim(net)
    assert sim.backend.generic
    solver = BatchedDCOPF(sim, 0.97, 0.93, 3)
    assert solver.backend.size_class == cls and os.path.basename(solver.backend.path).startswith("libmpc_class_")
    check_random_programs(sim, net, 5, seed, horizons=(1, 3, 8))


@pytest.mark.gpu
def test_mpc_size_class_equals_the_kernel_of_the_topology_gpu():
    """ANM6 through the size class `s1` (8 loads, 2 generators, 1 storage unit, 18 buses, 20 branches: 15 padded
    branches, 5 padded loads) next to the kernel compiled for ANM6: every recorded program of the reference's agents to
    the tolerance of the golden set, and the fused act() in closed loop."""
    sim = _gpu_sim(networks.anm6_network())
    net = networks.anm6_network()
    check_random_programs(sim, net, 12, 7, horizons=(1, 2, 5, 12), size_class=True)
    for k, kind, N, margin, gamma, g in _golden_configs():
        pl, pg, soc, ref = g["c%d_load" % k], g["c%d_gen" % k], g["c%d_soc" % k], g["c%d_objective" % k]
        own, cls = BatchedDCOPF(sim, gamma, margin, N), BatchedDCOPF(sim, gamma, margin, N, size_class=True)
        assert cls.backend.size_class == "s1" and own.backend.size_class is None
        own.solve(pl, pg, soc)
        cls.solve(pl, pg, soc)
        obj = cls.objective.cpu().numpy()
        assert int(cls.iters.max()) < cls.max_iter and bool(cls.converged.all())
        assert (np.abs(obj - ref) / (1 + np.abs(ref))).max() <= 1e-7, (kind, N)
        # (the set-points themselves: equal where the minimiser is unique -- check_golden decides that for the topology's
        # own kernel --, anywhere on the optimal face else: the padded rows move the central path)
        assert float(((own.objective - cls.objective).abs() / (1 + own.objective.abs())).max()) <= 2e-7
    env = ANM6EasyVec(num_envs=512, device="cuda:0", seed=11)
    env.reset(seed=11)
    ag = MPCAgentPerfect(env.simulator, env.action_space, env.gamma, safety_margin=0.92, planning_steps=4, size_class=True)
    assert ag._fused(env) and ag.solver.backend.size_class == "s1"
    tot = torch.zeros(512, dtype=torch.float64, device=env.device)
    for _ in range(8):
        a = ag.act(env)
        assert bool(ag.last_converged.all()) and env.action_space.contains(a[0].cpu().numpy())
        _, r, _, _, _ = env.step(a.clone())
        tot += r
    assert not bool(env.terminated.any()) and float(tot.mean()) / 8 > -5.0


@pytest.mark.gpu
def test_closed_loop_on_a_network_that_steps_in_generic_mode_gpu():
    """a task on the meshed 20-bus network (host hooks): the environment steps on the general lane-group kernel, the
    reference's constant-forecast agent plans with the MPC kernel of the network's own small library"""
    from gym_anm_amd.envs.anm_env import BatchedANMEnv

    net = networks.synthetic_meshed_network(20, 3, 6)

    class Task(BatchedANMEnv):
        def __init__(self, num_envs):
            super().__init__(net, "state", 1, 0.25, 0.995, 100, aux_bounds=np.array([[0, 23]]), costs_clipping=(1, 100), seed=0,
                             num_envs=num_envs, device="cuda:0")
            m = self.simulator.model
            self._pl = torch.as_tensor(0.5 * m.dev_p_min[m.load_idx] * m.baseMVA, device=self.device)   # half load, MW
            self._pg = torch.as_tensor(0.6 * m.dev_p_max[m.gen_idx] * m.baseMVA, device=self.device)

        def init_state(self):
            m = self.simulator.model
            s0 = torch.zeros((self.num_envs, self.state_N), dtype=torch.float64, device=self.device)
            s0[:, list(m.load_idx)] = self._pl
            D, nd = m.N_device, m.N_des
            s0[:, 2 * D : 2 * D + nd] = torch.as_tensor(0.5 * (m.dev_soc_min[m.des_idx] + m.dev_soc_max[m.des_idx]) * m.baseMVA, device=self.device)
            s0[:, 2 * D + nd : 2 * D + nd + m.N_non_slack_gen] = self._pg
            return s0

        def next_vars(self, s_t):
            E = s_t.shape[0]
            wob = 1.0 + 0.2 * torch.sin(s_t[:, -1:] * 0.7)
            return torch.cat((self._pl.expand(E, -1) * wob, self._pg.expand(E, -1) * (2.0 - wob), (s_t[:, -1:] + 1) % 24), 1)

    env = Task(64)
    assert env.simulator.backend.generic and env.simulator.impl == "mesh"
    env.reset(seed=0)
    agent = MPCAgentConstant(env.simulator, env.action_space, env.gamma, safety_margin=0.92, planning_steps=4)
    assert os.path.basename(agent.solver.backend.path).startswith("libmpc_")
    lo, hi = torch.as_tensor(env.action_space.low, device=env.device), torch.as_tensor(env.action_space.high, device=env.device)
    tot = torch.zeros(64, dtype=torch.float64, device=env.device)
    for _ in range(8):
        a = agent.act(env)
        assert a.shape == (64, env.action_space.shape[0]) and bool(agent.last_converged.all())
        assert bool(((a >= lo - 1e-9) & (a <= hi + 1e-9)).all())
        _, r, term, _, _ = env.step(a)
        tot += r
    assert not bool(env.terminated.any()) and bool(torch.isfinite(tot).all())
    # planning beats doing nothing (zero set-points curtail every generator and leave the storage idle)
    env2 = Task(64)
    env2.reset(seed=0)
    tot0 = torch.zeros(64, dtype=torch.float64, device=env.device)
    for _ in range(8):
        _, r, _, _, _ = env2.step(torch.zeros_like(a))
        tot0 += r
    assert float(tot.mean()) >= float(tot0.mean()) - 1e-9


@pytest.mark.gpu
def test_network_beyond_the_register_budget_gpu():
    """30-bus feeder, 5 storage units: 123 rows and 42 variables per stage -- the same solver source with its row arrays
    in scratch (Sz::IN_REGISTERS false)"""
    net = networks.synthetic_radial_network(30, 0)
    check_random_programs(_gpu_sim(net), net, 6, 3, horizons=(1, 3, 10))


@pytest.mark.gpu
def test_discount_extremes_gpu():
    check_discount_extremes(_gpu_sim)


@pytest.mark.gpu
def test_angle_rows_gpu():
    check_angle_rows(_gpu_sim)


@pytest.mark.gpu
def test_other_stock_networks_gpu():
    check_other_stock_networks(_gpu_sim, 12)


@pytest.mark.gpu
def test_gpu_kernel_equals_the_host_build_of_the_same_solver():
    """same source, two compilers: objective and first-stage set-points of the gfx950 kernel vs the host double
    (not bit-equal: the host build has no fused multiply-adds; both stop at the same tolerance)"""
    net = networks.anm6_network()
    g = np.load(GOLDEN)
    k = 2
    args = (g["c%d_load" % k], g["c%d_gen" % k], g["c%d_soc" % k])
    out = []
    for sim in (_gpu_sim(net), _host_sim(net)):
        s = BatchedDCOPF(sim, float(g["gamma"]), float(g["safety_margin"][k]), int(g["N"][k]))
        s.solve(*args)
        out.append((s.objective.cpu().numpy(), s.iters.cpu().numpy()))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-8, atol=1e-8)
    assert np.abs(out[0][1] - out[1][1]).max() <= 4


@pytest.mark.gpu
def test_full_batch_is_consistent_with_small_batches():
    """65 536 environments (N = 1: one thread each; N = 10: 4 per wavefront): tiling the recorded programs over the
    batch gives every copy the recorded value -- lane / group bookkeeping at scale"""
    net = networks.anm6_network()
    g = np.load(GOLDEN)
    sim = _gpu_sim(net)
    for k in (0, 2):
        N = int(g["N"][k])
        s = BatchedDCOPF(sim, float(g["gamma"]), float(g["safety_margin"][k]), N)
        reps = 65536 // 60 + 1
        pl, pg, soc = (np.tile(g["c%d_%s" % (k, nm)], (reps,) + (1,) * (g["c%d_%s" % (k, nm)].ndim - 1))[:65536]
                       for nm in ("load", "gen", "soc"))
        s.solve(pl, pg, soc)
        ref = np.tile(g["c%d_objective" % k], reps)[:65536]
        obj = s.objective.cpu().numpy()
        assert np.abs(obj - ref).max() <= 1e-7 * (1 + np.abs(ref).max())
        assert int(s.iters.max()) < s.max_iter


@pytest.mark.gpu
def test_fused_act_equals_the_unfused_path_gpu():
    """(4 096 autoresetting environments on the fast path of the step: the kernel reads the state through the
    state_same flags and the compact time index, like `env.state` assembles it)"""
    env = ANM6EasyVec(num_envs=4096, device="cuda:0", seed=5, autoreset=True)
    env.reset(seed=5)
    assert env._state_same is not None and env._aux_index is not None
    check_fused_act(env, 4096)


@pytest.mark.gpu
def test_mpc_closed_loop_gpu():
    env = ANM6EasyVec(num_envs=1024, device="cuda:0", seed=3)
    env.reset(seed=3)
    check_closed_loop(env, MPCAgentPerfect, 4, 10, 1024)
    check_closed_loop(env, MPCAgentConstant, 1, 10, 1024)
