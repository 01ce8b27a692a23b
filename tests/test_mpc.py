"""Batched MPC DC-OPF policy (gym_anm/agents/mpc.py:163-372).

Golden vectors (tests/golden/mpc_anm6.npz, oracle/make_golden_mpc.py): the UNMODIFIED reference agents
(MPCAgentPerfect / MPCAgentConstant, N = 1, 3, 10, 20) in closed loop on the reference's ANM6Easy, their
linear program -- built by the reference's own code -- solved by HiGHS behind a stand-in for cvxpy's modelling
API (cvxpy is not in the image; the reference's own tests of this path pass under the stand-in).  Pinned: the
optimal VALUE of every recorded program (solver-independent) and the optimality, in the oracle's program, of
the first-stage minimiser the reference's run returned.  An LP minimiser need not be unique, so actions are
compared through the value.  Then: the product's LP assembly + batched ADMM against the oracle and against
the golden values, and the closed loop on ANM6Easy."""
import os

import numpy as np
import pytest
import torch

import anm_oracle as O
import mpc_oracle as MO
from gym_anm_amd import networks
from gym_anm_amd.agents import MPCAgentConstant, MPCAgentPerfect
from gym_anm_amd.agents.mpc import BatchedADMM, DCOPFProgram
from gym_anm_amd.envs import ANM6EasyVec
from gym_anm_amd.model import NetworkModel


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_anm6.npz")


def _golden_configs():
    g = np.load(GOLDEN)
    for k in range(len(g["N"])):
        yield k, str(g["kind"][k]), int(g["N"][k]), float(g["safety_margin"][k]), float(g["gamma"]), g


def test_oracle_program_has_the_value_of_the_reference_program():
    """oracle/mpc_oracle.py (the restatement the GPU path is checked against) vs the program the reference's own
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


def _admm_on_golden(device, k, kind, N, margin, gamma, g, max_iter, every=1):
    """the product's program + batched ADMM on the recorded cases of one configuration: value vs the reference's"""
    m = NetworkModel(networks.anm6_network(), 0.25, 100)
    pr = DCOPFProgram(m, gamma, margin, N)
    pl, pg, soc = g["c%d_load" % k][::every], g["c%d_gen" % k][::every], g["c%d_soc" % k][::every]
    E = len(soc)
    params = np.concatenate((pl.transpose(0, 2, 1).reshape(E, -1), pg.transpose(0, 2, 1).reshape(E, -1), soc), 1)
    l, u = pr.bounds(torch.as_tensor(params, device=device))
    x, info = BatchedADMM(pr.A, pr.q, pr.l0 == pr.u0, device).solve(l, u, max_iter=max_iter, eps=1e-6)
    obj = pr.objective(x).cpu().numpy()
    ref = g["c%d_objective" % k][::every]
    assert np.all(np.abs(obj - ref) <= 2e-4 * (1 + np.abs(ref))), (kind, N, np.abs(obj - ref).max(), info)
    Ax = x @ torch.as_tensor(pr.A.T, device=x.device)
    assert float((l - Ax).clamp(min=0).max()) < 2e-5 and float((Ax - u).clamp(min=0).max()) < 2e-5


def test_admm_reaches_the_reference_values_cpu():
    for k, kind, N, margin, gamma, g in _golden_configs():
        if N <= 3:
            _admm_on_golden("cpu", k, kind, N, margin, gamma, g, 20000, every=10)


@pytest.mark.gpu
def test_admm_reaches_the_reference_values_gpu():
    for k, kind, N, margin, gamma, g in _golden_configs():
        _admm_on_golden("cuda:0", k, kind, N, margin, gamma, g, 150000)  # (N = 10 takes ~60 000 iterations to 1e-6)


def _program_case(N, E, seed, device):
    net = networks.anm6_network()
    m = NetworkModel(net, 0.25, 100)
    n = O.parse_network(net, 0.25, 100)
    tab = O.anm6easy_tables() / 100.0
    rng = np.random.default_rng(seed)
    pr = DCOPFProgram(m, 0.995, 0.92, N)
    t0 = rng.integers(0, 96, E)
    soc = rng.uniform(0.05, 0.95, (E, 1))
    idx = (t0[:, None] + 1 + np.arange(N)[None, :]) % 96
    pl = np.stack([tab[:3][:, idx[e]] for e in range(E)])
    pg = np.stack([tab[3:][:, idx[e]] for e in range(E)])
    params = np.concatenate((pl.transpose(0, 2, 1).reshape(E, -1), pg.transpose(0, 2, 1).reshape(E, -1), soc), 1)
    l, u = pr.bounds(torch.as_tensor(params, device=device))
    return pr, n, pl, pg, soc, l, u


def _check_against_highs(device, N, E, n_check, max_iter):
    pr, n, pl, pg, soc, l, u = _program_case(N, E, 7, device)
    sol = BatchedADMM(pr.A, pr.q, pr.l0 == pr.u0, device)
    x, info = sol.solve(l, u, max_iter=max_iter, eps=1e-6)
    obj = pr.objective(x).cpu().numpy()
    Ax = x @ torch.as_tensor(pr.A.T, device=x.device)
    assert float((l - Ax).clamp(min=0).max()) < 2e-5 and float((Ax - u).clamp(min=0).max()) < 2e-5  # feasible (p.u.)
    for e in range(0, E, max(1, E // n_check)):
        ref = MO.solve_dcopf(n, pl[e], pg[e], soc[e], 0.995, 0.92, N)
        assert ref["status"] == 0
        assert abs(obj[e] - ref["objective"]) <= 2e-4 * (1 + abs(ref["objective"])), (e, obj[e], ref["objective"], info)
        # the oracle's optimum is feasible for the product's constraint rows too (same program)
        Ar = pr.A @ ref["x"]
        assert (Ar >= l[e].cpu().numpy() - 1e-7).all() and (Ar <= u[e].cpu().numpy() + 1e-7).all()
    return info


def test_dcopf_objective_matches_highs_cpu():   # (N = 3 on the CPU: test_admm_reaches_the_reference_values_cpu)
    _check_against_highs("cpu", 1, 12, 6, 20000)


def test_program_rows_follow_the_reference_constraints():
    """Row counts of mpc.py:232-313 for ANM6, N = 2: per stage 6 balances, 3 loads, 2 x 2 generator rows,
    5 storage rows, 6 + 1 angle rows, 3 x 5 branch rows."""
    pr = DCOPFProgram(NetworkModel(networks.anm6_network(), 0.25, 100), 0.995, 0.9, 2)
    assert pr.n == 2 * (6 + 7 + 1 + 1 + 5) and pr.m == 2 * (6 + 3 + 4 + 5 + 7 + 15)
    assert pr.n_param == 2 * 5 + 1
    # objective: slack + nothing else is a classical generator in ANM6; overload epigraphs weighted by lamb
    assert pr.q[6 + 0] == 1.0 and pr.q[6 + 2] == 0.0 and pr.q[6 + 7 + 2] == 100.0 and pr.q[20 + 6] == 0.995


def test_closed_loop_on_the_host_double():
    from hostsim_backend import hostsim_backend

    net = networks.anm6_network()
    be = hostsim_backend(NetworkModel(net, 0.25, 100).topology())
    env = ANM6EasyVec(num_envs=8, device="cpu", seed=3, _backend=be)
    env.reset(seed=3)
    ag = MPCAgentConstant(env.simulator, env.action_space, env.gamma, safety_margin=0.92, planning_steps=1, eps=1e-4,
                          max_iter=3000)
    tot = torch.zeros(8, dtype=torch.float64)
    for _ in range(12):
        a = ag.act(env)
        assert a.shape == (8, 6) and bool((a[:, 2:4] == 0).all()) and bool((a[:, 5] == 0).all())  # Q set-points are 0
        assert env.action_space.contains(a[0].numpy())
        _, r, term, _, _ = env.step(a)
        tot += r
    assert not bool(env.terminated.any()) and float(tot.mean()) / 12 > -5.0  # random agent: about -175 per step


@pytest.mark.gpu
def test_dcopf_objective_matches_highs_gpu():
    info = _check_against_highs("cuda:0", 3, 1024, 8, 20000)
    assert info["iters"] <= 20000


@pytest.mark.gpu
def test_mpc_perfect_closed_loop_gpu():
    env = ANM6EasyVec(num_envs=1024, device="cuda:0", seed=3)
    env.reset(seed=3)
    ag = MPCAgentPerfect(env.simulator, env.action_space, env.gamma, safety_margin=0.92, planning_steps=4, eps=1e-4,
                         max_iter=4000)
    tot = torch.zeros(1024, dtype=torch.float64, device="cuda:0")
    for _ in range(10):
        _, r, term, _, _ = env.step(ag.act(env))
        tot += r
    assert not bool(env.terminated.any()) and float(tot.mean()) / 10 > -5.0


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
    agent = MPCAgentConstant(env.simulator, env.action_space, env.gamma, safety_margin=0.96, planning_steps=1)
    total = 0.0
    for t in range(4):
        a = agent.act(env)
        assert isinstance(a, np.ndarray) and a.shape == (6,) and env.action_space.contains(a)
        o, r, terminated, _, _ = env.step(a)
        assert not terminated and isinstance(r, float)
        total += r
    assert total > -20.0  # the random agent loses ~175 per step on this task; the MPC policy a fraction of one
