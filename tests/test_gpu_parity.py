"""GPU tier (pytest -m gpu): the hipcc-built gfx950 libraries, called through the C ABI from the
Python boundary, against (a) the golden vectors recorded from the reference, (b) the CPU oracle on
the same seeded inputs, (c) size-independent properties at BASELINE.json's full batch sizes.

Bar: flags, Newton iteration counts and indices bit-exact; floating point within 1e-9 p.u.
(BASELINE.json's north_star asks for 1e-6 p.u.)."""
import os

import numpy as np
import numpy.testing as npt
import pytest
import torch

import parity_common as pc
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

NETS = pc.golden_nets()
DEV = "cuda:0"


def _loaded_native(sim):
    assert sim.backend.device_type == "cuda" and sim.backend.path.endswith(".so")
    assert "gym_anm_amd/_build/libanm_" in sim.backend.path


@pytest.mark.parametrize("name", sorted(NETS))
def test_transition_golden(name):
    sim = pc.check_transition_against_golden(name, NETS[name], DEV)
    _loaded_native(sim)


@pytest.mark.parametrize("name", ["anm6", "case30", "3bus"])
def test_transition_golden_f32_solve(name):
    pc.check_transition_against_golden(name, NETS[name], DEV, precision="f32", atol=2e-6, check_iters=False)


def test_anm6easy_episodes():
    from gym_anm_amd.envs import ANM6EasyVec

    env = pc.run_episodes(lambda n: ANM6EasyVec(num_envs=n, device=DEV))
    _loaded_native(env.simulator)


def test_single_env_dropin_matches_reference_episode():
    """ANM6Easy (num_envs=1, NumPy-facing) with reset(seed=s) reproduces the reference episode,
    including the NumPy generator consumption across resets."""
    from gym_anm_amd.envs import ANM6Easy

    g = np.load(os.path.join(GOLDEN, "anm6easy_episodes.npz"))
    for e in (0, 4):  # the two seeds whose episodes contain terminations + resets
        env = ANM6Easy(device=DEV)
        obs, info = env.reset(seed=int(g["seed"][e]))
        assert info == {} and obs.shape == (18,) and obs.dtype == np.float64
        npt.assert_allclose(obs, g["obs0"][e], rtol=0, atol=1e-8)
        reset_off = np.concatenate(([0], np.cumsum(g["reset_at_count"])))
        j = 0
        for t in range(200):
            o, r, term, trunc, info = env.step(g["actions"][e][t])
            assert isinstance(r, float) and isinstance(term, bool) and trunc is False
            assert term == bool(g["terminated"][e][t])
            npt.assert_allclose(o, g["obs"][e][t], rtol=0, atol=1e-7)
            npt.assert_allclose(r, g["reward"][e][t], rtol=1e-9, atol=1e-8)
            if term:
                o2, r2, t2, _, _ = env.step(g["actions"][e][t])  # absorbing state (anm_env.py:365-367)
                assert t2 and r2 == 0.0 and not o2.any()
                o, _ = env.reset()
                npt.assert_allclose(o, g["reset_obs"][reset_off[e] + j], rtol=0, atol=1e-8)
                j += 1
        assert j == int((g["reset_at"][reset_off[e] : reset_off[e + 1]] < 200).sum()) and j > 0


def _oracle_env():
    import anm_oracle as O
    from gym_anm_amd import networks

    return O.OracleEnv(networks.anm6_network(), sparse=False)


@pytest.mark.parametrize("num_envs", [4096, 65536])
def test_full_batch_vs_oracle_and_properties(num_envs):
    """BASELINE.json configs 2/3: full batch on the GPU; a seeded sample of environments is
    replayed by the oracle; the whole batch is checked through size-independent properties."""
    from gym_anm_amd.envs import ANM6EasyVec

    env = ANM6EasyVec(num_envs=num_envs, device=DEV, seed=3)
    env.check_actions = False
    obs, _ = env.reset(seed=3)
    state0 = env.state.clone()
    soc0 = env.simulator.soc.clone()
    gen = torch.Generator(device=DEV).manual_seed(5)
    lo = torch.as_tensor(env.action_space.low, device=DEV)
    hi = torch.as_tensor(env.action_space.high, device=DEV)
    T = 6
    acts, obs_l, r_l, term_l, it_l = [], [], [], [], []
    for t in range(T):
        a = lo + (hi - lo) * torch.rand((num_envs, 6), generator=gen, dtype=torch.float64, device=DEV)
        o, r, term, trunc, _ = env.step(a)
        acts.append(a.cpu().numpy())
        obs_l.append(o.cpu().numpy().copy())
        r_l.append(r.cpu().numpy().copy())
        term_l.append(term.cpu().numpy().copy())
        it_l.append(env.simulator.nr_iters.cpu().numpy().copy())
    # (1) oracle replay of a seeded sample
    rng = np.random.default_rng(0)
    sample = rng.choice(num_envs, size=48, replace=False)
    s0, c0 = state0.cpu().numpy(), soc0.cpu().numpy()
    for e in sample:
        orc = _oracle_env()
        orc.state, orc.soc, orc.terminated = s0[e].copy(), c0[e].copy(), False
        for t in range(T):
            o, r, term = orc.step(acts[t][e])
            assert term == bool(term_l[t][e])
            npt.assert_allclose(obs_l[t][e], o, rtol=0, atol=1e-7)
            npt.assert_allclose(r_l[t][e], r, rtol=1e-9, atol=1e-8)
            if not term:
                assert it_l[t][e] == orc.last["n_iter"]
    # (2) properties over the whole batch
    low = torch.as_tensor(env.observation_space.low, device=DEV)
    high = torch.as_tensor(env.observation_space.high, device=DEV)
    o = torch.as_tensor(obs_l[-1], device=DEV)
    assert bool(((o >= low) & (o <= high)).all())  # observation inside its Box
    term = torch.as_tensor(term_l[-1], device=DEV)
    assert not bool(o[term].any())  # terminated environments expose the zero state
    frac = float(term.double().mean())
    assert 0.0 < frac < 0.2  # random agent: a few percent collapse, not none, not all
    # terminated is absorbing, reward is -c2/(1-gamma) exactly once
    for t in range(1, T):
        assert bool((term_l[t] | ~term_l[t - 1]).all())
        newly = term_l[t] & ~term_l[t - 1]
        npt.assert_allclose(r_l[t][newly], -100 / (1 - 0.995))
        npt.assert_array_equal(r_l[t][term_l[t - 1]], 0.0)
    # (3) determinism: the same inputs give bit-identical outputs
    env2 = ANM6EasyVec(num_envs=num_envs, device=DEV, seed=3)
    env2.check_actions = False
    env2.reset(options={"init_state": state0})
    env2.simulator.soc.copy_(soc0)
    for t in range(T):
        o2, r2, term2, _, _ = env2.step(torch.as_tensor(acts[t], device=DEV))
    npt.assert_array_equal(o2.cpu().numpy(), obs_l[-1])
    npt.assert_array_equal(r2.cpu().numpy(), r_l[-1])


def test_power_flow_equations_hold_case30():
    """BASELINE.json config 4 (30-bus radial, 16384 envs): every converged solution satisfies
    S = V conj(Y V) to the Newton tolerance, branch ends are consistent, and bus injections equal
    the sum of device injections (the invariants of tests/simulator/test_simulator_transitions.py:189-265)."""
    from gym_anm_amd import networks
    from gym_anm_amd.simulator import BatchedSimulator

    E_ = 16384
    net = networks.synthetic_radial_network(30, 0)
    sim = BatchedSimulator(net, 0.25, 100, num_envs=E_, device=DEV)
    m = sim.model
    gen = torch.Generator(device=DEV).manual_seed(1)

    def U(lo, hi):
        lo = torch.as_tensor(lo, device=DEV)
        hi = torch.as_tensor(hi, device=DEV)
        return lo + (hi - lo) * torch.rand((E_, lo.numel()), generator=gen, dtype=torch.float64, device=DEV)

    b = m.baseMVA
    pl = U(m.dev_p_min[m.load_idx] * b, 0 * m.dev_p_min[m.load_idx])
    pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * b)
    ps = U(m.dev_p_min[m.setp_idx] * b, m.dev_p_max[m.setp_idx] * b)
    qs = U(m.dev_q_min[m.setp_idx] * b, m.dev_q_max[m.setp_idx] * b)
    sim.soc.copy_(U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx]))
    st, r, e, p, conv = sim.transition(pl, pp, ps, qs)
    assert float(conv.double().mean()) > 0.99
    V = torch.polar(st.tensor("bus_v_magn", "pu"), st.tensor("bus_v_ang", "rad"))
    Y = torch.as_tensor(m.Y_bus, device=DEV)
    I = V @ Y.T
    S = V * I.conj()
    P, Q = st.tensor("bus_p", "pu"), st.tensor("bus_q", "pu")
    err = torch.maximum((S.real - P).abs(), (S.imag - Q).abs())[conv]
    assert float(err.max()) < 2e-5  # Newton stop rule 1e-5 (+ slack bus definition, exact)
    # bus injections = sum of device injections
    dp, dq = st.tensor("dev_p", "pu"), st.tensor("dev_q", "pu")
    idx = torch.as_tensor(m.dev_bus, device=DEV, dtype=torch.long)
    bp = torch.zeros_like(P).index_add_(1, idx, dp)
    npt.assert_allclose(bp[conv].cpu().numpy(), P[conv].cpu().numpy(), atol=1e-12)
    # sign and magnitude of the directed apparent power
    s, pf, qf = st.tensor("branch_s", "pu"), st.tensor("branch_p", "pu"), st.tensor("branch_q", "pu")
    assert bool((torch.sign(s) == torch.sign(pf))[conv].all())
    assert bool((s.abs() + 1e-12 >= torch.sqrt(pf**2 + qf**2))[conv].all())
    # oracle replay of a few environments
    import anm_oracle as O

    n = O.parse_network(net, 0.25, 100)
    plc, ppc, psc, qsc = (x.cpu().numpy() for x in (pl, pp, ps, qs))
    full = sim.full.cpu().numpy()
    sl = pc.full_slices(sim)
    # SoC before the step is gone (updated in place): regenerate it deterministically
    gen2 = torch.Generator(device=DEV).manual_seed(1)
    for shape in (pl, pp, ps, qs):
        torch.rand(shape.shape, generator=gen2, dtype=torch.float64, device=DEV)
    lo = torch.as_tensor(m.dev_soc_min[m.des_idx], device=DEV)
    hi = torch.as_tensor(m.dev_soc_max[m.des_idx], device=DEV)
    soc0 = (lo + (hi - lo) * torch.rand((E_, lo.numel()), generator=gen2, dtype=torch.float64, device=DEV)).cpu().numpy()
    for e_ in range(0, E_, 1024):
        out = O.transition(n, plc[e_], ppc[e_], psc[e_], qsc[e_], soc0[e_], sparse=False)
        assert out["converged"] == bool(conv[e_])
        npt.assert_allclose(full[e_, sl["bus_v_magn"]], np.abs(out["V"]), atol=1e-9)
        npt.assert_allclose(full[e_, sl["branch_s"]], out["br_s"], atol=1e-9)
        assert int(sim.nr_iters[e_]) == out["n_iter"]


def test_autoreset_and_device_rng():
    """Series-mode autoreset: a terminated environment is re-initialised at the next step from the
    counter-based RNG; the drawn initial state equals the host restatement (gym_anm_amd/rng.py)."""
    from gym_anm_amd import rng
    from gym_anm_amd.envs import ANM6EasyVec
    import anm_oracle as O
    from gym_anm_amd import networks

    E_ = 8192
    env = ANM6EasyVec(num_envs=E_, device=DEV, seed=11, autoreset=True)
    env.check_actions = False
    env.reset(seed=11)
    gen = torch.Generator(device=DEV).manual_seed(2)
    lo = torch.as_tensor(env.action_space.low, device=DEV)
    hi = torch.as_tensor(env.action_space.high, device=DEV)
    prev_term = env.terminated.clone()
    n_resets = 0
    for t in range(12):
        a = lo + (hi - lo) * torch.rand((E_, 6), generator=gen, dtype=torch.float64, device=DEV)
        counts_before = env._reset_count.clone()
        obs, r, term, _, _ = env.step(a)
        was = prev_term.cpu().numpy()
        if was.any():
            idx = np.where(was)[0]
            n_resets += len(idx)
            assert not bool(term[idx].any())  # ANM6Easy initial states always converge
            assert bool((r[idx] == 0).all())
            assert bool((env.timestep[idx] == 0).all())
            for e in idx[:8]:
                s0 = rng.series_init_state(env.simulator.model, env._series, 11, int(e), int(counts_before[e]))
                orc = O.OracleEnv(networks.anm6_network(), sparse=False)
                o_ref, conv = orc.reset_to(s0)
                assert conv
                npt.assert_allclose(obs[e].cpu().numpy(), o_ref, rtol=0, atol=1e-8)
                assert int(env.state[e, -1]) == int(s0[-1])  # same drawn time index (integer RNG path)
        prev_term = term.clone()
    assert n_resets > 0
    assert int(env._reset_count.sum()) == n_resets


def test_list_observation_gather():
    """List-form observation space (anm_env.py:497-521): gathered from the kernel's full dump."""
    from gym_anm_amd import networks
    from gym_anm_amd.envs.anm6 import ANM6Vec, anm6easy_series

    obs_spec = [("bus_v_magn", "all", "pu"), ("branch_s", [(1, 2), (2, 4)], "MVA"), ("des_soc", "all"),
                ("bus_v_ang", [3], "degree"), ("aux", [0])]  # fmt: skip
    env = ANM6Vec(obs_spec, 1, 0.25, 0.995, 100, aux_bounds=np.array([[0, 95]]), costs_clipping=(1, 100),
                  num_envs=64, device=DEV, series=anm6easy_series())  # fmt: skip
    g = np.load(os.path.join(GOLDEN, "anm6easy_episodes.npz"))
    s0 = np.tile(g["init_draws"][0], (64, 1))
    obs, _ = env.reset(options={"init_state": s0})
    assert obs.shape == (64, 6 + 2 + 1 + 1 + 1)
    st = env.simulator
    full = st.full.cpu().numpy()
    sl = pc.full_slices(st)
    npt.assert_allclose(obs[:, :6].cpu().numpy(), full[:, sl["bus_v_magn"]])
    npt.assert_allclose(obs[:, 6].cpu().numpy(), full[:, sl["branch_s"]][:, 1] * 100)
    npt.assert_allclose(obs[:, 8].cpu().numpy(), full[:, sl["des_soc"]][:, 0] * 100)
    npt.assert_allclose(obs[:, 9].cpu().numpy(), full[:, sl["bus_v_ang"]][:, 3] * 180 / np.pi)
    npt.assert_allclose(obs[:, 10].cpu().numpy(), s0[:, -1])
    a = torch.as_tensor(np.tile(g["actions"][0][0], (64, 1)), device=DEV)
    obs, r, term, _, _ = env.step(a)
    npt.assert_allclose(r.cpu().numpy(), g["reward"][0][0], rtol=1e-9)
    npt.assert_allclose(obs[:, 10].cpu().numpy(), g["state"][0][0][-1])


def test_generic_mode_next_vars_hook():
    """A task that supplies next_vars() on the host side (no fused series) steps identically."""
    from gym_anm_amd.envs import ANM6EasyVec
    from gym_anm_amd.envs.anm6 import ANM6Vec

    class HookEnv(ANM6Vec):
        def __init__(self, **kw):
            super().__init__("state", 1, 0.25, 0.995, 100, aux_bounds=np.array([[0, 95]]), costs_clipping=(1, 100), **kw)

        next_vars = ANM6EasyVec.next_vars

    g = np.load(os.path.join(GOLDEN, "anm6easy_episodes.npz"))
    n_ep = len(g["seed"])
    env = HookEnv(num_envs=n_ep, device=DEV)
    draw_off = np.concatenate(([0], np.cumsum(g["init_draws_count"])))
    s0 = np.stack([g["init_draws"][draw_off[e]] for e in range(n_ep)])
    obs, _ = env.reset(options={"init_state": s0})
    npt.assert_allclose(obs.cpu().numpy(), g["obs0"], atol=1e-8)
    first_term = np.argmax(g["terminated"], axis=1)
    horizon = int(min(np.where(g["terminated"].any(axis=1), first_term, 400).min(), 60))
    for t in range(horizon):
        obs, r, term, _, _ = env.step(torch.as_tensor(g["actions"][:, t], device=DEV))
        npt.assert_allclose(obs.cpu().numpy(), g["obs"][:, t], atol=1e-7)
        npt.assert_allclose(r.cpu().numpy(), g["reward"][:, t], rtol=1e-9, atol=1e-8)


# ---------------------------------------------------------------------------------------------
# lane-group ("radial") kernel family: same entry points, one wavefront lane per bus / device
# ---------------------------------------------------------------------------------------------
RADIAL_NETS = ["anm6", "2bus", "case30"]


@pytest.mark.parametrize("name", RADIAL_NETS)
def test_radial_transition_golden(name):
    sim = pc.check_transition_against_golden(name, NETS[name], DEV, impl="radial")
    assert sim.impl == "radial"


@pytest.mark.parametrize("name", ["anm6", "case30"])
def test_radial_transition_golden_f32_solve(name):
    pc.check_transition_against_golden(name, NETS[name], DEV, precision="f32", atol=2e-6, check_iters=False, impl="radial")


def test_radial_is_default_for_large_feeders_and_refused_for_meshed():
    from gym_anm_amd import errors
    from gym_anm_amd.simulator import BatchedSimulator

    assert BatchedSimulator(NETS["case30"], 0.25, 100, num_envs=4, device=DEV).impl == "radial"
    assert BatchedSimulator(NETS["anm6"], 0.25, 100, num_envs=4, device=DEV).impl == "thread"
    with pytest.raises(errors.HipExtensionError, match="radial"):
        BatchedSimulator(NETS["3bus"], 0.5, 100, num_envs=4, device=DEV, impl="radial")


def test_radial_anm6easy_episodes():
    from gym_anm_amd.envs import ANM6EasyVec

    env = pc.run_episodes(lambda n: ANM6EasyVec(num_envs=n, device=DEV, impl="radial"))
    assert env.simulator.impl == "radial"


def test_radial_equals_thread_kernels_with_autoreset():
    """The three kernel families on the same 8192 environments for 10 steps, in-kernel autoreset on:
    same flags, same RNG draws, values within summation-order round-off."""
    from gym_anm_amd.envs import ANM6EasyVec

    E_ = 8192
    envs = [ANM6EasyVec(num_envs=E_, device=DEV, seed=5, autoreset=True, impl=i) for i in ("thread", "radial", "mesh")]
    for env in envs:
        env.check_actions = False
        env.reset(seed=5)
    gen = torch.Generator(device=DEV).manual_seed(4)
    lo = torch.as_tensor(envs[0].action_space.low, device=DEV)
    hi = torch.as_tensor(envs[0].action_space.high, device=DEV)
    for t in range(10):
        a = lo + (hi - lo) * torch.rand((E_, 6), generator=gen, dtype=torch.float64, device=DEV)
        outs = [env.step(a) for env in envs]
        (o0, r0, t0, _, _) = outs[0]
        for other, (o1, r1, t1, _, _) in zip(envs[1:], outs[1:]):
            assert bool((t0 == t1).all())
            npt.assert_allclose(o1.cpu().numpy(), o0.cpu().numpy(), rtol=0, atol=1e-9)
            npt.assert_allclose(r1.cpu().numpy(), r0.cpu().numpy(), rtol=1e-10, atol=1e-9)
            ok = ~t0
            npt.assert_array_equal(envs[0].simulator.nr_iters[ok].cpu().numpy(), other.simulator.nr_iters[ok].cpu().numpy())
            npt.assert_array_equal(envs[0]._reset_count.cpu().numpy(), other._reset_count.cpu().numpy())
            npt.assert_array_equal(envs[0].timestep.cpu().numpy(), other.timestep.cpu().numpy())
    assert int(envs[0]._reset_count.sum()) > 0


def test_radial_reset_case30_matches_thread():
    """Simulator-level reset + env-level reset of the 30-bus feeder through both families."""
    from gym_anm_amd.envs.anm_env import BatchedANMEnv

    class Env30(BatchedANMEnv):
        def __init__(self, **kw):
            super().__init__(NETS["case30"], "state", 0, 0.25, 0.995, 100, costs_clipping=(1, 100), **kw)

        def next_vars(self, s_t):
            return self._exo

    rng = np.random.default_rng(3)
    E_ = 512
    envs = [Env30(num_envs=E_, device=DEV, impl=i) for i in ("thread", "radial")]
    m = envs[0].simulator.model
    b = m.baseMVA
    s0 = np.zeros((E_, envs[0].state_N))
    D = m.N_device
    for k in range(D):
        lo = m.dev_p_min[k] * b if np.isfinite(m.dev_p_min[k]) else -10
        hi_ = m.dev_p_max[k] * b if np.isfinite(m.dev_p_max[k]) else 10
        s0[:, k] = rng.uniform(lo, hi_, E_) * 0.5
        s0[:, D + k] = rng.uniform(-1, 1, E_)
    s0[:, 2 * D : 2 * D + m.N_des] = rng.uniform(0, 50, (E_, m.N_des))
    s0[:, 2 * D + m.N_des :] = rng.uniform(0, 20, (E_, m.N_non_slack_gen))
    obs = [env.reset(options={"init_state": s0})[0].cpu().numpy() for env in envs]
    npt.assert_allclose(obs[1], obs[0], rtol=0, atol=1e-8)
    npt.assert_array_equal(envs[0].pfe_converged.cpu().numpy(), envs[1].pfe_converged.cpu().numpy())
    exo = torch.as_tensor(rng.uniform(-5, 5, (E_, m.N_load + m.N_non_slack_gen)), device=DEV)
    lo_a, hi_a = m.action_bounds()
    for t in range(3):
        a = torch.as_tensor(rng.uniform(lo_a, hi_a, (E_, len(lo_a))), device=DEV)
        outs = []
        for env in envs:
            env._exo = exo
            env.check_actions = False
            outs.append(env.step(a))
        npt.assert_allclose(outs[1][0].cpu().numpy(), outs[0][0].cpu().numpy(), rtol=0, atol=1e-8)
        npt.assert_allclose(outs[1][1].cpu().numpy(), outs[0][1].cpu().numpy(), rtol=1e-9, atol=1e-8)
        assert bool((outs[0][2] == outs[1][2]).all())


# ---------------------------------------------------------------------------------------------
# randomised topologies: oracle on the same seeded inputs
# ---------------------------------------------------------------------------------------------
def _random_inputs(sim, seed):
    m, b, E_ = sim.model, sim.model.baseMVA, sim.num_envs
    rng = np.random.default_rng(seed)

    def U(lo, hi, wild=1.0):
        lo, hi = np.asarray(lo, float), np.asarray(hi, float)
        c, h = 0.5 * (lo + hi), 0.5 * (hi - lo) * wild
        return rng.uniform(c - h, c + h, size=(E_, len(lo)))

    pl = U(m.dev_p_min[m.load_idx] * b * 0.6, 0 * m.dev_p_min[m.load_idx])
    pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * b, 1.2)
    ps = U(m.dev_p_min[m.setp_idx] * b, m.dev_p_max[m.setp_idx] * b, 1.3)
    qs = U(m.dev_q_min[m.setp_idx] * b, m.dev_q_max[m.setp_idx] * b, 1.3)
    soc = U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx])
    return pl, pp, ps, qs, soc


def _compare_with_oracle(sim, net, inputs, n_check):
    import anm_oracle as O

    pl, pp, ps, qs, soc = inputs
    sim.soc.copy_(torch.as_tensor(soc))
    st, r, e, p, conv = sim.transition(pl, pp, ps, qs)
    full = sim.full.cpu().numpy()
    sl = pc.full_slices(sim)
    n = O.parse_network(net, sim.delta_t, sim.lamb)
    n_conv = 0
    for k in range(n_check):
        out = O.transition(n, pl[k], pp[k], ps[k], qs[k], soc[k], sparse=False)
        assert out["converged"] == bool(conv[k]), k
        npt.assert_allclose(full[k, sl["dev_p"]][1:], out["dev_p"][1:], rtol=0, atol=1e-12)
        npt.assert_allclose(sim.soc[k].cpu().numpy(), out["soc_after"], rtol=0, atol=1e-12)
        if not out["converged"]:
            continue
        n_conv += 1
        assert int(sim.nr_iters[k]) == out["n_iter"], k
        npt.assert_allclose(full[k, sl["bus_v_magn"]], np.abs(out["V"]), rtol=0, atol=1e-9)
        npt.assert_allclose(full[k, sl["bus_v_ang"]], np.angle(out["V"]), rtol=0, atol=1e-9)
        npt.assert_allclose(full[k, sl["branch_s"]], out["br_s"], rtol=0, atol=1e-9)
        npt.assert_allclose(full[k, sl["dev_p"]], out["dev_p"], rtol=0, atol=1e-9)
        npt.assert_allclose(float(r[k]), out["reward"], rtol=1e-9, atol=1e-8)
    return n_conv


@pytest.mark.parametrize("n_bus,seed", [(3, 1), (5, 2), (9, 3), (17, 4), (33, 5), (48, 6), (64, 7)])
def test_random_radial_networks_generic_lane_group_kernel(n_bus, seed):
    """Arbitrary radial networks need no per-topology build: the table-driven lane-group kernel of
    any already-built library serves them (G = 8 .. 64 lanes per environment)."""
    from gym_anm_amd import networks
    from gym_anm_amd.simulator import BatchedSimulator

    net = networks.synthetic_radial_network(n_bus, seed)
    E_ = 200
    sim = BatchedSimulator(net, 0.25, 100, num_envs=E_, device=DEV, impl="radial")
    assert sim.impl == "radial"
    npt.assert_allclose(sim.device_ybus(), sim.model.Y_bus, rtol=1e-15, atol=0)
    n_conv = _compare_with_oracle(sim, net, _random_inputs(sim, seed), 40)
    assert n_conv >= 20


@pytest.mark.parametrize("shape", ["star", "path", "feeders"])
def test_structured_trees_generic_lane_group_kernel(shape):
    """Trees the random feeders do not produce -- a hub with 20 leaves (one parent folds 20 children), a path of 24
    buses (as many levels as buses), five feeders off the slack (five roots) -- through the table-driven radial kernel."""
    from gym_anm_amd.simulator import BatchedSimulator

    net = pc.structured_network(shape)
    sim = BatchedSimulator(net, 0.25, 100, num_envs=128, device=DEV, impl="radial")
    assert sim.impl == "radial"
    npt.assert_allclose(sim.device_ybus(), sim.model.Y_bus, rtol=1e-15, atol=0)
    assert _compare_with_oracle(sim, net, _random_inputs(sim, 3), 40) >= 10


@pytest.mark.parametrize("n_bus,seed", [(4, 11), (7, 12)])
def test_new_topology_is_compiled_on_first_use(n_bus, seed):
    """A topology without a prebuilt library is specialised with hipcc on first use
    (thread-per-environment kernels) and agrees with the oracle and with the lane-group kernel."""
    from gym_anm_amd import codegen, networks
    from gym_anm_amd.simulator import BatchedSimulator

    if codegen.hipcc_path() is None:
        pytest.skip("hipcc not available on this box")
    net = networks.synthetic_radial_network(n_bus, seed)
    E_ = 128
    sim = BatchedSimulator(net, 0.25, 100, num_envs=E_, device=DEV, impl="thread")
    assert sim.impl == "thread" and not sim.backend.generic
    inputs = _random_inputs(sim, seed)
    _compare_with_oracle(sim, net, inputs, 30)
    sim2 = BatchedSimulator(net, 0.25, 100, num_envs=E_, device=DEV, impl="radial")
    sim2.soc.copy_(torch.as_tensor(inputs[4]))
    sim2.transition(*inputs[:4])
    ok = (sim.pfe_converged & sim2.pfe_converged).cpu().numpy()
    assert bool((sim.pfe_converged == sim2.pfe_converged).all())
    sl = pc.full_slices(sim)
    f1, f2 = sim.full.cpu().numpy()[ok], sim2.full.cpu().numpy()[ok]
    for key in sl:
        if key.endswith("_i_ang"):  # the angle of a ~zero current is ill-conditioned
            continue
        npt.assert_allclose(f2[:, sl[key]], f1[:, sl[key]], rtol=0, atol=1e-9, err_msg=key)


@pytest.mark.parametrize("impl", ["thread", "radial", "mesh"])
def test_device_sampler_reset(impl):
    """reset(options={"sampler": "device"}) on every kernel family: the in-kernel draws equal the host
    restatement of the counter-based RNG and the oracle's reset of that state."""
    import anm_oracle as O
    from gym_anm_amd import networks, rng
    from gym_anm_amd.envs import ANM6EasyVec

    E_ = 4096
    env = ANM6EasyVec(num_envs=E_, device=DEV, seed=21, env_offset=7 * E_, impl=impl)
    obs, _ = env.reset(seed=21, options={"sampler": "device"})
    assert not bool(env.terminated.any()) and bool((env._reset_count == 1).all())
    for e in range(0, E_, 401):
        s0 = rng.series_init_state(env.simulator.model, env._series, 21, 7 * E_ + e, 0)
        o_ref, conv = O.OracleEnv(networks.anm6_network(), sparse=False).reset_to(s0)
        assert conv
        npt.assert_allclose(obs[e].cpu().numpy(), o_ref, rtol=0, atol=1e-9)
    # stepping after a device-sampled reset works and stays inside the Box
    env.check_actions = False
    a = torch.zeros((E_, 6), dtype=torch.float64, device=DEV)
    o, r, term, _, _ = env.step(a)
    low = torch.as_tensor(env.observation_space.low, device=DEV)
    high = torch.as_tensor(env.observation_space.high, device=DEV)
    assert bool(((o >= low) & (o <= high)).all())


@pytest.mark.parametrize("small_workspace", [False, True])
@pytest.mark.parametrize("cap_after", [1, 4, 8])
def test_two_phase_step_is_bit_identical(cap_after, small_workspace):
    """The two-launch step (first launch stops after `cap_after` Newton iterations, the solves still
    running are continued by the straggler launch) returns bit-identical results to the one-launch
    step, including in-kernel autoreset, on 65 536 environments -- also when the record workspace is too
    small for all the stragglers (the ones without a slot are finished by the first launch itself)."""
    from gym_anm_amd.envs import ANM6EasyVec

    E_ = 65536
    envs = [ANM6EasyVec(num_envs=E_, device=DEV, seed=9, autoreset=True, tol=1e-6, straggler_after=sa, handoff_after=None)
            for sa in (None, cap_after)]  # fmt: skip  (no lane-group continuation: both stay thread-per-environment)
    assert envs[0]._ws is None and envs[1]._ws is not None
    if small_workspace:
        rec = envs[1].simulator.backend.lib.anm_step_ws_record_doubles()
        envs[1]._ws.n_doubles = 8 + 100 * rec  # room for 100 records; ~400 environments diverge per step
    for env in envs:
        env.check_actions = False
        env.reset(seed=9)
    gen = torch.Generator(device=DEV).manual_seed(6)
    lo = torch.as_tensor(envs[0].action_space.low, device=DEV)
    hi = torch.as_tensor(envs[0].action_space.high, device=DEV)
    n_term = 0
    for t in range(8):
        a = lo + (hi - lo) * torch.rand((E_, 6), generator=gen, dtype=torch.float64, device=DEV)
        (o0, r0, t0, _, _), (o1, r1, t1, _, _) = [env.step(a) for env in envs]
        n_term += int(t0.sum())
        for x0, x1 in ((o0, o1), (r0, r1), (t0, t1), (envs[0].state, envs[1].state), (envs[0].e_loss, envs[1].e_loss),
                       (envs[0].penalty, envs[1].penalty), (envs[0].simulator.soc, envs[1].simulator.soc),
                       (envs[0].simulator.nr_iters, envs[1].simulator.nr_iters), (envs[0].timestep, envs[1].timestep),
                       (envs[0]._reset_count, envs[1]._reset_count), (envs[0]._aux_index, envs[1]._aux_index)):  # fmt: skip
            assert torch.equal(x0, x1), t
    assert n_term > 100  # the workload did produce diverging solves


@pytest.mark.parametrize("small_workspace,precision", [(False, "f64"), (True, "f64"), (False, "f32")])
def test_two_phase_step_with_lane_group_stragglers_equals_the_in_wave_hand_over(small_workspace, precision):
    """With the lane-group continuation on (the default), the straggler launch of the two-launch step runs its
    records on lane groups, 8 per wavefront.  Handed over after the same number of iterations, these are the
    very solves the one-launch step continues on lane groups of their own wavefront -- same code, same inputs:
    bit-identical results, also when the workspace overflows (the rest then continues in-wave)."""
    from gym_anm_amd.envs import ANM6EasyVec

    E_ = 65536
    envs = [ANM6EasyVec(num_envs=E_, device=DEV, seed=11, autoreset=True, tol=1e-6, straggler_after=sa, handoff_after=6,
                        precision=precision) for sa in (None, 6)]  # fmt: skip
    assert envs[0]._ws is None and envs[1]._ws is not None
    if small_workspace:
        rec = envs[1].simulator.backend.lib.anm_step_ws_record_doubles()
        envs[1]._ws.n_doubles = 8 + 100 * rec
    for env in envs:
        env.check_actions = False
        env.reset(seed=11)
    gen = torch.Generator(device=DEV).manual_seed(8)
    lo = torch.as_tensor(envs[0].action_space.low, device=DEV)
    hi = torch.as_tensor(envs[0].action_space.high, device=DEV)
    n_term = 0
    for t in range(8):
        a = lo + (hi - lo) * torch.rand((E_, 6), generator=gen, dtype=torch.float64, device=DEV)
        (o0, r0, t0, _, _), (o1, r1, t1, _, _) = [env.step(a) for env in envs]
        n_term += int(t0.sum())
        for x0, x1 in ((o0, o1), (r0, r1), (t0, t1), (envs[0].state, envs[1].state), (envs[0].e_loss, envs[1].e_loss),
                       (envs[0].penalty, envs[1].penalty), (envs[0].simulator.soc, envs[1].simulator.soc),
                       (envs[0].simulator.nr_iters, envs[1].simulator.nr_iters), (envs[0].timestep, envs[1].timestep),
                       (envs[0]._reset_count, envs[1]._reset_count), (envs[0]._aux_index, envs[1]._aux_index)):  # fmt: skip
            assert torch.equal(x0, x1), t
    assert n_term > 100


@pytest.mark.parametrize("small_workspace,mid", [(False, 12), (True, 12), (False, 7), (False, 40)])
def test_two_level_straggler_continuation_is_bit_identical(small_workspace, mid):
    """Round 3: the first straggler launch stops at `straggler_mid` iterations and leaves the records still running
    to a second, smaller launch (anm_step_ws.mid_cap).  Same lane-group code, same inputs as the one-level two-launch
    step and as the in-wave hand-over: bit-identical, also with an overflowing workspace."""
    from gym_anm_amd.envs import ANM6EasyVec

    E_ = 65536
    envs = [ANM6EasyVec(num_envs=E_, device=DEV, seed=13, autoreset=True, tol=1e-6, straggler_after=sa, straggler_mid=sm, handoff_after=6)
            for sa, sm in ((None, None), (6, None), (6, mid))]  # fmt: skip
    assert envs[0]._ws is None and envs[1]._ws.mid_cap == -1 and envs[2]._ws.mid_cap == mid
    if small_workspace:
        rec = envs[2].simulator.backend.lib.anm_step_ws_record_doubles()
        envs[2]._ws.n_doubles = 8 + 100 * rec + 50
    for env in envs:
        env.check_actions = False
        env.reset(seed=13)
    gen = torch.Generator(device=DEV).manual_seed(10)
    lo = torch.as_tensor(envs[0].action_space.low, device=DEV)
    hi = torch.as_tensor(envs[0].action_space.high, device=DEV)
    n_term = 0
    for t in range(8):
        a = lo + (hi - lo) * torch.rand((E_, 6), generator=gen, dtype=torch.float64, device=DEV)
        outs = [env.step(a) for env in envs]
        n_term += int(outs[0][2].sum())
        for k in (1, 2):
            for x0, x1 in ((outs[0][0], outs[k][0]), (outs[0][1], outs[k][1]), (outs[0][2], outs[k][2]), (envs[0].state, envs[k].state),
                           (envs[0].e_loss, envs[k].e_loss), (envs[0].penalty, envs[k].penalty), (envs[0].simulator.soc, envs[k].simulator.soc),
                           (envs[0].simulator.nr_iters, envs[k].simulator.nr_iters), (envs[0].timestep, envs[k].timestep),
                           (envs[0]._reset_count, envs[k]._reset_count), (envs[0]._aux_index, envs[k]._aux_index)):  # fmt: skip
                assert torch.equal(x0, x1), (t, k)
    assert n_term > 100


def test_long_run_invariants():
    """3 000 steps of 65 536 autoresetting environments under a random agent: observations stay finite
    and inside the Box, rewards inside the clipping range, every environment keeps cycling through
    collapse -> reset, and a sample of environments replayed by the oracle from a late snapshot still
    agrees (no slow drift of the persistent state)."""
    from gym_anm_amd import networks
    from gym_anm_amd.envs import ANM6EasyVec

    import anm_oracle as O

    E_ = 65536
    env = ANM6EasyVec(num_envs=E_, device=DEV, seed=3, autoreset=True, tol=1e-6)
    env.check_actions = False
    env.reset(seed=3)
    gen = torch.Generator(device=DEV).manual_seed(11)
    lo = torch.as_tensor(env.action_space.low, device=DEV)
    hi = torch.as_tensor(env.action_space.high, device=DEV)
    olo = torch.as_tensor(env.observation_space.low, device=DEV)
    ohi = torch.as_tensor(env.observation_space.high, device=DEV)
    r_min = -env.costs_clipping[1] / (1 - env.gamma)
    n_term = torch.zeros((), dtype=torch.int64, device=DEV)
    for t in range(3000):
        a = lo + (hi - lo) * torch.rand((E_, 6), generator=gen, dtype=torch.float64, device=DEV)
        obs, rew, term, _, _ = env.step(a)
        n_term += term.sum()
        if t % 500 == 499:
            assert bool(torch.isfinite(obs).all()) and bool(((obs >= olo) & (obs <= ohi)).all())
            assert bool(torch.isfinite(rew).all()) and bool((rew <= 1.0 + 1e-12).all()) and bool((rew >= r_min - 1e-9).all())
    frac = float(n_term) / (3000 * E_)
    assert 0.003 < frac < 0.01, frac  # ~0.6 % of the random-agent steps collapse
    assert int(env._reset_count.min()) >= 1 and int(env.timestep.max()) < 3000
    # replay 16 live environments for 5 more steps with the oracle, starting from the device state
    idx = torch.nonzero(~env.terminated)[:16, 0]
    oracles = []
    for i in idx.tolist():
        o = O.OracleEnv(networks.anm6_network(), sparse=False, tol=1e-6)
        o.load_state(env.state[i].cpu().numpy(), float(env.simulator.soc[i, 0]))
        o.done = False
        oracles.append(o)
    for _ in range(5):
        a = lo + (hi - lo) * torch.rand((E_, 6), generator=gen, dtype=torch.float64, device=DEV)
        obs, rew, term, _, _ = env.step(a)
        for o, i in zip(oracles, idx.tolist()):
            if o.done:
                continue
            oo, rr, tt = o.step(a[i].cpu().numpy())
            assert bool(term[i]) == tt
            if tt:
                o.done = True
                continue
            np.testing.assert_allclose(obs[i].cpu().numpy(), oo, rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(float(rew[i]), rr, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("impl", ["thread", "radial", "mesh"])
def test_results_do_not_depend_on_batch_size_or_neighbours(impl):
    """Environment e gets bit-identical results whatever the batch it is stepped in (1, 63, 65, 130 or
    256 environments: ragged last wavefront, different wave neighbours, diverging neighbours or not)."""
    from gym_anm_amd.envs import ANM6EasyVec

    sizes = (256, 1, 63, 65, 130)
    gen = torch.Generator(device=DEV).manual_seed(4)
    envs = []
    for n in sizes:
        env = ANM6EasyVec(num_envs=n, device=DEV, seed=8, autoreset=True, tol=1e-6, impl=impl)
        env.check_actions = False
        env.reset(seed=8, options={"sampler": "device"})  # keyed by the environment index, not by the batch
        envs.append(env)
    lo = torch.as_tensor(envs[0].action_space.low, device=DEV)
    hi = torch.as_tensor(envs[0].action_space.high, device=DEV)
    n_term = 0
    for t in range(40):
        a = lo + (hi - lo) * torch.rand((256, 6), generator=gen, dtype=torch.float64, device=DEV)
        ref = None
        for n, env in zip(sizes, envs):
            obs, rew, term, _, _ = env.step(a[:n].contiguous())
            cur = (obs, rew, term, env.state, env.simulator.soc, env.simulator.nr_iters, env.timestep)
            if ref is None:
                ref = [x.clone() for x in cur]
                n_term += int(term.sum())
            else:
                for x, y in zip(cur, ref):
                    assert torch.equal(x, y[:n]), (n, t)
    assert n_term > 10  # collapses and in-kernel resets happened along the way
