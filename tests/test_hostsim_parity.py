"""CPU tier: the kernel templates compiled for the host (test double, tests/hostsim) + the real
Python host layer, against the golden vectors recorded from the reference.  This is what keeps
the kernel logic, the constant packing and the host code honest when no GPU is present; the GPU
tier (test_gpu_parity.py) repeats the same checks through the hipcc-built library."""
import numpy as np
import pytest
import torch

from gym_anm_amd.envs import ANM6EasyVec
from gym_anm_amd.model import NetworkModel

import parity_common as pc
from hostsim_backend import hostsim_backend

NETS = pc.golden_nets()


def _backend(net):
    return hostsim_backend(NetworkModel(net, 0.25, 100).topology())


@pytest.mark.parametrize("name", sorted(NETS))
def test_transition_golden(name):
    pc.check_transition_against_golden(name, NETS[name], "cpu", backend=_backend(NETS[name]))


@pytest.mark.parametrize("name", ["anm6", "case30", "3bus"])
def test_transition_golden_f32_solve(name):
    """fp32 Jacobian/LU with fp64 mismatch and update: same flags, answers within 1e-6 p.u.
    (iteration counts may grow by one, so they are not compared)."""
    pc.check_transition_against_golden(name, NETS[name], "cpu", backend=_backend(NETS[name]), precision="f32",
                                       atol=2e-6, check_iters=False)  # fmt: skip


def test_anm6easy_episodes():
    net = NETS["anm6"]
    pc.run_episodes(lambda n: ANM6EasyVec(num_envs=n, device="cpu", _backend=_backend(net)))


def test_device_sampler_reset():
    """reset(options={"sampler": "device"}): initial states drawn inside the reset kernel equal the host
    restatement of the counter-based RNG (gym_anm_amd/rng.py) + the oracle's reset; masked variant."""
    import numpy.testing as npt

    import anm_oracle as O
    from gym_anm_amd import networks, rng

    net = NETS["anm6"]
    E_ = 96
    env = ANM6EasyVec(num_envs=E_, device="cpu", seed=21, env_offset=1000, _backend=_backend(net))
    obs, _ = env.reset(seed=21, options={"sampler": "device"})
    assert not bool(env.terminated.any()) and bool((env.timestep == 0).all())
    assert bool((env._reset_count == 1).all())
    for e in range(0, E_, 7):
        s0 = rng.series_init_state(env.simulator.model, env._series, 21, 1000 + e, 0)
        orc = O.OracleEnv(networks.anm6_network(), sparse=False)
        o_ref, conv = orc.reset_to(s0)
        assert conv
        npt.assert_allclose(obs[e].numpy(), o_ref, rtol=0, atol=1e-9)
    # masked: only the selected environments are redrawn (with their next epoch)
    before = env.state.clone()
    mask = torch.zeros(E_, dtype=torch.bool)
    mask[::3] = True
    obs2, _ = env.reset(options={"sampler": "device", "mask": mask})
    npt.assert_array_equal(env.state[~mask].numpy(), before[~mask].numpy())
    assert bool((env._reset_count[mask] == 2).all()) and bool((env._reset_count[~mask] == 1).all())
    s0 = rng.series_init_state(env.simulator.model, env._series, 21, 1000 + 3, 1)
    o_ref, _ = O.OracleEnv(networks.anm6_network(), sparse=False).reset_to(s0)
    npt.assert_allclose(obs2[3].numpy(), o_ref, rtol=0, atol=1e-9)


def test_numpy_vector_env_adapter():
    """NumpyVectorEnv: NumPy in / NumPy out around a batched environment with Gymnasium's next-step
    autoreset convention; the step after a collapse returns the first observation of a new episode with
    reward 0 and terminated False; an action outside the Box raises like the reference."""
    import anm_oracle as O
    from gym_anm_amd import networks
    from gym_anm_amd.envs import NumpyVectorEnv

    net = NETS["anm6"]
    E_ = 32
    venv = NumpyVectorEnv(ANM6EasyVec(num_envs=E_, device="cpu", seed=5, autoreset=True, _backend=_backend(net)))
    assert venv.num_envs == E_ and venv.single_action_space.shape == (6,)
    obs, info = venv.reset(seed=5)
    assert isinstance(obs, np.ndarray) and obs.shape == (E_, 18) and obs.dtype == np.float64
    orc = O.OracleEnv(networks.anm6_network(), sparse=False)
    orc.load_state(venv.env.state[0].numpy(), float(venv.env.simulator.soc[0, 0]))
    rng_ = np.random.default_rng(0)
    lo, hi = venv.single_action_space.low, venv.single_action_space.high
    was_term = np.zeros(E_, dtype=bool)
    seen_reset = False
    for t in range(60):
        a = rng_.uniform(lo, hi, size=(E_, 6))
        if t == 3:
            a[:, 2] = 10 * hi[2]  # outside the action Box
            with pytest.raises(AssertionError):
                venv.step(a)
            a[:, 2] = hi[2]
        obs, rew, term, trunc, info = venv.step(a)
        assert all(isinstance(x, np.ndarray) for x in (obs, rew, term, trunc)) and not trunc.any()
        # environments that were terminated at the previous call have been re-initialised by this one
        if was_term.any():
            seen_reset = True
            assert (rew[was_term] == 0).all() and not term[was_term].any()
            assert (venv.env.timestep.numpy()[was_term] == 0).all()
        if not orc.terminated:
            o_ref, r_ref, t_ref = orc.step(a[0])
            assert bool(term[0]) == t_ref
            if not t_ref:
                np.testing.assert_allclose(obs[0], o_ref, rtol=0, atol=1e-9)
                np.testing.assert_allclose(rew[0], r_ref, rtol=1e-9)
        was_term = term.copy()
    assert seen_reset


def _meshed_network(n_bus, seed, n_chords):
    from gym_anm_amd import networks

    return networks.synthetic_meshed_network(n_bus, seed, n_chords)


@pytest.mark.parametrize("n_bus,seed,n_chords", [(4, 1, 1), (5, 2, 2), (7, 3, 2), (9, 4, 3)])
def test_random_meshed_networks_against_oracle(n_bus, seed, n_chords):
    """Topology compiler + kernel templates on networks nobody hand-checked: random meshed networks with
    loops, line charging and a phase-shifting transformer, compared case by case with the oracle."""
    import anm_oracle as O
    from gym_anm_amd.simulator import BatchedSimulator

    net = _meshed_network(n_bus, seed, n_chords)
    model = NetworkModel(net, 0.25, 100)
    M = 48
    sim = BatchedSimulator(net, 0.25, 100, num_envs=M, device="cpu", tol=1e-8, _backend=hostsim_backend(model.topology()))
    rng = np.random.default_rng(seed)
    b = model.baseMVA

    def U(lo, hi, scale=1.0):
        lo, hi = np.asarray(lo, float) * scale, np.asarray(hi, float) * scale
        return lo + (hi - lo) * rng.uniform(size=(M, lo.size))

    pl = U(model.dev_p_min[model.load_idx], 0 * model.dev_p_min[model.load_idx], b)
    pp = U(0 * model.dev_p_max[model.gen_idx], model.dev_p_max[model.gen_idx], b)
    ps = U(model.dev_p_min[model.setp_idx], model.dev_p_max[model.setp_idx], 1.2 * b)
    qs = U(model.dev_q_min[model.setp_idx], model.dev_q_max[model.setp_idx], 1.2 * b)
    soc = U(model.dev_soc_min[model.des_idx], model.dev_soc_max[model.des_idx])
    pl[-4:] *= 40.0  # a few hopeless cases: both sides must give up the same way
    sim.soc.copy_(torch.as_tensor(soc))
    sim.transition(pl, pp, ps, qs)
    full = sim.full.numpy()
    off = sim.full_offsets
    n = O.parse_network(net, 0.25, 100)
    n_conv = 0
    for e in range(M):
        ref = O.transition(n, pl[e], pp[e], ps[e], qs[e], soc[e], tol=1e-8, sparse=False)
        assert bool(sim.pfe_converged[e]) == bool(ref["converged"]), e
        if not ref["converged"]:
            continue
        n_conv += 1
        assert int(sim.nr_iters[e]) == ref["n_iter"], e
        V = ref["V"]
        np.testing.assert_allclose(full[e, off["bus_v_magn"]: off["bus_v_magn"] + n_bus], np.abs(V), rtol=0, atol=1e-9)
        np.testing.assert_allclose(full[e, off["bus_v_ang"]: off["bus_v_ang"] + n_bus], np.angle(V), rtol=0, atol=1e-9)
        np.testing.assert_allclose(full[e, off["dev_p"]: off["dev_p"] + model.N_device], ref["dev_p"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(full[e, off["branch_s"]: off["branch_s"] + model.N_branch], ref["br_s"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(float(sim.reward[e]), ref["reward"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(sim.soc[e].numpy(), ref["soc_after"], rtol=0, atol=1e-12)
    assert n_conv >= M // 2


# ---------------------------------------------------------------------------------------------------
# round 2: the bodies of tests/test_gpu_headline.py on the host test double (host layer + kernel logic)
# ---------------------------------------------------------------------------------------------------
_KW = lambda net: {"device": "cpu", "_backend": _backend(net)}


def test_headline_config_oracle_replay_small():
    n, n_term, n_reset = pc.headline_replay(_KW, 4096, 10, 24, 24)
    assert n >= 32 and n_term >= 8 and n_reset >= 8


@pytest.mark.parametrize("name", ["anm6", "3bus"])
def test_reset_golden_simulator_and_env(name):
    pc.reset_golden(name, _KW)


def test_sharded_batches_equal_the_unsharded_batch():
    n_term, max_resets = pc.sharded_equal_unsharded(_KW, 2048, 30)
    assert n_term > 20 and max_resets >= 1


def test_callable_observation_and_two_aux_variables():
    pc.callable_observation_and_two_aux(_KW, E_=24, T=12)


def test_all_state_variables_observation():
    pc.all_state_variables_observation(_KW, n_cases=300)


def test_reference_custom_obs_space_known_answers():
    pc.reference_custom_obs_space(_KW)


def _reference_available():
    try:
        import ref_harness

        return ref_harness.reference_available()
    except Exception:
        return False


@pytest.mark.skipif(not _reference_available(), reason="reference checkout not present (GPU box)")
def test_observation_spaces_of_random_lists_equal_the_reference():
    """Differential test of the observation-spec compiler's host side (anm_env.py:193-233, 497-521): 60 random
    list-form observations (random quantities, 'all' or random ids, every unit) on ANM6 -- same expanded
    obs_values and bit-identical Box bounds as the reference's ANMEnv.  Dev container only."""
    import ref_harness

    ref_harness.load_reference()
    from gym_anm.envs import ANMEnv
    import numpy.testing as npt
    from gym_anm_amd import networks
    from gym_anm_amd.envs.anm_env import BatchedANMEnv
    from gym_anm_amd.model import STATE_VARIABLES

    net = networks.anm6_network()
    ids = {"bus": [0, 1, 2, 3, 4, 5], "dev": [0, 1, 2, 3, 4, 5, 6], "des": [6], "gen": [2, 4],
           "branch": [(0, 1), (1, 2), (1, 3), (2, 4), (2, 5)]}
    rng = np.random.default_rng(3)
    keys = [k for k in STATE_VARIABLES if k != "aux"]
    for case in range(60):
        spec = []
        for key in rng.choice(keys, size=rng.integers(1, 6), replace=False):
            units = STATE_VARIABLES[key]
            unit = units[rng.integers(len(units))] if isinstance(units, tuple) else units
            pool = ids[key.split("_")[0]]
            if rng.uniform() < 0.5:
                nodes = "all"
            else:
                pick = sorted(rng.choice(len(pool), size=rng.integers(1, len(pool) + 1), replace=False))
                nodes = [pool[i] for i in pick]
            spec.append((str(key), nodes, str(unit)))
        ref = ANMEnv(net, [tuple(s) for s in spec], 0, 0.25, 0.995, 100, None, (None, None), None)
        env = BatchedANMEnv(net, [tuple(s) for s in spec], 0, 0.25, 0.995, 100, num_envs=2, **_KW(net))
        assert env.obs_values == ref.obs_values, (case, spec)
        npt.assert_array_equal(env.observation_space.low, ref.observation_space.low, err_msg=str(spec))
        npt.assert_array_equal(env.observation_space.high, ref.observation_space.high, err_msg=str(spec))


@pytest.mark.skipif(not _reference_available(), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("seed", [1, 3, 4])
def test_list_observation_episode_equals_the_live_reference(seed):
    """ANM6Easy's dynamics with a LIST observation (the committed golden episodes use "state"): the reference's
    environment and the batched one (one environment, host double), same seed and actions, 150 steps with resets --
    observations, rewards and terminations step by step.  Dev container only."""
    import numpy.testing as npt
    import ref_harness

    ref_harness.load_reference()
    from gym_anm.envs import ANM6Easy
    from gym_anm.envs.anm6_env.anm6 import ANM6
    from gym_anm_amd import networks
    from gym_anm_amd.envs.anm6 import ANM6Vec, anm6easy_series, random_date

    observation = [("bus_v_magn", "all", "pu"), ("branch_s", [(1, 2), (2, 4)], "MVA"), ("des_soc", "all", "MWh"),
                   ("dev_q", [2, 4, 6], "MVAr"), ("bus_v_ang", [3, 5], "degree"), ("gen_p_max", "all", "MW"), ("aux", "all")]

    class RefTask(ANM6Easy):
        def __init__(self):
            ANM6.__init__(self, observation, 1, 0.25, 0.995, 100, np.array([[0, 95]]), (1, 100))
            from gym_anm.envs.anm6_env.anm6_easy import _get_gen_time_series, _get_load_time_series

            self.P_loads, self.P_maxs = _get_load_time_series(), _get_gen_time_series()

    class Task(ANM6EasyVec):
        def __init__(self, **kw):
            self.P_loads, self.P_maxs = anm6easy_series()[:3], anm6easy_series()[3:]
            ANM6Vec.__init__(self, observation, 1, 0.25, 0.995, 100, aux_bounds=np.array([[0, 95]]), costs_clipping=(1, 100),
                             series=anm6easy_series(), **kw)

    ref = RefTask()
    env = Task(num_envs=1, seed=seed, **_KW(networks.anm6_network()))
    o_ref, _ = ref.reset(seed=seed)
    o, _ = env.reset(seed=seed)
    random_date(env.np_random, 2020)  # ANM6.reset draws the rendering date from np_random (anm6.py:138): the batched
    npt.assert_allclose(o[0].cpu().numpy(), o_ref, rtol=0, atol=1e-9)  # class leaves that to its NumPy-facing wrapper
    rng = np.random.default_rng(50 + seed)
    n_term = 0
    for t in range(150):
        a = rng.uniform(ref.action_space.low, ref.action_space.high)
        if t % 3:
            a[2:] *= 0.3
        o_ref, r_ref, term_ref, _, _ = ref.step(a)
        o, r, term, _, _ = env.step(torch.as_tensor(a[None, :]))
        assert bool(term[0]) == bool(term_ref), t
        npt.assert_allclose(o[0].cpu().numpy(), o_ref, rtol=0, atol=1e-7, err_msg="step %d" % t)
        npt.assert_allclose(float(r[0]), r_ref, rtol=1e-9, atol=1e-8)
        if term_ref:
            n_term += 1
            o_ref, _ = ref.reset()
            o, _ = env.reset()
            random_date(env.np_random, 2020)
            npt.assert_allclose(o[0].cpu().numpy(), o_ref, rtol=0, atol=1e-9)


def _simple_env_classes(kw):
    """The reference's examples/simple_env.py (2-bus grid, random loads, one useless aux variable), written against
    `ANMEnv` twice: the reference's class and this package's NumPy-facing single-environment class."""
    from gym_anm_amd import ANMEnv as Ours

    network = {
        "baseMVA": 100,
        "bus": np.array([[0, 0, 132, 1.0, 1.0], [1, 1, 33, 1.1, 0.9]]),
        "device": np.array([[0, 0, 0, None, 200, -200, 200, -200, None, None, None, None, None, None, None],
                            [1, 1, -1, 0.2, 0, -10, None, None, None, None, None, None, None, None, None]]),
        "branch": np.array([[0, 1, 0.01, 0.1, 0.0, 3, 1, 0]]),
    }

    def make(base, extra):
        class SimpleEnvironment(base):
            def __init__(self):
                super().__init__(network, "state", 1, 0.25, 0.9, 100, np.array([[0, 10]]), (1, 100), 1, **extra)

            def init_state(self):
                n_dev, n_des, n_gen = self.simulator.N_device, self.simulator.N_des, self.simulator.N_non_slack_gen
                return np.random.rand(2 * n_dev + n_des + n_gen + self.K)

            def next_vars(self, s_t):
                P_load = -10 * np.random.rand(1)[0]
                aux = np.random.randint(0, 10)
                return np.array([P_load, aux])

        return SimpleEnvironment

    return make, Ours, kw(network)


def test_reference_example_simple_env_runs_unmodified():
    """`class SimpleEnvironment(ANMEnv)` of examples/simple_env.py, body unchanged: 1-D states, 1-D next_vars, 1-D
    NumPy observations, float rewards, Python bools."""
    make, Ours, extra = _simple_env_classes(_KW)
    env = make(Ours, extra)()
    np.random.seed(5)
    o, info = env.reset()
    assert isinstance(o, np.ndarray) and o.shape == (5,) and info == {}
    for t in range(10):
        a = env.action_space.sample()
        o, r, terminated, truncated, info = env.step(a)
        assert isinstance(o, np.ndarray) and o.shape == (5,) and isinstance(r, float) and isinstance(terminated, bool)
        assert truncated is False and env.observation_space.contains(o)
    assert env.state.shape == (5,) and env.timestep == 10


@pytest.mark.skipif(not _reference_available(), reason="reference checkout not present (GPU box)")
def test_reference_example_simple_env_equals_the_live_reference():
    """... and it walks the same trajectory as the reference's class under the same global NumPy seed."""
    import numpy.testing as npt
    import ref_harness

    ref_harness.load_reference()
    from gym_anm import ANMEnv as Ref

    make, Ours, extra = _simple_env_classes(_KW)
    traj = []
    for base, ex in ((Ref, {}), (Ours, extra)):
        env = make(base, ex)()
        np.random.seed(11)
        o, _ = env.reset()
        rows = [np.concatenate((o, [0.0, 0.0]))]
        for t in range(40):
            o, r, term, _, _ = env.step(env.action_space.sample())
            rows.append(np.concatenate((o, [r, float(term)])))
        traj.append(np.array(rows))
    npt.assert_allclose(traj[1], traj[0], rtol=1e-9, atol=1e-9)


@pytest.mark.skipif(not _reference_available(), reason="reference checkout not present (GPU box)")
def test_reference_example_custom_anm6_equals_the_live_reference():
    """`class CustomANM6Environment(ANM6)` of examples/custom_anm6.py (random loads and generation potentials within
    their limits, time-of-day aux variable), body unchanged, on `gym_anm.envs.ANM6` and on this package's `ANM6`:
    same trajectory under the same global NumPy seed (random agent from a seeded generator)."""
    import numpy.testing as npt
    import ref_harness

    ref_harness.load_reference()
    from gym_anm.envs import ANM6 as Ref
    from gym_anm_amd import networks
    from gym_anm_amd.envs import ANM6 as Ours

    def make(base, extra):
        class CustomANM6Environment(base):
            def __init__(self):
                super().__init__("state", 1, 0.25, 0.9, 100, np.array([[0, 10]]), (1, 100), 1, **extra)

            def init_state(self):
                n_dev, n_des, n_gen = self.simulator.N_device, self.simulator.N_des, self.simulator.N_non_slack_gen
                s = np.random.rand(2 * n_dev + n_des + n_gen)
                return np.hstack((s, 0))

            def next_vars(self, s_t):
                next_var = [-10 * np.random.rand(1)[0], 30 * np.random.rand(1)[0], -30 * np.random.rand(1)[0],
                            50 * np.random.rand(1)[0], -30 * np.random.rand(1)[0]]
                next_var.append(int((s_t[-1] + 1) % (24 / self.delta_t)))
                return np.array(next_var)

        return CustomANM6Environment

    traj = []
    for base, ex in ((Ref, {}), (Ours, _KW(networks.anm6_network()))):
        env = make(base, ex)()
        np.random.seed(3)
        o, _ = env.reset()
        rng = np.random.default_rng(9)
        rows = [np.concatenate((o, [0.0, 0.0]))]
        for t in range(60):
            a = rng.uniform(env.action_space.low, env.action_space.high)
            a[2:] *= 0.2
            o, r, term, _, _ = env.step(a)
            rows.append(np.concatenate((o, [r, float(term)])))
            if term:
                o, _ = env.reset()
                rows.append(np.concatenate((o, [0.0, 0.0])))
        traj.append(np.array(rows))
    assert traj[0].shape == traj[1].shape
    npt.assert_allclose(traj[1], traj[0], rtol=1e-9, atol=1e-7)


@pytest.mark.skipif(not _reference_available(), reason="reference checkout not present (GPU box)")
def test_new_env_template_with_callable_observation_and_own_bounds_equals_the_live_reference():
    """examples/new_env_template.py filled in with the optional pieces: a CALLABLE observation (anm_env.py:511-514)
    and an overridden observation_bounds() (anm_env.py:193-233) on the 3-bus loop -- the reference's ANMEnv and this
    package's, same global NumPy seed."""
    import numpy.testing as npt
    import ref_harness

    ref_harness.load_reference()
    from gym_anm import ANMEnv as Ref
    from gym_anm_amd import ANMEnv as Ours
    from gym_anm_amd import networks
    from gym_anm_amd.spaces import Box

    net = networks.three_bus_loop_network(base_mva=10, gen_max=100.0)

    def make(base, extra, box):
        class CustomEnvironment(base):
            def __init__(self):
                obs = lambda s: np.array([s[1] + s[2], s[-1] ** 2, np.tanh(s[0])])  # noqa: E731
                super().__init__(net, obs, 2, 0.5, 0.95, 100, np.array([[0, 50], [-1, 1]]), (10, 200), 4, **extra)

            def init_state(self):
                s = np.random.rand(self.state_N)
                s[-2:] = [3.0, 0.5]
                return s

            def next_vars(self, s_t):
                return np.array([-3 * np.random.rand(), 40 * np.random.rand(), 60 * np.random.rand(), (s_t[-2] + 1) % 50, -s_t[-1]])

            def observation_bounds(self):
                return box(low=np.array([-50.0, 0.0, -1.0]), high=np.array([150.0, 1.0, 1.0]))

        return CustomEnvironment

    import gymnasium  # the harness's stand-in in the dev container

    traj = []
    for base, ex, box in ((Ref, {}, gymnasium.spaces.Box), (Ours, _KW(net), Box)):
        env = make(base, ex, box)()
        np.random.seed(21)
        o, _ = env.reset()
        npt.assert_array_equal(env.observation_space.high, [150.0, 1.0, 1.0])
        rng = np.random.default_rng(2)
        rows = [np.concatenate((o, [0.0, 0.0]))]
        for t in range(40):
            a = rng.uniform(env.action_space.low, env.action_space.high) * 0.2
            o, r, term, _, _ = env.step(a)
            rows.append(np.concatenate((o, [r, float(term)])))
            if term:
                break
        traj.append(np.array(rows))
    assert traj[0].shape == traj[1].shape and len(traj[0]) > 5
    npt.assert_allclose(traj[1], traj[0], rtol=1e-9, atol=1e-8)


@pytest.mark.skipif(not _reference_available(), reason="reference checkout not present (GPU box)")
def test_the_references_anm6easy_methods_run_on_this_packages_anm6_base():
    """The reference's OWN task code -- `ANM6Easy.init_state` and `ANM6Easy.next_vars`, the unmodified function objects
    imported from /root/reference -- grafted onto this package's `ANM6` base class: they read `self.np_random`,
    `self.simulator.devices[i].qp_ratio / q_min / q_max / soc_min / soc_max`, `self.K`, `self.delta_t`, and the
    environment walks the trajectory of the reference's ANM6Easy."""
    import numpy.testing as npt
    import ref_harness

    ref_harness.load_reference()
    from gym_anm.envs import ANM6Easy as Ref
    from gym_anm.envs.anm6_env import anm6_easy as RE
    from gym_anm_amd import networks
    from gym_anm_amd.envs import ANM6 as OursANM6

    extra = _KW(networks.anm6_network())

    def init(self):
        OursANM6.__init__(self, "state", 1, 0.25, 0.995, 100, np.array([[0, 95]]), (1, 100), **extra)
        self.P_loads, self.P_maxs = RE._get_load_time_series(), RE._get_gen_time_series()

    Grafted = type("ANM6EasyOnThisPackage", (OursANM6,), {"__init__": init, "init_state": RE.ANM6Easy.init_state,
                                                          "next_vars": RE.ANM6Easy.next_vars})
    traj = []
    for cls in (Ref, Grafted):
        env = cls()
        o, _ = env.reset(seed=6)
        rng = np.random.default_rng(8)
        rows = [np.concatenate((o, [0.0, 0.0]))]
        for t in range(80):
            a = rng.uniform(env.action_space.low, env.action_space.high)
            if t % 4:
                a[2:] *= 0.25
            o, r, term, _, _ = env.step(a)
            rows.append(np.concatenate((o, [r, float(term)])))
            if term:
                o, _ = env.reset()
                rows.append(np.concatenate((o, [0.0, 0.0])))
        traj.append(np.array(rows))
    assert traj[0].shape == traj[1].shape
    npt.assert_allclose(traj[1], traj[0], rtol=1e-9, atol=1e-7)


def test_component_views_of_the_simulator():
    """`simulator.devices[i]`, `.buses[i]`, `.branches[(i, j)]` (simulator.py:148-179): constants under the reference's
    attribute names, dynamic quantities as tensors over the batch; read-only."""
    from gym_anm_amd import networks
    from gym_anm_amd.simulator import BatchedSimulator

    net = networks.anm6_network()
    sim = BatchedSimulator(net, 0.25, 100, num_envs=4, **_KW(net))
    assert list(sim.devices) == [0, 1, 2, 3, 4, 5, 6] and list(sim.buses) == [0, 1, 2, 3, 4, 5]
    assert list(sim.branches) == [(0, 1), (1, 2), (1, 3), (2, 4), (2, 5)]
    d6, d1, d0 = sim.devices[6], sim.devices[1], sim.devices[0]
    assert (d6.soc_min, d6.soc_max, d6.eff, d6.p_min, d6.p_max) == (0.0, 1.0, 0.9, -0.5, 0.5)
    assert d1.qp_ratio == 0.2 and d1.type == -1 and d0.is_slack and not d6.is_slack and d6.bus_id == 5
    assert sim.branches[(0, 1)].rate == 0.32 and sim.buses[1].v_max == 1.1 and sim.buses[0].is_slack
    sim.soc[:] = 0.5
    sim.transition({1: -5.0, 3: -20.0, 5: -25.0}, {2: 30.0, 4: 40.0}, {2: 30.0, 4: 50.0, 6: 10.0}, {2: 5.0, 4: -20.0, 6: 30.0})
    # SURVEY 8(c)'s hand-checked case
    assert abs(float(d6.soc[0]) - 0.4722222222222222) < 1e-12 and abs(float(sim.devices[2].p[0]) - 0.3) < 1e-12
    assert abs(float(sim.buses[1].v.abs()[0]) - 1.006185017186) < 1e-9 and abs(float(sim.devices[4].p_pot[0]) - 0.4) < 1e-12
    assert abs(float(sim.branches[(0, 1)].s_apparent_max[0]) + 0.295959430807) < 1e-9
    with pytest.raises(AttributeError):
        d6.soc = 0.3
    with pytest.raises(AttributeError):
        d1.soc_min


def test_single_environment_simulator_state_follows_the_steps():
    """In the reference `env.simulator.state` / `.devices[i].p` are current after every step (simulator.py:529-537);
    the NumPy-facing classes track the electrical state by default."""
    from hostsim_backend import hostsim_backend

    from gym_anm_amd import networks
    from gym_anm_amd.envs import ANM6Easy

    net = networks.anm6_network()
    env = ANM6Easy(device="cpu", _backend=hostsim_backend(NetworkModel(net, 0.25, 100).topology()))
    env.reset(seed=1)
    for t in range(3):
        a = 0.5 * (env.action_space.low + env.action_space.high)
        o, r, term, _, _ = env.step(a)
        assert not term
        st = env.simulator.state
        for k, dev in enumerate(range(7)):
            assert abs(float(st["dev_p"]["MW"][dev][0]) - o[k]) < 1e-9          # the observation IS the state here
            assert abs(float(env.simulator.devices[dev].p[0]) * 100.0 - o[k]) < 1e-9
        assert abs(float(env.simulator.devices[6].soc[0]) * 100.0 - o[14]) < 1e-9


@pytest.mark.skipif(not _reference_available(), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("kind,n_bus,seed,n_chords", [("mesh", 5, 51, 2), ("radial", 8, 52, 0), ("mesh", 9, 53, 3)])
def test_random_task_on_a_random_network_equals_the_live_reference(kind, n_bus, seed, n_chords):
    """A task nobody wrote down: random network (several loads, generators, storage units), "state" observation,
    K = 1, exogenous variables drawn from NumPy's global stream within the devices' limits, random agent --
    the reference's ANMEnv and this package's, 60 steps with resets, same trajectory (state layout, device
    ordering, SoC bookkeeping of several storage units, clipping, terminal handling)."""
    import numpy.testing as npt
    import ref_harness

    ref_harness.load_reference()
    from gym_anm import ANMEnv as Ref
    from gym_anm_amd import ANMEnv as Ours
    from gym_anm_amd import networks

    net = networks.synthetic_meshed_network(n_bus, seed, n_chords) if kind == "mesh" else networks.synthetic_radial_network(n_bus, seed)
    m = NetworkModel(net, 0.25, 100)
    b = m.baseMVA
    load_lo = m.dev_p_min[m.load_idx] * b
    gen_hi = m.dev_p_max[m.gen_idx] * b

    def make(base, extra):
        class Task(base):
            def __init__(self):
                if seed == 52:  # the optional arguments left at their defaults (anm_env.py:79-90)
                    super().__init__(net, "state", 1, 0.25, 0.99, 100, **extra)
                else:
                    super().__init__(net, "state", 1, 0.25, 0.99, 100, np.array([[0, 23]]), (1, 100), 0, **extra)

            def init_state(self):
                s = np.zeros(self.state_N)
                D = self.simulator.N_device
                s[:D] = np.random.uniform(-1, 1, D)
                s[D : 2 * D] = np.random.uniform(-0.5, 0.5, D)
                s[2 * D : 2 * D + self.simulator.N_des] = np.random.uniform(5, 20, self.simulator.N_des)
                s[2 * D + self.simulator.N_des : -1] = np.random.uniform(0, 5, self.simulator.N_non_slack_gen)
                return s

            def next_vars(self, s_t):
                return np.concatenate((np.random.uniform(0.5 * load_lo, 0.0), np.random.uniform(0.0, 0.6 * gen_hi), [(s_t[-1] + 1) % 24]))

        return Task

    traj, spaces_ = [], []
    for base, ex in ((Ref, {}), (Ours, _KW(net))):
        env = make(base, ex)()
        spaces_.append((env.state_N, list(env.state_values), np.asarray(env.observation_space.low), np.asarray(env.observation_space.high),
                        np.asarray(env.action_space.low), np.asarray(env.action_space.high)))
        np.random.seed(100 + seed)
        o, _ = env.reset()
        rng = np.random.default_rng(seed)
        rows = [np.concatenate((o, [0.0, 0.0]))]
        for t in range(60):
            a = rng.uniform(env.action_space.low, env.action_space.high) * 0.3
            o, r, term, _, _ = env.step(a)
            rows.append(np.concatenate((o, [r, float(term)])))
            if term:
                o, _ = env.reset()
                rows.append(np.concatenate((o, [0.0, 0.0])))
        traj.append(np.array(rows))
    assert traj[0].shape == traj[1].shape
    npt.assert_allclose(traj[1], traj[0], rtol=1e-9, atol=1e-7)
    assert spaces_[0][0] == spaces_[1][0] and spaces_[0][1] == spaces_[1][1]          # state_N, state_values
    for a, b in zip(spaces_[0][2:], spaces_[1][2:]):                                      # observation and action boxes
        npt.assert_array_equal(a, b)


@pytest.mark.skipif(not _reference_available(), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("kind,n_bus,seed,n_chords", [("mesh", 6, 61, 2), ("radial", 10, 62, 0)])
def test_the_whole_state_dictionary_equals_the_live_reference(kind, n_bus, seed, n_chords):
    """`Simulator.transition(...)` with the reference's dict arguments on a random network with mixed base voltages,
    then EVERY entry of `simulator.state[quantity][unit][id]` (simulator.py:551-636: all quantities, all units,
    kV / kA conversions per bus) against the reference's dictionary."""
    import numpy.testing as npt
    import ref_harness

    ref_harness.load_reference()
    from gym_anm.simulator import Simulator
    from gym_anm_amd import networks
    from gym_anm_amd.simulator import BatchedSimulator

    net = networks.synthetic_meshed_network(n_bus, seed, n_chords) if kind == "mesh" else networks.synthetic_radial_network(n_bus, seed)
    net = dict(net, bus=np.array(net["bus"], dtype=float).copy())
    net["bus"][1:, 2] = [33.0 if i % 2 else 11.0 for i in range(1, n_bus)]       # mixed base voltages
    ref = Simulator(net, 0.25, 100)
    sim = BatchedSimulator(net, 0.25, 100, num_envs=2, **_KW(net))
    m = sim.model
    rng = np.random.default_rng(seed)
    b = m.baseMVA
    loads = [m.dev_ids[k] for k in m.load_idx]
    gens = [m.dev_ids[k] for k in m.gen_idx]
    setp = [m.dev_ids[k] for k in m.setp_idx]
    n_checked = 0
    for trial in range(6):
        P_load = {i: float(rng.uniform(0.6 * m.dev_p_min[m.dev_ids.index(i)] * b, 0.0)) for i in loads}
        P_pot = {i: float(rng.uniform(0.0, m.dev_p_max[m.dev_ids.index(i)] * b)) for i in gens}
        P_set = {i: float(rng.uniform(0.3 * m.dev_p_min[m.dev_ids.index(i)] * b, 0.3 * m.dev_p_max[m.dev_ids.index(i)] * b)) for i in setp}
        Q_set = {i: float(rng.uniform(0.3 * m.dev_q_min[m.dev_ids.index(i)] * b, 0.3 * m.dev_q_max[m.dev_ids.index(i)] * b)) for i in setp}
        soc = rng.uniform(0.2, 0.8, m.N_des) * np.array([m.dev_soc_max[k] for k in m.des_idx])
        for j, k in enumerate(m.des_idx):
            ref.devices[m.dev_ids[k]].soc = float(soc[j])
        sim.soc[:] = torch.as_tensor(soc)
        _, r_ref, e_ref, p_ref, conv_ref = ref.transition(P_load, P_pot, P_set, Q_set)
        st, r, e, p, conv = sim.transition(P_load, P_pot, P_set, Q_set)
        assert bool(conv[0]) == bool(conv_ref)
        if not conv_ref:
            continue
        npt.assert_allclose([float(r[0]), float(e[0]), float(p[0])], [r_ref, e_ref, p_ref], rtol=1e-9, atol=1e-9)
        assert set(st.keys()) == set(ref.state.keys())
        for key in ref.state:
            ours = st[key]
            assert set(ours.keys()) == set(ref.state[key].keys()), key
            for unit, per_id in ref.state[key].items():
                assert list(ours[unit].keys()) == list(per_id.keys()), (key, unit)
                for i, v in per_id.items():
                    atol = 1e-9
                    if key.endswith("_ang"):  # the angle of a small phasor is ill-conditioned: 1e-9 on the phasor itself
                        mag = abs(ref.state[key.replace("_ang", "_magn")]["pu"][i])
                        if mag < 1e-9:   # a bus nothing is injected at: the angle of rounding noise
                            continue
                        atol = min(1e-4, 1e-9 / mag) * (180.0 / np.pi if unit == "degree" else 1.0)
                    npt.assert_allclose(float(ours[unit][i][0]), v, rtol=1e-9, atol=atol, err_msg="%s %s %s" % (key, unit, i))
                    n_checked += 1
    assert n_checked > 500


@pytest.mark.skipif(not _reference_available(), reason="reference checkout not present (GPU box)")
def test_bad_observation_specs_fail_like_the_live_reference():
    """Observation arguments the reference refuses (anm_env.py:497-545, 193-233): same exception class at
    construction, or -- where the reference only fails later, in reset() -- a failure at the same stage."""
    import ref_harness

    ref_harness.load_reference()
    from gym_anm import ANMEnv as Ref
    from gym_anm_amd import ANMEnv as Ours
    from gym_anm_amd import networks

    net = networks.anm6_network()
    extra = _KW(net)
    bad = [
        42,                                             # neither "state", nor a list, nor callable
        "full",                                         # an unknown string
        [("foo_p", "all", "MW")],                       # unknown quantity with 'all'
        [("bus_p", "all", "kV")],                       # unit of another quantity
        [("dev_p", [0, 99], "MW")],                     # unknown device id
        [("branch_s", [(0, 5)], "pu")],                 # unknown branch
        [("des_soc", [2], "MWh")],                      # not a storage unit
    ]

    def outcome(cls, obs, ex):
        class Task(cls):
            def __init__(self):
                super().__init__(net, obs, 1, 0.25, 0.9, 100, np.array([[0, 10]]), (1, 100), 1, **ex)

            def init_state(self):
                s = np.zeros(self.state_N)
                s[-1] = 3
                return s

            def next_vars(self, s_t):
                return np.array([-1.0, -2.0, -3.0, 10.0, 10.0, 4.0])

        try:
            env = Task()
        except Exception as e:  # noqa: BLE001
            return "init", type(e).__name__
        try:
            env.reset()
        except Exception as e:  # noqa: BLE001
            return "reset", type(e).__name__
        return "ok", None

    for obs in bad:
        r, o = outcome(Ref, obs, {}), outcome(Ours, obs, extra)
        assert r[0] != "ok", obs
        assert o[0] == r[0], (obs, r, o)
        if r[1] not in ("KeyError", "TypeError", "IndexError", "ValueError", "AttributeError"):   # the reference's own classes
            assert o[1] == r[1], (obs, r, o)


def test_single_environment_class_with_callable_observation():
    pc.single_env_with_callable_observation(_KW)


def test_bounds_hook_reads_aux_bounds():
    pc.bounds_hook_reads_aux_bounds(_KW)


def test_convergence_flags_after_a_masked_reset():
    pc.convergence_flags_after_a_masked_reset(_KW)


def test_general_step_with_rows_wider_than_the_default_lds_limit():
    pc.general_step_with_wide_rows(_KW)


def test_dump_abs_and_arg_against_numpy():
    """|z| and arg z of the electrical-state dump (csrc/anm_device.hpp: dump_abs, dump_arg -- one division and a degree-9
    polynomial instead of hypot / atan2) against np.abs / np.angle (simulator.py:551-636) over the plane, on the axes, at
    the origin and with NaNs."""
    import ctypes as C

    lib = hostsim_backend(NetworkModel(NETS["anm6"], 0.25, 100).topology()).lib
    fn = lib.anm_hostsim_dump_abs_arg
    fn.restype, fn.argtypes = None, [C.c_int64] + [C.POINTER(C.c_double)] * 4
    rng = np.random.default_rng(7)
    n = 400000
    x = rng.normal(size=n) * 10.0 ** rng.uniform(-6, 6, size=n)
    y = rng.normal(size=n) * 10.0 ** rng.uniform(-6, 6, size=n)
    edge = np.array([[1, 0], [-1, 0], [0, 1], [0, -1], [0, 0], [1, 1], [-1, -1], [-1, 1], [1, -1], [1e-300, 1e-300], [3, 4],
                     [np.tan(np.pi / 8), 1.0], [1.0, np.tan(np.pi / 8)], [np.nan, 1.0], [1.0, np.nan]], dtype=float)
    x = np.concatenate((x, edge[:, 0])); y = np.concatenate((y, edge[:, 1]))
    mag, ang = np.empty_like(x), np.empty_like(x)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    fn(x.size, P(x), P(y), P(mag), P(ang))
    z = x + 1j * y
    fin = np.isfinite(x) & np.isfinite(y)
    assert np.max(np.abs(ang[fin] - np.angle(z[fin]))) < 2e-15
    big = fin & (np.maximum(np.abs(x), np.abs(y)) > 1e-150)   # (per-unit quantities: |z| is formed without scaling against underflow)
    np.testing.assert_allclose(mag[big], np.abs(z[big]), rtol=1e-15, atol=0)
    assert np.isnan(ang[~fin]).all() and np.isnan(mag[~fin]).all()
    # zeros, signs included: np.angle / np.arctan2 distinguish (+0, +-0) -> +-0 from (-0, +-0) -> +-pi
    xs = np.array([0.0, 0.0, -0.0, -0.0, -0.0, 0.0, 5e-324, -5e-324])
    ys = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 0.0, 0.0])
    m2, a2 = np.empty_like(xs), np.empty_like(xs)
    fn(xs.size, P(xs), P(ys), P(m2), P(a2))
    ref = np.arctan2(ys, xs)
    np.testing.assert_array_equal(a2, ref)
    np.testing.assert_array_equal(np.signbit(a2), np.signbit(ref))
