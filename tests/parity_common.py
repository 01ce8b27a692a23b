"""Shared parity checks: run the batched boundary on a backend (GPU library or the host test
double) and compare with golden vectors / the oracle.  Tolerances: bit-exact for indices, flags
and Newton iteration counts; 1e-9 p.u. for floating point (BASELINE.json asks for 1e-6)."""
import os

import numpy as np
import pytest
import numpy.testing as npt
import torch

from gym_anm_amd import networks
from gym_anm_amd.simulator import BatchedSimulator

from conftest import GOLDEN

ATOL = 1e-9


def golden_nets():
    p = np.load(os.path.join(GOLDEN, "3bus_tx_params.npz"))
    nets = {
        "anm6": networks.anm6_network(),
        "2bus": networks.two_bus_network(),
        "3bus": networks.three_bus_loop_network(gen_max=1.5),
        "case30": networks.synthetic_radial_network(30, 0),
    }
    for k in range(10):
        nets["3bus_tx%d" % k] = networks.three_bus_loop_network(p["tap"][k], p["shift"][k], gen_max=1.0)
    return nets


def full_slices(sim):
    off, cnt = sim.full_offsets, sim.full_counts
    return {k: slice(off[k], off[k] + cnt[k]) for k in off}


def check_transition_against_golden(name, net, device, backend=None, precision="f64", atol=ATOL, check_iters=True,
                                    impl=None):
    g = np.load(os.path.join(GOLDEN, "transition_%s.npz" % name))
    M = len(g["n_iter"])
    sim = BatchedSimulator(net, float(g["delta_t"]), float(g["lamb"]), num_envs=M, device=device,
                           precision=precision, impl=impl, _backend=backend)  # fmt: skip
    npt.assert_allclose(sim.device_ybus(), g["Y_bus"], rtol=1e-15, atol=0)
    sim.soc.copy_(torch.as_tensor(g["soc0"]))
    state, r, e, p, conv = sim.transition(g["P_load"], g["P_pot"], g["P_set"], g["Q_set"])
    conv = conv.cpu().numpy()
    full = sim.full.cpu().numpy()
    sl = full_slices(sim)
    ok = g["converged"].astype(bool)
    # flags: exact
    npt.assert_array_equal(conv, ok)
    if check_iters:
        npt.assert_array_equal(sim.nr_iters.cpu().numpy()[ok], g["n_iter"][ok])
    # post-projection injections (independent of the power flow): all cases, slack excluded
    slack = sim.model.slack_dev
    keep = [k for k in range(sim.N_device) if k != slack]
    npt.assert_allclose(full[:, sl["dev_p"]][:, keep], g["dev_p"][:, keep], rtol=0, atol=1e-12)
    npt.assert_allclose(full[:, sl["dev_q"]][:, keep], g["dev_q"][:, keep], rtol=0, atol=1e-12)
    npt.assert_allclose(sim.soc.cpu().numpy(), g["soc_after"], rtol=0, atol=1e-12)
    npt.assert_allclose(full[:, sl["gen_p_max"]], g["p_pot"], rtol=0, atol=1e-15)
    # electrical quantities of the converged cases
    cmp = {
        "dev_p": g["dev_p"], "dev_q": g["dev_q"], "bus_p": g["bus_p"], "bus_q": g["bus_q"],
        "bus_v_magn": np.abs(g["V"]), "bus_v_ang": np.angle(g["V"]), "bus_i_magn": np.abs(g["I"]),
        "branch_p": g["br_p_from"], "branch_q": g["br_q_from"], "branch_s": g["br_s"],
        "branch_i_magn": (np.sign(g["br_i_from"]).real * np.abs(g["br_i_from"])),
    }  # fmt: skip
    for k, ref in cmp.items():
        npt.assert_allclose(full[:, sl[k]][ok], ref[ok], rtol=0, atol=atol, err_msg="%s %s" % (name, k))
    # angles of currents: ill-conditioned for a ~zero current, compared modulo 2 pi elsewhere
    for key, ref in (("bus_i_ang", g["I"]), ("branch_i_ang", g["br_i_from"])):
        d = np.abs(np.angle(np.exp(1j * (full[:, sl[key]] - np.angle(ref))))) * np.minimum(np.abs(ref), 1.0)
        assert d[ok].max(initial=0.0) < atol * 10, key
    for t, ref in ((r, g["reward"]), (e, g["e_loss"]), (p, g["penalty"])):
        npt.assert_allclose(t.cpu().numpy()[ok], ref[ok], rtol=1e-9, atol=atol * 100)
    # the StateView reproduces _gather_state's units
    st = sim.state
    did = sim.model.dev_ids[1]
    npt.assert_allclose(st["dev_p"]["MW"][did].cpu().numpy()[ok], g["dev_p"][ok, 1] * sim.baseMVA, atol=1e-7)
    return sim


def run_episodes(env_factory, n_steps=150):
    """ANM6EasyVec on 6 environments, one per golden episode (reset with the recorded draws,
    recorded actions, masked re-resets after termination) -> compare every step."""
    g = np.load(os.path.join(GOLDEN, "anm6easy_episodes.npz"))
    n_ep = len(g["seed"])
    env = env_factory(n_ep)
    dev = env.device
    npt.assert_array_equal(env.observation_space.low, g["obs_low"])
    npt.assert_array_equal(env.observation_space.high, g["obs_high"])
    npt.assert_array_equal(env.action_space.low, g["act_low"])
    npt.assert_array_equal(env.action_space.high, g["act_high"])
    draw_off = np.concatenate(([0], np.cumsum(g["init_draws_count"])))
    reset_off = np.concatenate(([0], np.cumsum(g["reset_at_count"])))
    ptr = [0] * n_ep
    s0 = np.stack([g["init_draws"][draw_off[e]] for e in range(n_ep)])
    obs, _ = env.reset(options={"init_state": torch.as_tensor(s0, device=dev)})
    npt.assert_allclose(obs.cpu().numpy(), g["obs0"], rtol=0, atol=1e-8)
    for e in range(n_ep):
        ptr[e] = 1
    for t in range(n_steps):
        a = torch.as_tensor(g["actions"][:, t], device=dev)
        obs, r, term, trunc, _ = env.step(a)
        npt.assert_array_equal(term.cpu().numpy(), g["terminated"][:, t], err_msg="step %d" % t)
        npt.assert_allclose(obs.cpu().numpy(), g["obs"][:, t], rtol=0, atol=1e-7, err_msg="step %d" % t)
        npt.assert_allclose(env.state.cpu().numpy(), g["state"][:, t], rtol=0, atol=1e-7)
        npt.assert_allclose(r.cpu().numpy(), g["reward"][:, t], rtol=1e-9, atol=1e-8)
        npt.assert_allclose(env.e_loss.cpu().numpy(), g["e_loss"][:, t], rtol=1e-9, atol=1e-9)
        npt.assert_allclose(env.penalty.cpu().numpy(), g["penalty"][:, t], rtol=1e-9, atol=1e-8)
        ok = ~g["terminated"][:, t]
        npt.assert_array_equal(env.simulator.nr_iters.cpu().numpy()[ok], g["n_iter"][:, t][ok])
        assert not bool(trunc.any())
        tm = term.cpu().numpy()
        if tm.any():
            s0 = np.zeros((n_ep, 18))
            for e in np.where(tm)[0]:
                s0[e] = g["init_draws"][draw_off[e] + ptr[e]]
                ptr[e] += 1
            obs, _ = env.reset(options={"init_state": torch.as_tensor(s0, device=dev), "mask": torch.as_tensor(tm)})
            for e in np.where(tm)[0]:
                j = list(g["reset_at"][reset_off[e] : reset_off[e + 1]]).index(t)
                npt.assert_allclose(obs[e].cpu().numpy(), g["reset_obs"][reset_off[e] + j], rtol=0, atol=1e-8)
    npt.assert_array_equal(env.timestep.cpu().numpy() > 0, np.ones(n_ep, bool))
    return env


# ---------------------------------------------------------------------------------------------------
# round 2: bodies shared by the GPU tier (tests/test_gpu_headline.py) and the host test double
# (tests/test_hostsim_parity.py).  `kw(net)` returns the backend keywords (device=, _backend=).
# ---------------------------------------------------------------------------------------------------
def uniform_actions(env, gen):
    dev = env.device
    lo = torch.as_tensor(env.action_space.low, device=dev)
    hi = torch.as_tensor(env.action_space.high, device=dev)
    return lo + (hi - lo) * torch.rand((env.num_envs, lo.numel()), generator=gen, dtype=torch.float64, device=dev)


def headline_replay(kw, E_, T, n_random, n_collapsed, twin_kw=None, check_env=None, **env_kw):
    """ANM6EasyVec(tol=1e-6, autoreset) for T steps; a seeded sample plus environments that collapsed early
    are replayed by OracleEnv(tol=1e-6), autoreset draws included: observation <= 1e-9, reward rtol 1e-9,
    terminated and Newton iteration counts exact.
    twin_kw: a second batch of the same size built with these keywords instead of env_kw (another step mode: e.g. the
    one-launch in-wave hand-over next to the automatic policy) is stepped beside the first with the same actions;
    every output of every step must be bit-identical.  check_env(env): assertions on the policy the batch chose."""
    import anm_oracle as O
    from gym_anm_amd import rng
    from gym_anm_amd.envs import ANM6EasyVec

    net = networks.anm6_network()
    env = ANM6EasyVec(num_envs=E_, seed=1234, tol=1e-6, autoreset=True, **kw(net), **env_kw)
    dev = env.device
    env.check_actions = False
    env.reset(seed=1234)
    if check_env is not None:
        check_env(env)
    twin = None
    if twin_kw is not None:
        twin = ANM6EasyVec(num_envs=E_, seed=1234, tol=1e-6, autoreset=True, **kw(net), **twin_kw)
        twin.check_actions = False
        twin.reset(seed=1234)
        assert torch.equal(twin.state, env.state) and torch.equal(twin.simulator.soc, env.simulator.soc)
    state0, soc0 = env.state.clone(), env.simulator.soc.clone()
    gen = torch.Generator(device=dev).manual_seed(99)
    rec = {k: [] for k in ("a", "obs", "r", "term", "it", "rc", "el", "pen")}
    for t in range(T):
        a = uniform_actions(env, gen)
        rec["rc"].append(env._reset_count.clone())
        obs, r, term, _, _ = env.step(a)
        if twin is not None:
            o2, r2, t2, _, _ = twin.step(a)
            for name, x, y in (("obs", obs, o2), ("reward", r, r2), ("terminated", term, t2), ("state", env.state, twin.state),
                               ("nr_iters", env.simulator.nr_iters, twin.simulator.nr_iters), ("e_loss", env.e_loss, twin.e_loss),
                               ("penalty", env.penalty, twin.penalty), ("soc", env.simulator.soc, twin.simulator.soc),
                               ("timestep", env.timestep, twin.timestep), ("reset_count", env._reset_count, twin._reset_count)):  # fmt: skip
                assert torch.equal(x, y), "step %d: %s differs between the two step modes" % (t, name)
        for k, v in zip(("a", "obs", "r", "term", "it", "el", "pen"),
                        (a, obs, r, term, env.simulator.nr_iters, env.e_loss, env.penalty)):  # fmt: skip
            rec[k].append(v.clone())
    collapsed = torch.nonzero(torch.stack(rec["term"])[: T - 2].any(dim=0))[:, 0].cpu().numpy()
    sample = np.unique(np.concatenate((np.random.default_rng(0).choice(E_, n_random, replace=False),
                                       collapsed[:n_collapsed])))  # fmt: skip
    idx = torch.as_tensor(sample, device=dev)
    R = {k: torch.stack([x[idx] for x in v]).cpu().numpy() for k, v in rec.items()}
    s0, c0 = state0[idx].cpu().numpy(), soc0[idx].cpu().numpy()
    n_reset = n_term = 0
    for j, e in enumerate(sample):
        orc = O.OracleEnv(net, sparse=False, tol=1e-6)
        orc.load_state(s0[j], c0[j])
        for t in range(T):
            if orc.terminated:  # Gymnasium next-step autoreset: this call returns the first observation
                s_init = rng.series_init_state(env.simulator.model, env._series, 1234, int(e), int(R["rc"][t][j]))
                o, conv = orc.reset_to(s_init)
                assert conv and not R["term"][t][j]
                assert R["r"][t][j] == 0.0 and R["el"][t][j] == 0.0 and R["pen"][t][j] == 0.0
                n_reset += 1
            else:
                o, r, term = orc.step(R["a"][t][j])
                assert term == bool(R["term"][t][j]), (e, t)
                npt.assert_allclose(R["r"][t][j], r, rtol=1e-9, atol=1e-12)
                if term:
                    n_term += 1
                    assert not R["obs"][t][j].any()
                    continue
                npt.assert_allclose(R["el"][t][j], orc.e_loss, rtol=1e-9, atol=1e-12)
                npt.assert_allclose(R["pen"][t][j], orc.penalty, rtol=1e-9, atol=1e-10)
            npt.assert_allclose(R["obs"][t][j], o, rtol=0, atol=1e-9, err_msg="env %d step %d" % (e, t))
            assert int(R["it"][t][j]) == orc.last["n_iter"], (e, t)
    return len(sample), n_term, n_reset


def _check_reset_outputs(sim, g, conv):
    ok = g["converged"].astype(bool)
    npt.assert_array_equal(conv, ok)  # flags exact, the non-converged ones included
    assert (~ok).any()
    full = sim.full.cpu().numpy()
    sl = full_slices(sim)
    slack = sim.model.slack_dev
    keep = [k for k in range(sim.N_device) if k != slack]
    npt.assert_allclose(full[:, sl["dev_p"]][:, keep], g["dev_p"][:, keep], rtol=0, atol=1e-12)
    npt.assert_allclose(full[:, sl["dev_q"]][:, keep], g["dev_q"][:, keep], rtol=0, atol=1e-12)
    npt.assert_allclose(sim.soc.cpu().numpy(), g["soc_after"], rtol=0, atol=1e-13)
    npt.assert_array_equal(sim.nr_iters.cpu().numpy()[ok], g["n_iter"][ok])
    for key, ref in (("dev_p", g["dev_p"]), ("dev_q", g["dev_q"]), ("bus_v_magn", np.abs(g["V"])),
                     ("bus_v_ang", np.angle(g["V"])), ("branch_s", g["br_s"]), ("branch_p", g["br_p_from"]),
                     ("bus_i_magn", np.abs(g["I"]))):  # fmt: skip
        npt.assert_allclose(full[:, sl[key]][ok], ref[ok], rtol=0, atol=1e-9, err_msg=key)


def reset_golden(name, kw, impl=None):
    """tests/golden/reset_<name>.npz (recorded from the reference's Simulator.reset, simulator.py:225-293)
    through BatchedSimulator.reset and through ANMEnv.reset -> anm_reset_f64 (anm_env.py:266-311 tail)."""
    from gym_anm_amd.envs.anm_env import BatchedANMEnv

    net = networks.anm6_network() if name == "anm6" else networks.three_bus_loop_network(base_mva=10, gen_max=100.0)
    g = np.load(os.path.join(GOLDEN, "reset_%s.npz" % name))
    M = len(g["n_iter"])
    extra = {} if impl is None else {"impl": impl}
    sim = BatchedSimulator(net, float(g["delta_t"]), float(g["lamb"]), num_envs=M, **kw(net), **extra)
    if impl is not None:
        assert sim.impl == impl
    conv = sim.reset(g["init_state"])
    _check_reset_outputs(sim, g, conv.cpu().numpy())

    class Env(BatchedANMEnv):
        def __init__(self, **k):
            super().__init__(net, "state", 1, float(g["delta_t"]), 0.995, float(g["lamb"]), **k)

    env = Env(num_envs=M, **kw(net), **extra)
    env._need_full = True  # ask the reset kernel for the electrical-state dump as well
    obs, _ = env.reset(options={"init_state": g["init_state"]})
    _check_reset_outputs(env.simulator, g, env.pfe_converged.cpu().numpy())
    ok = g["converged"].astype(bool)
    npt.assert_array_equal(env.terminated.cpu().numpy(), ~ok)  # explicit non-converging state: terminated
    b = env.simulator.baseMVA
    st = np.concatenate((g["dev_p"] * b, g["dev_q"] * b, g["soc_after"] * b, g["p_pot"] * b, g["init_state"][:, -1:]), 1)
    npt.assert_allclose(env.state.cpu().numpy()[ok], st[ok], rtol=0, atol=1e-7)
    lo, hi = env.observation_space.low, env.observation_space.high
    npt.assert_allclose(obs.cpu().numpy()[ok], np.clip(st, lo, hi)[ok], rtol=0, atol=1e-7)
    assert bool((env.timestep == 0).all())
    return env


def sharded_equal_unsharded(kw, E_, T, **env_kw):
    """BASELINE.json config 5's partition rule: environments [0, E/2) and [E/2, E) stepped as two batches with
    env_offset 0 | E/2 give bit-identical outputs to the batch of E, autoreset (device RNG keyed by the
    global environment index) included."""
    from gym_anm_amd.envs import ANM6EasyVec

    net = networks.anm6_network()
    H = E_ // 2
    mk = lambda n, off: ANM6EasyVec(num_envs=n, seed=7, tol=1e-6, autoreset=True, env_offset=off, **kw(net), **env_kw)
    whole, parts = mk(E_, 0), [mk(H, 0), mk(H, H)]
    for env in [whole] + parts:
        env.check_actions = False
        env.reset(seed=7, options={"sampler": "device"})
    gen = torch.Generator(device=whole.device).manual_seed(8)
    fields = lambda env, out: (out[0], out[1], out[2], env.state, env.simulator.soc, env.simulator.nr_iters,
                               env.timestep, env._reset_count, env.e_loss, env.penalty)  # fmt: skip
    n_term = 0
    for t in range(T):
        a = uniform_actions(whole, gen)
        ref = fields(whole, whole.step(a))
        n_term += int(ref[2].sum())
        for k, env in enumerate(parts):
            cur = fields(env, env.step(a[k * H : (k + 1) * H].contiguous()))
            for x, y in zip(cur, ref):
                assert torch.equal(x, y[k * H : (k + 1) * H]), (t, k)
    return n_term, int(whole._reset_count.max())


def callable_observation_and_two_aux(kw, E_=96, T=24):
    """A task with K = 2 auxiliary variables, a host next_vars() hook and a callable observation
    (anm_env.py:176-191, 372-374, 511-514) against the oracle."""
    import anm_oracle as O
    from gym_anm_amd.envs.anm6 import anm6easy_series
    from gym_anm_amd.envs.anm_env import BatchedANMEnv

    net = networks.anm6_network()
    tab = anm6easy_series()

    def next_vars_np(s):  # aux0: time index; aux1: a counter that modulates the generation potential
        t = int((s[-2] + 1) % 96)
        w = 0.5 + 0.5 * np.cos(0.3 * (s[-1] + 1.0))
        return np.concatenate((tab[:3, t], tab[3:, t] * w, [t, s[-1] + 1.0]))

    class Task(BatchedANMEnv):
        def __init__(self, **k):
            obs = lambda s: torch.stack((s[:, 0] + s[:, 2], s[:, 14] * 0.01, s[:, -1], s[:, 7:14].sum(dim=1)), dim=1)
            super().__init__(net, obs, 2, 0.25, 0.995, 100, aux_bounds=np.array([[0, 95], [0, 1e6]]),
                             costs_clipping=(1, 100), **k)  # fmt: skip

        def next_vars(self, s_t):
            t = torch.remainder(s_t[:, -2] + 1, 96).long()
            w = 0.5 + 0.5 * torch.cos(0.3 * (s_t[:, -1] + 1.0))
            T_ = torch.as_tensor(tab, device=s_t.device)
            return torch.cat((T_[:3, t].T, T_[3:, t].T * w[:, None], t[:, None].double(), (s_t[:, -1] + 1.0)[:, None]), 1)

    env = Task(num_envs=E_, seed=5, **kw(net))
    dev = env.device
    assert env.observation_space is None  # unknown until the first reset (anm_env.py:303-309)
    rng_ = np.random.default_rng(12)
    s0 = np.zeros((E_, env.state_N))
    t0 = rng_.integers(0, 96, E_)
    s0[:, [1, 3, 5]] = tab[:3, t0].T
    s0[:, [2, 4]] = tab[3:, t0].T
    s0[:, 15:17] = tab[3:, t0].T
    s0[:, 14] = rng_.uniform(0, 100, E_)
    s0[:, 17], s0[:, 18] = t0, rng_.integers(0, 50, E_)
    obs, _ = env.reset(options={"init_state": s0})
    assert obs.shape == (E_, 4) and env.observation_space.shape == (4,) and np.isinf(env.observation_space.high).all()
    oracles = []
    for e in range(E_):
        o = O.OracleEnv(net, sparse=False, aux_bounds=((0, 95), (0, 1e6)), next_vars=next_vars_np)
        _, conv = o.reset_to(s0[e])
        assert conv == (not bool(env.terminated[e]))
        oracles.append(o)
    f = lambda s: np.array([s[0] + s[2], s[14] * 0.01, s[-1], s[7:14].sum()])
    npt.assert_allclose(obs.cpu().numpy(), np.stack([f(o.state) for o in oracles]), rtol=0, atol=1e-8)
    lo, hi = env.action_space.low, env.action_space.high
    for t in range(T):
        a = rng_.uniform(lo, hi, (E_, len(lo)))
        obs, r, term, _, _ = env.step(torch.as_tensor(a, device=dev))
        for e, o in enumerate(oracles):
            oo, rr, tt = o.step(a[e])
            assert tt == bool(term[e]), (t, e)
            npt.assert_allclose(float(r[e]), rr, rtol=1e-9, atol=1e-10)
            npt.assert_allclose(env.state[e].cpu().numpy(), o.state, rtol=0, atol=1e-8)
            npt.assert_allclose(obs[e].cpu().numpy(), 0.0 if tt else f(o.state), rtol=0, atol=1e-8)
    assert float(term.double().mean()) < 0.5  # most environments survive the random steps
    return env


def all_state_variables_observation(kw, n_cases=None, **env_kw):
    """A list-form observation naming every one of the reference's 15 STATE_VARIABLES in every unit
    (constants.py:31-48, anm_env.py:562-592, unit scalings of simulator.py:559-616) + the aux variable, on
    the golden ANM6 transitions (one environment per recorded case, fed through a host next_vars hook)."""
    from gym_anm_amd.envs.anm_env import BatchedANMEnv
    from gym_anm_amd.model import STATE_VARIABLES

    net = networks.anm6_network()
    g = np.load(os.path.join(GOLDEN, "transition_anm6.npz"))
    M = len(g["n_iter"]) if n_cases is None else n_cases
    spec = []
    for key, units in STATE_VARIABLES.items():
        if key == "aux":
            continue
        for unit in (units if isinstance(units, tuple) else (units,)):
            spec.append((key, "all", unit))
    spec.append(("aux", "all"))

    class Task(BatchedANMEnv):
        def __init__(self, **k):
            super().__init__(net, spec, 1, float(g["delta_t"]), 0.995, float(g["lamb"]), **k)

        def next_vars(self, s_t):
            return self._vars

    env = Task(num_envs=M, **kw(net), **env_kw)
    dev = env.device
    ep = np.load(os.path.join(GOLDEN, "anm6easy_episodes.npz"))
    obs0, _ = env.reset(options={"init_state": ep["init_draws"][0]})
    assert not bool(env.terminated.any()) and obs0.shape == (M, env.observation_N)
    env.simulator.soc.copy_(torch.as_tensor(g["soc0"][:M], device=dev))
    env._vars = torch.as_tensor(np.concatenate((g["P_load"][:M], g["P_pot"][:M], np.full((M, 1), 7.0)), 1), device=dev)
    ps, qs = g["P_set"][:M], g["Q_set"][:M]  # set-point devices by ascending id: gen 2, gen 4, des 6
    action = np.stack((ps[:, 0], ps[:, 1], qs[:, 0], qs[:, 1], ps[:, 2], qs[:, 2]), 1)
    env.check_actions = False  # recorded set-points reach beyond the action Box on purpose
    obs, r, term, _, _ = env.step(torch.as_tensor(action, device=dev))
    obs = obs.cpu().numpy()
    ok = g["converged"][:M].astype(bool)
    npt.assert_array_equal(term.cpu().numpy(), ~ok)
    assert not obs[~ok].any()  # terminal observation (anm_env.py:442-446)
    m, b = env.simulator.model, env.simulator.baseMVA
    kv = np.asarray(m.bus_baseKV, float)
    ifr = g["br_i_from"][:M]
    pu = {
        "bus_p": g["bus_p"][:M], "bus_q": g["bus_q"][:M], "bus_v_magn": np.abs(g["V"][:M]), "bus_v_ang": np.angle(g["V"][:M]),
        "bus_i_magn": np.abs(g["I"][:M]), "bus_i_ang": np.angle(g["I"][:M]), "dev_p": g["dev_p"][:M], "dev_q": g["dev_q"][:M],
        "des_soc": g["soc_after"][:M], "gen_p_max": g["p_pot"][:M], "branch_p": g["br_p_from"][:M],
        "branch_q": g["br_q_from"][:M], "branch_s": g["br_s"][:M],
        "branch_i_magn": np.sign(ifr).real * np.abs(ifr), "branch_i_ang": np.angle(ifr),
    }  # fmt: skip
    scale = {"pu": 1.0, "rad": 1.0, "MW": b, "MVAr": b, "MVA": b, "MWh": b, "degree": 180 / np.pi, "kV": kv, "kA": b / kv}
    lo, hi = env.observation_space.low, env.observation_space.high
    col = 0
    for key, ids, unit in env.obs_values:
        n = len(ids)
        got = obs[:, col : col + n][ok]
        if key == "aux":
            npt.assert_array_equal(got, 7.0)
        else:
            ref = np.clip(pu[key] * scale[unit], lo[col : col + n], hi[col : col + n])[ok]
            if key.endswith("_i_ang"):  # the angle of a ~zero current is ill-conditioned; compare modulo a turn
                turn = 360.0 if unit == "degree" else 2 * np.pi
                mag = pu[key.replace("_ang", "_magn")][ok]
                d = np.abs((got - ref + turn / 2) % turn - turn / 2) * np.minimum(np.abs(mag), 1.0) / (turn / (2 * np.pi))
                assert d.max(initial=0.0) < 1e-8, (key, unit)
            else:
                tol = 1e-9 * float(np.max(scale[unit])) if unit not in ("pu", "rad") else 1e-9
                npt.assert_allclose(got, ref, rtol=0, atol=max(tol, 1e-9), err_msg="%s %s" % (key, unit))
        col += n
    assert col == obs.shape[1] == env.observation_N
    npt.assert_allclose(r.cpu().numpy()[ok], np.maximum(-(np.sign(g["e_loss"][:M]) * np.minimum(np.abs(g["e_loss"][:M]), np.inf)
                                                           + g["penalty"][:M]), -np.inf)[ok], rtol=1e-9, atol=1e-9)
    return env


def heterogeneous_networks(kw, n_variants=64, per_variant=64, n_check=2, impl=None):
    """Per-environment heterogeneous networks (the reference builds a Simulator per network,
    examples/custom_anm6.py:20): n_variants perturbed copies of ANM6 (same topology, other admittances,
    ratings, limits), `per_variant` environments each; transitions against the oracle built from each
    environment's own network, <= 1e-9; then a few ANM6Easy steps through the environment layer."""
    import anm_oracle as O
    from gym_anm_amd.envs import ANM6EasyVec

    base = networks.anm6_network()
    variants = [networks.perturbed_network(base, 100 + k) for k in range(1, n_variants)]
    E_ = n_variants * per_variant
    env_variant = np.repeat(np.arange(n_variants), per_variant)
    extra = {} if impl is None else {"impl": impl}
    sim = BatchedSimulator(base, 0.25, 100, num_envs=E_, variants=variants, env_variant=env_variant, **kw(base), **extra)
    nets = [base] + variants
    m, b = sim.model, sim.model.baseMVA
    rng = np.random.default_rng(5)
    U = lambda lo, hi, w=1.0: rng.uniform(np.asarray(lo, float) * w, np.asarray(hi, float) * w, size=(E_, len(lo)))
    pl = U(m.dev_p_min[m.load_idx] * b, 0 * m.dev_p_min[m.load_idx])
    pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * b)
    ps = U(m.dev_p_min[m.setp_idx] * b, m.dev_p_max[m.setp_idx] * b, 1.2)
    qs = U(m.dev_q_min[m.setp_idx] * b, m.dev_q_max[m.setp_idx] * b, 1.2)
    soc = U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx], 0.8)
    sim.soc.copy_(torch.as_tensor(soc))
    st, r, el, pen, conv = sim.transition(pl, pp, ps, qs)
    full, sl = sim.full.cpu().numpy(), full_slices(sim)
    n_conv = 0
    for k in range(n_variants):
        n = O.parse_network(nets[k], 0.25, 100)
        for j in range(n_check):
            e = k * per_variant + (j * 37) % per_variant
            out = O.transition(n, pl[e], pp[e], ps[e], qs[e], soc[e], sparse=False)
            assert out["converged"] == bool(conv[e]), (k, e)
            npt.assert_allclose(full[e, sl["dev_p"]][1:], out["dev_p"][1:], rtol=0, atol=1e-12)
            npt.assert_allclose(sim.soc[e].cpu().numpy(), out["soc_after"], rtol=0, atol=1e-12)
            if not out["converged"]:
                continue
            n_conv += 1
            assert int(sim.nr_iters[e]) == out["n_iter"], (k, e)
            npt.assert_allclose(full[e, sl["bus_v_magn"]], np.abs(out["V"]), rtol=0, atol=1e-9)
            npt.assert_allclose(full[e, sl["bus_v_ang"]], np.angle(out["V"]), rtol=0, atol=1e-9)
            npt.assert_allclose(full[e, sl["branch_s"]], out["br_s"], rtol=0, atol=1e-9)
            npt.assert_allclose(full[e, sl["dev_p"]], out["dev_p"], rtol=0, atol=1e-9)
            npt.assert_allclose(float(r[e]), out["reward"], rtol=1e-9, atol=1e-9)
    assert n_conv >= n_variants
    # the class of an environment really is its own: environments of different variants given identical
    # inputs answer differently
    same = np.tile(np.concatenate((pl[:1], pp[:1], ps[:1], qs[:1]), 1), (E_, 1))
    sim.soc.fill_(0.5)
    sim.transition(same[:, :3], same[:, 3:5], same[:, 5:8], same[:, 8:11])
    vm = sim.full[:, sl["bus_v_magn"]][:, 3].cpu().numpy().reshape(n_variants, per_variant)
    assert (np.ptp(vm, axis=1) == 0).all() and len(np.unique(vm[:, 0])) == n_variants
    # environment layer: ANM6Easy on the same variants, a few steps against OracleEnv
    env = ANM6EasyVec(num_envs=E_, seed=4, variants=variants, env_variant=env_variant, **kw(base), **extra)
    env.check_actions = False
    env.reset(seed=4)
    s0, c0 = env.state.cpu().numpy().copy(), env.simulator.soc.cpu().numpy().copy()
    picks = [k * per_variant + 5 for k in range(0, n_variants, max(1, n_variants // 16))]
    oracles = {}
    for e in picks:
        o = O.OracleEnv(nets[e // per_variant], sparse=False)
        o.load_state(s0[e], c0[e])
        # every environment clips its observation to the Box of ITS network, like the reference's one environment per
        # network (anm_env.py:193-233, 313-331): the oracle's own bounds are the class's
        npt.assert_array_equal(env.class_observation_bounds[0][e // per_variant], o.obs_low)
        npt.assert_array_equal(env.class_observation_bounds[1][e // per_variant], o.obs_high)
        oracles[e] = o
    gen = torch.Generator(device=env.device).manual_seed(6)
    n_bites = 0
    for t in range(4):
        a = uniform_actions(env, gen)
        obs, rew, term, _, _ = env.step(a)
        for e, o in oracles.items():
            oo, rr, tt = o.step(a[e].cpu().numpy())
            assert tt == bool(term[e])
            npt.assert_allclose(obs[e].cpu().numpy(), oo, rtol=0, atol=1e-8)
            npt.assert_allclose(float(rew[e]), rr, rtol=1e-9, atol=1e-9)
            if not tt:  # a bound of the class's own Box that class 0's Box would not have applied (or the reverse)
                base_clip = np.clip(o.state, env.observation_space.low, env.observation_space.high)
                n_bites += int(np.abs(base_clip - oo).max() > 1e-6)
    assert n_bites > 0, "no environment was clipped differently by its own Box than by the Box of class 0"
    return sim


def reference_custom_obs_space(kw):
    """The reference's own test of a list-form observation (tests/envs/custom_obs_space.py:30-77), restated on the
    batched environment: 3-bus chain with the slack's 90-degree shifter and a tap-2 transformer, baseMVA 10,
    observation [bus_p all MW, dev_q [0, 2] pu, branch_s all pu], K = 0.  Known answers: the expanded obs_values,
    the Box bounds [-200,-10,-50,-20,-3,-inf,-inf] .. [200,0,80,20,3,inf,inf], and the observation vector read
    from the simulator's state."""
    from gym_anm_amd.envs.anm_env import BatchedANMEnv

    N = None
    net = {
        "baseMVA": 10,
        "bus": np.array([[0, 0, 50, 1.1, 0.9], [1, 1, 50, 1.1, 0.9], [2, 1, 100, 1.0, 1.0]]),
        "branch": np.array([[0, 1, 0.1, 0.2, 0.3, 20, 1, 90], [1, 2, 0.4, 0.5, 0.6, 20, 2, 0]]),
        "device": np.array([[0, 0, 0, N, 200, -200, 200, -200, N, N, N, N, N, N, N],
                            [1, 1, -1, 0.2, 0, -10, N, N, N, N, N, N, N, N, N],
                            [2, 2, 2, N, 30, 0, 30, -30, N, N, N, N, N, N, N],
                            [3, 2, 3, N, 50, -50, 50, -50, N, N, N, N, 100, 0, 0.9]], dtype=object),
    }
    observation = [("bus_p", "all", "MW"), ("dev_q", [0, 2], "pu"), ("branch_s", "all", "pu")]
    E_ = 8

    class Task(BatchedANMEnv):
        def __init__(self, **k):
            super().__init__(net, observation, 0, 1, 0.9, 100, **k)

        def init_state(self):
            rng = np.random.default_rng(0)
            return rng.uniform(size=(self.num_envs, self.state_N))

        def next_vars(self, s_t):
            return torch.zeros((self.num_envs, 2), dtype=torch.float64, device=self.device) + torch.as_tensor([-3.0, 20.0], device=self.device)

    env = Task(num_envs=E_, **kw(net))
    assert env.obs_values == [("bus_p", [0, 1, 2], "MW"), ("dev_q", [0, 2], "pu"), ("branch_s", [(0, 1), (1, 2)], "pu")]
    npt.assert_allclose(env.observation_space.low, [-200, -10, -50, -20, -3, -np.inf, -np.inf])
    npt.assert_allclose(env.observation_space.high, [200, 0, 80, 20, 3, np.inf, np.inf])
    # With the 90-degree shifter at the slack no flat-started Newton solve of this network converges -- neither
    # here, nor in the oracle, nor in the reference (whose test file is not collected by its test runner: its
    # `env.reset()` raises after 100 draws).  Same behaviour: the reference's error type.
    from gym_anm_amd import errors
    with pytest.raises(errors.EnvInitializationError):
        env.reset()
    # the observation vector itself: the same task without the shifter
    net = dict(net, branch=np.array([[0, 1, 0.1, 0.2, 0.3, 20, 1, 0], [1, 2, 0.4, 0.5, 0.6, 20, 2, 0]]))
    env = Task(num_envs=E_, track_full=True, **kw(net))   # (track_full: simulator.state follows the steps)
    obs, _ = env.reset()
    assert obs.shape == (E_, 7)
    lo = torch.as_tensor(env.action_space.low, device=env.device)
    hi = torch.as_tensor(env.action_space.high, device=env.device)
    a = torch.as_tensor([[5.0, 0.0, 1.0, 0.5]], dtype=torch.float64, device=env.device).expand(E_, -1).clone()  # P_gen, Q_gen, P_des, Q_des
    assert bool(((a >= lo) & (a <= hi)).all())
    obs, r, term, _, _ = env.step(a)
    # the observation is what the simulator's state says, in the requested units, in the requested order
    st = env.simulator.state
    live = ~term
    want = torch.cat((st.tensor("bus_p", "MW"), st.tensor("dev_q", "pu")[:, [0, 2]], st.tensor("branch_s", "pu")), dim=1)
    want = torch.minimum(torch.maximum(want, torch.as_tensor(env.observation_space.low, device=env.device)),
                         torch.as_tensor(env.observation_space.high, device=env.device))
    npt.assert_allclose(obs[live].cpu().numpy(), want[live].cpu().numpy(), rtol=0, atol=1e-12)
    assert bool(live.any())
    return env


def reference_example_simple_env(kw):
    """The reference's examples/simple_env.py, class body unchanged, on this package's NumPy-facing `ANMEnv`."""
    from gym_anm_amd import ANMEnv

    network = {
        "baseMVA": 100,
        "bus": np.array([[0, 0, 132, 1.0, 1.0], [1, 1, 33, 1.1, 0.9]]),
        "device": np.array([[0, 0, 0, None, 200, -200, 200, -200, None, None, None, None, None, None, None],
                            [1, 1, -1, 0.2, 0, -10, None, None, None, None, None, None, None, None, None]]),
        "branch": np.array([[0, 1, 0.01, 0.1, 0.0, 3, 1, 0]]),
    }
    extra = kw(network)

    class SimpleEnvironment(ANMEnv):
        def __init__(self):
            super().__init__(network, "state", 1, 0.25, 0.9, 100, np.array([[0, 10]]), (1, 100), 1, **extra)

        def init_state(self):
            n_dev, n_des, n_gen = self.simulator.N_device, self.simulator.N_des, self.simulator.N_non_slack_gen
            return np.random.rand(2 * n_dev + n_des + n_gen + self.K)

        def next_vars(self, s_t):
            return np.array([-10 * np.random.rand(1)[0], np.random.randint(0, 10)])

    env = SimpleEnvironment()
    np.random.seed(5)
    o, info = env.reset()
    assert isinstance(o, np.ndarray) and o.shape == (5,) and info == {}
    for t in range(10):
        o, r, terminated, truncated, info = env.step(env.action_space.sample())
        assert isinstance(o, np.ndarray) and o.shape == (5,) and isinstance(r, float) and isinstance(terminated, bool)
        assert truncated is False and env.observation_space.contains(o)
        # the load follows next_vars: dev_p of the load in MW is what the hook returned, within [-10, 0]
        assert -10.0 <= o[1] <= 0.0 and abs(o[3] - 0.2 * o[1]) < 1e-12
    assert env.state.shape == (5,) and env.timestep == 10
    return env


def single_env_with_callable_observation(kw):
    """`gym_anm_amd.ANMEnv` (NumPy-facing, one environment) with a callable observation, an overridden
    observation_bounds() and K = 2 on the 3-bus loop: the observation is the callable of the state vector, the state
    follows the oracle's environment step by step."""
    import anm_oracle as O
    from gym_anm_amd import ANMEnv
    from gym_anm_amd.spaces import Box

    net = networks.three_bus_loop_network(base_mva=10, gen_max=100.0)
    extra = kw(net)
    fn = lambda s: np.array([s[1] + s[2], s[-1] ** 2, np.tanh(s[0])])  # noqa: E731

    class Task(ANMEnv):
        def __init__(self):
            super().__init__(net, fn, 2, 0.5, 0.95, 100, np.array([[0, 50], [-1, 1]]), (10, 200), 4, **extra)
            self._rng = np.random.default_rng(17)

        def init_state(self):
            s = self._rng.uniform(size=self.state_N)
            s[-2:] = [3.0, 0.5]
            return s

        def next_vars(self, s_t):
            return np.array([-3 * self._rng.uniform(), 40 * self._rng.uniform(), 60 * self._rng.uniform(), (s_t[-2] + 1) % 50, -s_t[-1]])

        def observation_bounds(self):
            return Box(low=np.array([-50.0, 0.0, -1.0]), high=np.array([150.0, 1.0, 1.0]))

    env = Task()
    o, _ = env.reset()
    npt.assert_array_equal(env.observation_space.high, [150.0, 1.0, 1.0])
    npt.assert_allclose(o, np.clip(fn(env.state), env.observation_space.low, env.observation_space.high), rtol=0, atol=1e-12)
    # the oracle's environment with the same hooks drawing from a twin of the task's generator: same draws in the
    # same order (init_state until the reset converges, anm_env.py:266-289, then one next_vars per step)
    twin = np.random.default_rng(17)

    def twin_next_vars(s_t):
        return np.array([-3 * twin.uniform(), 40 * twin.uniform(), 60 * twin.uniform(), (s_t[-2] + 1) % 50, -s_t[-1]])

    orc = O.OracleEnv(net, delta_t=0.5, gamma=0.95, lamb=100, costs_clipping=(10, 200), aux_bounds=((0, 50), (-1, 1)),
                      next_vars=twin_next_vars, sparse=False)
    for _ in range(100):
        s0 = twin.uniform(size=env.state_N)
        s0[-2:] = [3.0, 0.5]
        _, ok = orc.reset_to(s0)
        if ok:
            break
    npt.assert_allclose(env.state, orc.state, rtol=0, atol=1e-9)
    rng = np.random.default_rng(2)
    for t in range(25):
        a = rng.uniform(env.action_space.low, env.action_space.high) * 0.2
        o, r, term, trunc, _ = env.step(a)
        _, r_o, term_o = orc.step(a)
        assert isinstance(o, np.ndarray) and o.shape == (3,) and isinstance(r, float) and trunc is False
        assert term == term_o and abs(r - r_o) <= 1e-9 * (1 + abs(r_o)), (t, r, r_o, term, term_o)
        if term:
            break
        npt.assert_allclose(env.state, orc.state, rtol=0, atol=1e-9)          # the state follows the oracle's, step by step
        npt.assert_allclose(o, np.clip(fn(orc.state), env.observation_space.low, env.observation_space.high), rtol=0, atol=1e-9)
        assert abs(env.state[-2] - (3.0 + t + 1) % 50) < 1e-12          # the aux variables follow next_vars
    return env


def large_network_env(kw, n_bus=150, seed=21, n_chords=16, n_steps=12):
    """`gym_anm_amd.ANMEnv` over a network larger than a wavefront (the reference takes any size: simulator.py:113-181),
    reset and stepped next to the oracle's environment with the same hooks: state, reward and termination step by step."""
    import anm_oracle as O
    from gym_anm_amd import ANMEnv

    net = networks.synthetic_meshed_network(n_bus, seed, n_chords)
    extra = kw(net)
    light = 40.0 / n_bus   # (the synthetic feeders carry the same load per bus whatever their size)

    def hooks(gen, model):
        lo = model.dev_p_min[model.load_idx] * model.baseMVA
        hi = model.dev_p_max[model.gen_idx] * model.baseMVA

        def init_state(n):
            s = np.zeros(n)
            D = model.N_device
            s[:D] = gen.uniform(-1.0, 1.0, D) * light
            s[D : 2 * D] = gen.uniform(-0.2, 0.2, D) * light
            s[2 * D : 2 * D + len(model.des_idx)] = gen.uniform(0.0, 5.0, len(model.des_idx))
            s[2 * D + len(model.des_idx) : -1] = gen.uniform(0.0, 1.0, len(model.gen_idx)) * hi
            s[-1] = 7.0
            return s

        def next_vars(s_t):
            return np.concatenate((lo * light * 0.6 * gen.uniform(size=lo.size), hi * gen.uniform(size=hi.size), [(s_t[-1] + 1) % 24]))

        return init_state, next_vars

    class Task(ANMEnv):
        def __init__(self):
            super().__init__(net, "state", 1, 0.25, 0.99, 100, np.array([[0, 24]]), (10, 500), 3, **extra)
            self._init, self._next = hooks(np.random.default_rng(31), self.simulator.model)

        def init_state(self):
            return self._init(self.state_N)

        def next_vars(self, s_t):
            return self._next(s_t)

    env = Task()
    assert env.simulator.impl == "mesh"
    o, _ = env.reset()
    model = env.simulator.model
    t_init, t_next = hooks(np.random.default_rng(31), model)
    orc = O.OracleEnv(net, delta_t=0.25, gamma=0.99, lamb=100, costs_clipping=(10, 500), aux_bounds=((0, 24),), next_vars=t_next,
                      sparse=False)
    for _ in range(100):
        _, ok = orc.reset_to(t_init(env.state_N))
        if ok:
            break
    npt.assert_allclose(env.state, orc.state, rtol=0, atol=1e-9)
    npt.assert_allclose(o, np.clip(orc.state, env.observation_space.low, env.observation_space.high), rtol=0, atol=1e-9)
    rng = np.random.default_rng(4)
    n_ok = 0
    for t in range(n_steps):
        a = rng.uniform(env.action_space.low, env.action_space.high) * light
        o, r, term, trunc, _ = env.step(a)
        _, r_o, term_o = orc.step(a)
        assert term == term_o and abs(r - r_o) <= 1e-9 * (1 + abs(r_o)), (t, r, r_o, term, term_o)
        if term:
            break
        n_ok += 1
        npt.assert_allclose(env.state, orc.state, rtol=0, atol=1e-9)
        npt.assert_allclose(o, np.clip(orc.state, env.observation_space.low, env.observation_space.high), rtol=0, atol=1e-9)
    assert n_ok >= n_steps // 2
    return env


def bounds_hook_reads_aux_bounds(kw):
    """anm_env.py:122-139: aux_bounds and costs_clipping are attributes BEFORE observation_bounds() runs -- the
    reference's own default implementation reads aux_bounds, and so may a user's override."""
    from gym_anm_amd import ANMEnv
    from gym_anm_amd.spaces import Box

    net = networks.three_bus_loop_network(base_mva=10, gen_max=100.0)
    seen = {}

    class Task(ANMEnv):
        def __init__(self):
            super().__init__(net, lambda s: s[-1:], 1, 0.5, 0.95, 100, np.array([[2, 9]]), (10, 200), 4, **kw(net))

        def init_state(self):
            s = np.full(self.state_N, 0.3)
            s[-1] = 4.0
            return s

        def next_vars(self, s_t):
            return np.array([-1.0, 5.0, 5.0, s_t[-1]])

        def observation_bounds(self):
            seen["aux"], seen["clip"] = np.array(self.aux_bounds), tuple(self.costs_clipping)
            return Box(low=np.array([self.aux_bounds[0][0]], dtype=float), high=np.array([self.aux_bounds[0][1]], dtype=float))

    env = Task()
    npt.assert_array_equal(seen["aux"], [[2, 9]])
    assert seen["clip"] == (10, 200)
    npt.assert_array_equal(env.observation_space.low, [2.0])
    o, _ = env.reset()
    assert o[0] == 4.0


def convergence_flags_after_a_masked_reset(kw, E_=512):
    """BatchedANMEnv.pfe_converged / simulator.pfe_converged after a reset of SOME environments: the ones the reset
    touched report the reset's power flow, the others what their last step left (converged <=> not terminated)."""
    import torch

    from gym_anm_amd.envs import ANM6EasyVec

    net = networks.anm6_network()
    env = ANM6EasyVec(num_envs=E_, seed=11, track_full=True, **kw(net))
    env.reset(seed=11)
    g = torch.Generator(device="cpu").manual_seed(3)
    lo, hi = torch.as_tensor(env.action_space.low), torch.as_tensor(env.action_space.high)
    for _ in range(40):
        a = (lo + (hi - lo) * torch.rand((E_, 6), generator=g, dtype=torch.float64)).to(env.device)
        env.step(a)
        if int(env.terminated.sum()) >= 2:
            break
    term = env.terminated.clone()
    assert int(term.sum()) >= 2, "the random agent collapses ~0.5 % of the environments per step"
    idx = torch.nonzero(term).flatten()
    mask = torch.zeros(E_, dtype=torch.bool, device=env.device)
    mask[idx[0]] = True          # reset ONE of the collapsed environments (and one healthy one)
    healthy = torch.nonzero(~term).flatten()[0]
    mask[healthy] = True
    env.reset(options={"mask": mask})
    conv = env.pfe_converged
    assert bool(conv[idx[0]]) and bool(conv[healthy])                       # freshly reset: converged
    assert not bool(conv[idx[1]])                                           # still collapsed, not touched by the reset
    others = ~mask
    assert torch.equal(conv[others], ~term[others])
    assert torch.equal(env.simulator.pfe_converged[others], ~term[others])


def general_step_with_wide_rows(kw):
    """8 buses, 9 devices: the electrical-state rows of 64 environments (113 doubles each) plus their state rows are
    69 632 bytes of dynamic LDS for k_step_general -- more than the 64 KB a workgroup gets by default.  A list
    observation with the dump (the default of the single-environment class) against the oracle."""
    import anm_oracle as O
    import torch

    from gym_anm_amd.envs.anm_env import BatchedANMEnv
    from gym_anm_amd.model import NetworkModel

    net = networks.synthetic_radial_network(8, 3)
    dev = [list(r) for r in net["device"]]
    N_ = None
    while len(dev) < 9:   # more loads until the rows are wide enough
        dev.append([len(dev), 2 + len(dev) % 5, -1, 0.2, 0, -4.0 - len(dev), N_, N_, N_, N_, N_, N_, N_, N_, N_])
    net["device"] = np.array(dev, dtype=object)
    E_ = 192
    spec = [("bus_v_magn", "all", "pu"), ("branch_s", "all", "MVA"), ("dev_p", "all", "MW"), ("bus_v_ang", "all", "rad")]
    m = NetworkModel(net, 0.25, 100)
    n_exo = len(m.load_idx) + len(m.gen_idx)

    class Task(BatchedANMEnv):
        def __init__(self):
            super().__init__(net, spec, 0, 0.25, 0.99, 100, None, (1, 100), 5, num_envs=E_, track_full=True, **kw(net))
            self._g = torch.Generator(device="cpu").manual_seed(9)

        def init_state(self):
            s = torch.zeros((E_, self.state_N), dtype=torch.float64)
            s[:, : 2 * m.N_device] = 0.0
            return s.numpy()

        def next_vars(self, s_t):
            u = torch.rand((E_, n_exo), generator=self._g, dtype=torch.float64)
            lo = torch.as_tensor(np.concatenate((m.dev_p_min[m.load_idx], 0 * m.dev_p_max[m.gen_idx])) * m.baseMVA)
            hi = torch.as_tensor(np.concatenate((0 * m.dev_p_min[m.load_idx], m.dev_p_max[m.gen_idx])) * m.baseMVA)
            self.last_exo = (lo + (hi - lo) * u).numpy()
            return self.last_exo

    env = Task()
    dims_full = env.simulator.full_dim
    assert (64 * ((dims_full + 8) | 1) + 64 * ((env.state_N + 8) | 1)) * 8 > 64 * 1024   # the case this test is about
    env.reset()
    n = O.parse_network(net, 0.25, 100)
    g = torch.Generator(device="cpu").manual_seed(1)
    lo, hi = torch.as_tensor(env.action_space.low), torch.as_tensor(env.action_space.high)
    soc = env.simulator.soc.cpu().numpy().copy()
    for t in range(3):
        a = lo + (hi - lo) * torch.rand((E_, len(lo)), generator=g, dtype=torch.float64) * 0.5
        o, r, term, _, _ = env.step(a.to(env.device))
        full = env.simulator.state
        vm = full.tensor("bus_v_magn", "pu").cpu().numpy()
        for e in range(0, E_, 37):
            ng, nd = len(m.gen_idx), len(m.des_idx)
            P_set = {k: a[e, j].item() for j, k in enumerate(list(m.gen_idx))}
            P_set.update({k: a[e, 2 * ng + j].item() for j, k in enumerate(list(m.des_idx))})
            Q_set = {k: a[e, ng + j].item() for j, k in enumerate(list(m.gen_idx))}
            Q_set.update({k: a[e, 2 * ng + nd + j].item() for j, k in enumerate(list(m.des_idx))})
            out = O.transition(n, env.last_exo[e, : len(m.load_idx)], env.last_exo[e, len(m.load_idx):],
                               [P_set[k] for k in n.setp], [Q_set[k] for k in n.setp], soc[e], sparse=False)
            if out["converged"] and not bool(term[e]):
                npt.assert_allclose(vm[e], np.abs(out["V"]), rtol=0, atol=1e-9)
                npt.assert_allclose(o[e, : m.N_bus].cpu().numpy(), np.clip(np.abs(out["V"]), env.observation_space.low[: m.N_bus],
                                                                          env.observation_space.high[: m.N_bus]), rtol=0, atol=1e-9)
        soc = env.simulator.soc.cpu().numpy().copy()


def per_environment_networks(kw, impl, n_variants=24, E_=192, n_check=48, base=None):
    """A different network in every environment (the reference: one Simulator per environment, examples/custom_anm6.py:20):
    the class changes from one environment to the next, in no order -- served by the lane-group families (one
    environment per lane group, constants by vector loads).  Transitions against the oracle of each environment's own
    network (<= 1e-9, iteration counts exact), then ANM6Easy steps with each environment's own observation Box."""
    import anm_oracle as O
    import torch

    from gym_anm_amd.envs import ANM6EasyVec

    anm6 = base is None     # (base: another network -- the transitions only; the environment layer below is ANM6Easy's)
    base = networks.anm6_network() if anm6 else base
    variants = [networks.perturbed_network(base, 300 + k) for k in range(1, n_variants)]
    nets = [base] + variants
    rng = np.random.default_rng(8)
    env_variant = rng.integers(0, n_variants, E_)
    env_variant[:8] = np.arange(8)                      # (a class change between neighbours of one wavefront for sure)
    sim = BatchedSimulator(base, 0.25, 100, num_envs=E_, variants=variants, env_variant=env_variant, impl=impl, **kw(base))
    assert sim.impl == impl
    m, b = sim.model, sim.model.baseMVA
    U = lambda lo, hi, w=1.0: rng.uniform(np.asarray(lo, float) * w, np.asarray(hi, float) * w, size=(E_, len(lo)))
    pl = U(m.dev_p_min[m.load_idx] * b, 0 * m.dev_p_min[m.load_idx])
    pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * b)
    ps = U(m.dev_p_min[m.setp_idx] * b, m.dev_p_max[m.setp_idx] * b, 1.2)
    qs = U(m.dev_q_min[m.setp_idx] * b, m.dev_q_max[m.setp_idx] * b, 1.2)
    soc = U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx], 0.8)
    sim.soc.copy_(torch.as_tensor(soc))
    st, r, el, pen, conv = sim.transition(pl, pp, ps, qs)
    full, sl = sim.full.cpu().numpy(), full_slices(sim)
    parsed = {}
    n_conv = 0
    for e in range(0, E_, max(1, E_ // n_check)):
        k = int(env_variant[e])
        n = parsed.setdefault(k, O.parse_network(nets[k], 0.25, 100))
        out = O.transition(n, pl[e], pp[e], ps[e], qs[e], soc[e], sparse=False)
        assert out["converged"] == bool(conv[e]), (k, e)
        npt.assert_allclose(full[e, sl["dev_p"]][1:], out["dev_p"][1:], rtol=0, atol=1e-12)
        if not out["converged"]:
            continue
        n_conv += 1
        assert int(sim.nr_iters[e]) == out["n_iter"], (k, e)
        npt.assert_allclose(full[e, sl["bus_v_magn"]], np.abs(out["V"]), rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["bus_v_ang"]], np.angle(out["V"]), rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["branch_s"]], out["br_s"], rtol=0, atol=1e-9)
        npt.assert_allclose(float(r[e]), out["reward"], rtol=1e-9, atol=1e-9)
    assert n_conv >= n_check // 2
    if not anm6:
        return sim
    # the thread-per-environment family cannot serve such an assignment: asked for, the model moves to a lane group
    sim_t = BatchedSimulator(base, 0.25, 100, num_envs=E_, variants=variants, env_variant=env_variant, impl="thread", **kw(base))
    assert sim_t.impl in ("radial", "mesh")
    # environment layer
    env = ANM6EasyVec(num_envs=E_, seed=4, variants=variants, env_variant=env_variant, impl=impl, **kw(base))
    env.check_actions = False
    env.reset(seed=4)
    s0, c0 = env.state.cpu().numpy().copy(), env.simulator.soc.cpu().numpy().copy()
    oracles = {}
    for e in range(0, E_, 13):
        o = O.OracleEnv(nets[int(env_variant[e])], sparse=False)
        o.load_state(s0[e], c0[e])
        npt.assert_array_equal(env.class_observation_bounds[0][int(env_variant[e])], o.obs_low)
        oracles[e] = o
    gen = torch.Generator(device=env.device).manual_seed(6)
    for t in range(3):
        a = uniform_actions(env, gen)
        obs, rew, term, _, _ = env.step(a)
        for e, o in oracles.items():
            oo, rr, tt = o.step(a[e].cpu().numpy())
            assert tt == bool(term[e])
            npt.assert_allclose(obs[e].cpu().numpy(), oo, rtol=0, atol=1e-8)
            npt.assert_allclose(float(rew[e]), rr, rtol=1e-9, atol=1e-9)
    return sim


def structured_network(shape):
    """Shapes the random feeders do not produce (bus 0 the slack; devices of the random feeder of the same size): a hub
    with 20 leaves, a ring, a path, a complete graph, five feeders off the slack, a ladder of two coupled chains."""
    n = {"star": 22, "ring": 17, "path": 25, "complete": 8, "feeders": 21, "ladder": 21}[shape]
    if shape == "star":
        edges = [(0, 1)] + [(1, i) for i in range(2, n)]
    elif shape == "ring":
        edges = [(0, 1)] + [(i, i + 1) for i in range(1, n - 1)] + [(n - 1, 1)]
    elif shape == "path":
        edges = [(i, i + 1) for i in range(n - 1)]
    elif shape == "complete":
        edges = [(0, 1)] + [(i, j) for i in range(1, n) for j in range(i + 1, n)]
    elif shape == "feeders":
        edges = [(0, 1 + 4 * k) for k in range(5)] + [(1 + 4 * k + j, 2 + 4 * k + j) for k in range(5) for j in range(3)]
    else:
        half = (n - 1) // 2
        edges = [(0, 1)] + [(i, i + 1) for i in range(1, half)] + [(half + i, half + i + 1) for i in range(1, half)] + \
                [(i, half + i) for i in range(1, half + 1)]
    net = networks.synthetic_radial_network(n, 1)
    rng = np.random.default_rng(5)
    net["branch"] = np.array([[f, t, float(rng.uniform(0.005, 0.03)), float(rng.uniform(0.03, 0.08)), 0.0, 30.0, 1, 0]
                              for f, t in edges])
    return net


# ======================================================================================
# Newton ITERATES against the reference (tests/golden/iterates_<net>.npz, oracle/make_golden_iterates.py): x_k and
# ||F(x_k)||inf after k = 1 ... 12 iterations of the reference's own _newton_raphson_sparse(lim_iter=k)
# (solve_load_flow.py:176-226).  Final V + iteration count cannot tell an exact Jacobian / linear solve (_dfdx :123-164,
# spsolve :220) from a 1e-7-accurate one -- Newton corrects itself --, an iterate can: an error in step k is in x_k
# at first order.  Tolerances (stated, and measured values are printed by the GPU tier):
#   converging solves   |V_k - V_k^ref| <= 1e-11 p.u. on every bus, every recorded k (cumulative: flat start, cap k)
#   diverging solves    cumulative for k <= 4 at 1e-9 * max(1, |V_k^ref|): these iterates are thrown to |V| ~ 1e3 ... 1e8 and
#                       every further step multiplies the rounding differences it inherits by the local expansion rate of a
#                       chaotic map (the host build, exact 1/x and no contraction, is 6e-12 off at k = 4, 6e-7 at k = 6,
#                       7e-5 at k = 12: ANY other double-precision factorisation would be); beyond k = 4 the comparison
#                       is ONE STEP at a time from the reference's own iterate (newton_one_step, anm_model_bind_nr_start):
#                       all 12 iterates, relative to the larger of the two points, within max(1e-10, 16 eps cond(J(x_{k-1})))
#                       (30-bus feeder: cond reaches 5e7 on these iterates, and LAPACK's dense solve is 2 eps cond from
#                       the reference's SuperLU step there)
#   mismatch            |diff_k - diff_k^ref| <= 1e-6 diff_k^ref + 1e-13 (the floor: rounding of p.u. sums of terms up to ~30)
# ======================================================================================
ITER_NETS = ("anm6", "case30", "3bus", "3bus_tx2", "3bus_tx7")
ITER_ATOL_CONV, ITER_RTOL_DIV, ITER_DIFF_RTOL, ITER_DIFF_ATOL = 1e-11, 1e-9, 1e-6, 1e-13
ONE_STEP_COND_FACTOR = 16
ITER_DIV_CUM_K = 4   # diverging solves: cumulative comparison up to this iterate, one step at a time beyond


def iterate_nets():
    nets = golden_nets()
    return {k: nets[k] for k in ITER_NETS}


def _iterate_reference(g, k):
    """rows that have an iterate k, its V (non-slack buses) and mismatch"""
    rows = g["n_k"] >= k
    V = g["vm_k"][:, k - 1] * np.exp(1j * g["th_k"][:, k - 1])
    return rows, V, g["diff_k"][:, k]


def _compare_iterate(name, k, g, V, it, diff, worst):
    rows, Vref, dref = _iterate_reference(g, k)
    # the reference's loop after at most k iterations: min(n_iter, k) of them were done
    npt.assert_array_equal(it, np.minimum(g["n_iter"], k), err_msg="%s: iteration counts at cap %d" % (name, k))
    div = g["diverging"].astype(bool)
    for sel, lab in ((rows & ~div, "conv"), (rows & div, "div")):
        if not sel.any():
            continue
        a, b = V[sel], Vref[sel]
        assert np.array_equal(np.isnan(a), np.isnan(b)), "%s k=%d: NaN pattern of the iterates" % (name, k)
        fin = np.isfinite(b)
        if lab == "div" and k > ITER_DIV_CUM_K:
            continue
        err = np.abs(a - b)[fin]
        scale = np.ones_like(err) if lab == "conv" else np.maximum(1.0, np.abs(b)[fin])
        rel = float((err / scale).max(initial=0.0))
        worst[lab] = max(worst[lab], rel)
        tol = ITER_ATOL_CONV if lab == "conv" else ITER_RTOL_DIV
        assert rel <= tol, "%s: iterate %d of the %s solves deviates by %.3g (allowed %.1g)" % (name, k, "converging" if lab == "conv" else "diverging", rel, tol)
    if diff is not None:
        a, b = diff[rows], dref[rows]
        assert np.array_equal(np.isnan(a), np.isnan(b)), "%s k=%d: NaN pattern of the mismatch" % (name, k)
        fin = np.isfinite(b) & ~div[rows]   # (a diverging iterate's mismatch is as far off as the iterate: see newton_one_step)
        e = np.abs(a[fin] - b[fin])
        worst["diff"] = max(worst["diff"], float((e / (np.abs(b[fin]) + ITER_DIFF_ATOL / ITER_DIFF_RTOL)).max(initial=0.0)))
        assert np.all(e <= ITER_DIFF_RTOL * np.abs(b[fin]) + ITER_DIFF_ATOL), "%s: mismatch after %d iterations deviates" % (name, k)


def newton_iterates_transition(name, net, device, backend=None, impl=None):
    """anm_transition_f64 with max_iter = k, k = 1 ... 12: V_k from the electrical-state dump, ||F(x_k)||inf through
    anm_model_bind_nr_diff, both against the reference's iterates."""
    g = np.load(os.path.join(GOLDEN, "iterates_%s.npz" % name))
    M = len(g["n_iter"])
    sim = BatchedSimulator(net, float(g["delta_t"]), float(g["lamb"]), num_envs=M, device=device, tol=float(g["x_tol"]),
                           impl=impl, handoff_after=None, _backend=backend)
    diff_t = sim.track_nr_diff()
    sl = full_slices(sim)
    worst = {"conv": 0.0, "div": 0.0, "diff": 0.0}
    for k in range(1, g["vm_k"].shape[1] + 1):
        sim.opts.max_iter = k
        sim.soc.copy_(torch.as_tensor(g["soc0"]))
        sim.transition(g["P_load"], g["P_pot"], g["P_set"], g["Q_set"])
        full = sim.full.cpu().numpy()
        if k == 1:   # what the solver is given: the reference's p, q to the last bits (slack entry: the solver's own result)
            npt.assert_allclose(full[:, sl["bus_p"]][:, 1:], g["bus_p"][:, 1:], rtol=0, atol=1e-12)
            npt.assert_allclose(full[:, sl["bus_q"]][:, 1:], g["bus_q"][:, 1:], rtol=0, atol=1e-12)
        V = (full[:, sl["bus_v_magn"]] * np.exp(1j * full[:, sl["bus_v_ang"]]))[:, 1:]
        _compare_iterate(name, k, g, V, sim.nr_iters.cpu().numpy(), diff_t.cpu().numpy(), worst)
    # the full solve: the mismatch the reference's loop ended with
    sim.opts.max_iter = 100
    sim.soc.copy_(torch.as_tensor(g["soc0"]))
    sim.transition(g["P_load"], g["P_pot"], g["P_set"], g["Q_set"])
    ok = g["converged"].astype(bool)
    npt.assert_array_equal(sim.pfe_converged.cpu().numpy(), ok)
    npt.assert_array_equal(sim.nr_iters.cpu().numpy()[ok], g["n_iter"][ok])
    d = diff_t.cpu().numpy()
    assert np.all(np.abs(d[ok] - g["diff"][ok]) <= ITER_DIFF_RTOL * g["diff"][ok] + ITER_DIFF_ATOL)
    assert np.all(~(d[~ok] <= float(g["x_tol"])))   # not converged: above the tolerance, or NaN
    return worst


def newton_iterates_step(name, net, device, backend=None, handoff_after=None, impl=None):
    """The same iterates through anm_step_f64 (host next_vars, electrical-state dump): the path on which a
    thread-per-environment solve hands over to a lane group of its wavefront -- handoff_after = 0 sends EVERY solve
    there from its first iteration (csrc/anm_group.hpp: the one-step v_rcp_f64 pivot inverse, DPP hand-overs)."""
    from gym_anm_amd.envs.anm_env import BatchedANMEnv

    g = np.load(os.path.join(GOLDEN, "iterates_%s.npz" % name))
    M = len(g["n_iter"])
    exo = torch.as_tensor(np.concatenate([g["P_load"], g["P_pot"], np.zeros((M, 1))], axis=1))

    class Env(BatchedANMEnv):
        def __init__(self):
            super().__init__(net, "state", 1, float(g["delta_t"]), 0.995, float(g["lamb"]), aux_bounds=np.array([[0.0, 10.0]]),
                             num_envs=M, device=device, tol=float(g["x_tol"]), track_full=True, handoff_after=handoff_after,
                             impl=impl, _backend=backend)

        def next_vars(self, s):
            return exo.to(self.device)

    env = Env()
    env.check_actions = False   # the fixtures' set-points go beyond the action Box on purpose (wild cases)
    sim, m = env.simulator, env.simulator.model
    pos = {d: j for j, d in enumerate(m.setp_idx)}
    gi, di = [pos[d] for d in m.gen_idx], [pos[d] for d in m.des_idx]
    act = torch.as_tensor(np.concatenate([g["P_set"][:, gi], g["Q_set"][:, gi], g["P_set"][:, di], g["Q_set"][:, di]], axis=1))
    sl = full_slices(sim)
    worst = {"conv": 0.0, "div": 0.0, "diff": 0.0}
    for k in range(1, g["vm_k"].shape[1] + 1):
        sim.opts.max_iter = k
        sim.soc.copy_(torch.as_tensor(g["soc0"]))
        env._term_u8.zero_()
        env.step(act)
        full = sim.full.cpu().numpy()
        V = (full[:, sl["bus_v_magn"]] * np.exp(1j * full[:, sl["bus_v_ang"]]))[:, 1:]
        _compare_iterate(name, k, g, V, sim.nr_iters.cpu().numpy(), None, worst)
    return worst


def newton_one_step(name, net, device, backend=None, impl=None):
    """ONE Newton step from the reference's own iterate x_{k-1} (anm_model_bind_nr_start, max_iter = 1) against the
    reference's x_k, k = 1 ... 12, every recorded iterate of every case: what a step of THIS Jacobian and THIS linear
    solve does to the very point the reference stepped from.  A diverging solve multiplies rounding differences from
    step to step (the cumulative comparison above is therefore bounded to its first iterates); one step does not."""
    g = np.load(os.path.join(GOLDEN, "iterates_%s.npz" % name))
    M, N1 = len(g["n_iter"]), g["vm_k"].shape[2]
    sim = BatchedSimulator(net, float(g["delta_t"]), float(g["lamb"]), num_envs=M, device=device, tol=float(g["x_tol"]),
                           max_iter=1, impl=impl, handoff_after=None, _backend=backend)
    diff_t = sim.track_nr_diff()
    sl = full_slices(sim)
    div = g["diverging"].astype(bool)
    worst = {"conv": 0.0, "div": 0.0, "diff": 0.0}
    import anm_oracle as _O   # (test infrastructure: the dense Jacobian of the reference's iterate, for its condition number)

    ynet = _O.parse_network(net, float(g["delta_t"]), float(g["lamb"])).Y
    flat = np.concatenate([np.zeros((M, N1)), np.ones((M, N1))], axis=1)
    for k in range(1, g["vm_k"].shape[1] + 1):
        rows, Vref, dref = _iterate_reference(g, k)
        if not rows.any():
            break
        x0 = flat.copy()
        if k > 1:
            x0[rows] = np.concatenate([g["th_k"][rows, k - 2], g["vm_k"][rows, k - 2]], axis=1)
        sim.set_nr_start(x0)
        sim.soc.copy_(torch.as_tensor(g["soc0"]))
        sim.transition(g["P_load"], g["P_pot"], g["P_set"], g["Q_set"])
        full = sim.full.cpu().numpy()
        V = (full[:, sl["bus_v_magn"]] * np.exp(1j * full[:, sl["bus_v_ang"]]))[:, 1:]
        npt.assert_array_equal(sim.nr_iters.cpu().numpy()[rows], 1)
        for sel, lab in ((rows & ~div, "conv"), (rows & div, "div")):
            if not sel.any():
                continue
            a, b = V[sel], Vref[sel]
            assert np.array_equal(np.isnan(a), np.isnan(b)), "%s step %d: NaN pattern" % (name, k)
            fin = np.isfinite(b)
            if lab == "conv":
                rel = float(np.abs(a - b)[fin].max(initial=0.0))
                worst["conv"] = max(worst["conv"], rel)
                assert rel <= ITER_ATOL_CONV, "%s: step %d of the converging solves lands %.3g p.u. from the reference's (allowed %.1g)" % (name, k, rel, ITER_ATOL_CONV)
                continue
            # diverging solves step from points with |V| ~ 1e3 ... 1e8 through Jacobians that are close to singular (that is why
            # they diverge): what ANY double-precision solve can promise is eps * cond(J(x_{k-1})) relative to the larger of
            # the two points.  The allowance is ONE_STEP_COND_FACTOR times that, per case, cond from the oracle's dense
            # Jacobian of the reference's iterate (measured: LAPACK's partial-pivoting solve lands 2 eps cond from the
            # reference's SuperLU step where cond = 2e7; the kernels 0.7 ... 1.5), and never below ITER_RTOL_DIV / 10.
            Vprev = (x0[:, N1:] * np.exp(1j * x0[:, :N1]))[sel]
            scale = np.maximum(1.0, np.maximum(np.abs(b), np.abs(Vprev).max(axis=1, keepdims=True)))
            err = np.where(fin, np.abs(a - b) / scale, 0.0).max(axis=1)
            for j, m_ in enumerate(np.flatnonzero(sel)):
                cond = np.linalg.cond(_O.nr_jacobian(x0[m_], ynet, False))
                allow = max(ITER_RTOL_DIV / 10, ONE_STEP_COND_FACTOR * np.finfo(float).eps * cond)
                worst["div"] = max(worst["div"], float(err[j]))
                if err[j] > ITER_RTOL_DIV / 100:   # (below that the floor of the allowance speaks, not the conditioning)
                    worst["div_over_eps_cond"] = max(worst.get("div_over_eps_cond", 0.0), float(err[j] / (np.finfo(float).eps * cond)))
                assert err[j] <= allow, ("%s: step %d of diverging case %d lands %.3g (relative) from the reference's; allowed %.3g = "
                                         "max(%.0e, %d eps cond(J)), cond(J) = %.3g" % (name, k, m_, err[j], allow, ITER_RTOL_DIV / 10, ONE_STEP_COND_FACTOR, cond))
        d = diff_t.cpu().numpy()
        a, b = d[rows & ~div], dref[rows & ~div]
        if len(b):
            e = np.abs(a - b) / (np.abs(b) + ITER_DIFF_ATOL / ITER_DIFF_RTOL)
            worst["diff"] = max(worst["diff"], float(e.max()))
            assert np.all(np.abs(a - b) <= ITER_DIFF_RTOL * np.abs(b) + ITER_DIFF_ATOL), "%s: mismatch after step %d" % (name, k)
    sim.set_nr_start(None)
    return worst
