"""`MixedBatchedANMEnv`: the Gymnasium surface over environments on DIFFERENT networks in one batch (the reference builds
one ANMEnv per network: gym_anm/envs/anm_env.py:79-156, examples/custom_anm6.py:20).  ANM6 (thread-per-environment
kernels), the 3-bus loop (thread-per-environment, not a tree: no lane-group hand-over), a meshed 20-bus network (general
lane-group kernel, no library of its own) and config 4's 30-bus feeder (tree kernel) dealt at random to 16 384 environments,
40 autoresetting steps on the tasks' own streams: sampled environments of EVERY network replayed by their own OracleEnv --
initial draws, autoreset draws, observation <= 1e-9, reward rtol 1e-9, terminated and Newton iteration counts exact -- and the
padding of every row untouched."""
import numpy as np
import numpy.testing as npt
import pytest
import torch

import anm_oracle as O
from gym_anm_amd import networks, rng
from gym_anm_amd.envs import MixedBatchedANMEnv
from gym_anm_amd.envs.anm6 import anm6easy_series
from gym_anm_amd.model import NetworkModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def daily_series(net, period, seed):
    """a periodic table for any network: every load between 25 % and 85 % of its largest demand, every generator's
    potential between 10 % and 100 % of its capacity, smooth over the period, phases by device"""
    m = NetworkModel(net, 0.25, 100)
    b = m.baseMVA
    r = np.random.default_rng(seed)
    t = np.arange(period) / period
    rows = []
    for k in m.load_idx:
        ph, lo, hi = r.uniform(0, 1), 0.25, 0.85
        rows.append(m.dev_p_min[k] * b * (lo + (hi - lo) * 0.5 * (1 + np.sin(2 * np.pi * (t + ph)))))
    for k in m.gen_idx:
        ph = r.uniform(0, 1)
        rows.append(m.dev_p_max[k] * b * (0.1 + 0.9 * 0.5 * (1 + np.sin(2 * np.pi * (t + ph)))))
    return np.array(rows)


def make_tasks():
    nets = [networks.anm6_network(), networks.three_bus_loop_network(gen_max=1.5), networks.synthetic_meshed_network(20, 3, 6),
            networks.synthetic_radial_network(30, 0)]
    series = [anm6easy_series(), daily_series(nets[1], 24, 1), daily_series(nets[2], 48, 2), daily_series(nets[3], 96, 3)]
    dts = [0.25, 0.5, 0.25, 0.25]
    return [dict(network=n, series=s, delta_t=dt, gamma=0.995, lamb=100, costs_clipping=(1, 100)) for n, s, dt in zip(nets, series, dts)]


def test_four_topologies_in_one_batch_against_their_own_oracles():
    tasks = make_tasks()
    E_, T, SEED = 16384, 40, 2025
    env_task = np.random.default_rng(5).integers(0, len(tasks), E_)
    env = MixedBatchedANMEnv(tasks, env_task, device=DEV, seed=SEED, tol=1e-6, autoreset=True)
    assert env.impls == ["thread", "thread", "mesh", "radial"]
    env.check_actions = False
    W, A = env.W, env.A
    # poison the padding: nothing may ever touch it
    env.state.fill_(-777.0)
    env._obs.fill_(-777.0)
    obs0, _ = env.reset(seed=SEED)
    assert not bool(env.terminated.any())
    state0, soc0, rc0 = env.state.clone(), env.soc.clone(), env._reset_count.clone()
    lo, hi = torch.as_tensor(env.action_space.low, device=DEV), torch.as_tensor(env.action_space.high, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(77)
    rec = {k: [] for k in ("a", "obs", "r", "term", "it", "rc", "el", "pen")}
    for t in range(T):
        a = lo + (hi - lo) * torch.rand((E_, A), generator=gen, dtype=torch.float64, device=DEV)
        rec["rc"].append(env._reset_count.clone())
        obs, r, term, _, _ = env.step(a)
        for k, v in zip(("a", "obs", "r", "term", "it", "el", "pen"), (a, obs, r, term, env.nr_iters, env.e_loss, env.penalty)):
            rec[k].append(v.clone())
    for k in range(len(tasks)):   # the padding of every row, state and observation
        rows = env.task_rows(k)
        assert bool((env.state[rows][:, env.state_N[k]:] == -777.0).all()) and bool((env._obs[rows][:, env.state_N[k]:] == -777.0).all())
    terms = torch.stack(rec["term"])
    n_checked = n_term = n_reset = 0
    for k, task in enumerate(tasks):
        rows = env.task_rows(k).cpu().numpy()
        assert rows.size > 3000
        collapsed = rows[terms[: T - 2][:, rows].any(dim=0).cpu().numpy()]
        sample = np.unique(np.concatenate((np.random.default_rng(k).choice(rows, 10, replace=False), collapsed[:6])))
        idx = torch.as_tensor(sample, device=DEV)
        R = {key: torch.stack([x[idx] for x in v]).cpu().numpy() for key, v in rec.items()}
        S, NA = env.state_N[k], env.action_N[k]
        s0, c0 = state0[idx].cpu().numpy(), soc0[idx].cpu().numpy()
        model = env.tasks[k].simulator.model
        series = np.asarray(task["series"], dtype=float)
        nd = model.N_des
        for j, e in enumerate(sample):
            orc = O.OracleEnv(task["network"], delta_t=task["delta_t"], gamma=0.995, lamb=100, costs_clipping=(1, 100),
                              aux_bounds=((0, series.shape[1] - 1),), tables=series, sparse=False, tol=1e-6)
            # the initial state the device drew for this environment: its first draw that converged
            first = rng.series_init_state(model, series, SEED, int(e), int(rc0[e]) - 1)
            o, conv = orc.reset_to(first)
            assert conv
            npt.assert_allclose(s0[j][:S], orc.state, rtol=0, atol=1e-9)
            npt.assert_allclose(c0[j][:nd], orc.soc, rtol=0, atol=1e-12)
            for t in range(T):
                if orc.terminated:  # next-step autoreset: this call returns the first observation of a new episode
                    s_init = rng.series_init_state(model, series, SEED, int(e), int(R["rc"][t][j]))
                    o, conv = orc.reset_to(s_init)
                    assert conv == (not R["term"][t][j])
                    assert R["r"][t][j] == 0.0 and R["el"][t][j] == 0.0 and R["pen"][t][j] == 0.0
                    n_reset += 1
                    if not conv:       # a redraw that does not converge stays terminal: the next call draws again
                        orc.terminated = True
                        assert not R["obs"][t][j][:S].any()
                        continue
                else:
                    o, r, term = orc.step(R["a"][t][j][:NA])
                    assert term == bool(R["term"][t][j]), (k, e, t)
                    npt.assert_allclose(R["r"][t][j], r, rtol=1e-9, atol=1e-12)
                    if term:
                        n_term += 1
                        assert not R["obs"][t][j][:S].any()
                        continue
                    npt.assert_allclose(R["el"][t][j], orc.e_loss, rtol=1e-9, atol=1e-12)
                    npt.assert_allclose(R["pen"][t][j], orc.penalty, rtol=1e-9, atol=1e-10)
                npt.assert_allclose(R["obs"][t][j][:S], o, rtol=0, atol=1e-9, err_msg="task %d env %d step %d" % (k, e, t))
                assert int(R["it"][t][j]) == orc.last["n_iter"], (k, e, t)
            n_checked += 1
    assert n_checked >= 40 and n_term > 10 and n_reset > 10


def test_mixed_batch_equals_each_network_on_its_own():
    """the same environments stepped through a mixed batch and through one BatchedANMEnv-family object per network (same
    seeds and RNG keys through env_offset = 0 and the same environment indices is not possible -- so: same initial states
    handed over, same actions): state, observation, reward, flags equal bit for bit where the kernel family is the same"""
    from gym_anm_amd.envs import ANM6EasyVec

    tasks = make_tasks()[:1] + make_tasks()[3:]   # ANM6Easy + the 30-bus feeder
    E_ = 4096
    env_task = np.arange(E_) % 2
    env = MixedBatchedANMEnv(tasks, env_task, device=DEV, seed=9, tol=1e-6, autoreset=False, streams=False)
    env.check_actions = False
    env.reset(seed=9)
    rows = env.task_rows(0)
    own = ANM6EasyVec(num_envs=int(rows.numel()), device=DEV, seed=9, tol=1e-6, handoff_after=6)
    own.reset(options={"init_state": env.state[rows][:, : env.state_N[0]].clone()})
    own.check_actions = False
    npt.assert_array_equal(own.state.cpu().numpy(), env.state[rows][:, : env.state_N[0]].cpu().numpy())
    gen = torch.Generator(device=DEV).manual_seed(3)
    lo, hi = torch.as_tensor(env.action_space.low, device=DEV), torch.as_tensor(env.action_space.high, device=DEV)
    for t in range(12):
        a = lo + (hi - lo) * torch.rand((E_, env.A), generator=gen, dtype=torch.float64, device=DEV)
        obs, r, term, _, _ = env.step(a)
        o2, r2, t2, _, _ = own.step(a[rows][:, :6].contiguous())
        assert torch.equal(obs[rows][:, :18], o2) and torch.equal(r[rows], r2) and torch.equal(term[rows], t2), t
        assert torch.equal(env.nr_iters[rows], own.simulator.nr_iters)


def test_view_step_variants_are_bit_identical(monkeypatch):
    """k_step_view exists budgeted for one and for two wavefronts per SIMD (the launch picks by batch size; ANM_VIEW_WAVES, read
    when the model is created, forces one): the same arithmetic -- states, observations, rewards, flags equal bit for bit
    over 30 autoresetting steps of two ANM6 models in one batch, terminations and diverging solves included"""
    import copy

    net_b = copy.deepcopy(networks.anm6_network())
    net_b["branch"][:, 5] *= 1.1
    tasks = [dict(network=networks.anm6_network(), series=anm6easy_series(), costs_clipping=(1, 100)),
             dict(network=net_b, series=anm6easy_series(), costs_clipping=(1, 100))]
    E_ = 8192
    envs = []
    for v in ("1", "2"):
        monkeypatch.setenv("ANM_VIEW_WAVES", v)
        envs.append(MixedBatchedANMEnv(tasks, np.arange(E_) % 2, device=DEV, seed=4, tol=1e-6, autoreset=True))
    monkeypatch.delenv("ANM_VIEW_WAVES")
    lo, hi = envs[0]._act_low, envs[0]._act_high
    gen = torch.Generator(device=DEV).manual_seed(5)
    for env in envs:
        env.check_actions = False
        env.reset(seed=4)
    n_term = 0
    for t in range(30):
        a = lo + (hi - lo) * torch.rand(lo.shape, generator=gen, dtype=torch.float64, device=DEV)
        outs = [env.step(a) for env in envs]
        n_term += int(outs[0][2].sum())
        for x, y in zip(outs[0][:3], outs[1][:3]):
            assert torch.equal(x, y), t
        assert torch.equal(envs[0].state, envs[1].state) and torch.equal(envs[0].nr_iters, envs[1].nr_iters)
    assert n_term > 20
