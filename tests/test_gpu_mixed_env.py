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


def test_mixed_step_launches_only_from_the_first_call(monkeypatch):
    """`step()` never synchronises and never allocates -- the FIRST step after reset() is captured into a HIP
    graph (a synchronize inside it would fail the capture), replayed, and equals the eager step bit for bit; the stream plan
    is made by reset() (`tune_streams`: timed launches on a snapshot of the batch, restored afterwards)."""
    tasks = make_tasks()
    E_ = 8192
    env_task = np.random.default_rng(11).integers(0, len(tasks), E_)
    kw = dict(device=DEV, seed=21, tol=1e-6, autoreset=True)
    env, twin = MixedBatchedANMEnv(tasks, env_task, **kw), MixedBatchedANMEnv(tasks, env_task, streams=False, **kw)
    assert env.launch_us is None and not env.tuned and twin.tuned and twin.launch_us is None
    for e in (env, twin):
        e.check_actions = False
        e.reset(seed=21)
    # tuning ran inside reset() and left the batch exactly as the untuned twin's
    assert env.tuned and sorted(env.launch_us) == [0, 1, 2, 3] and all(v > 0 for v in env.launch_us.values())
    assert sorted(k for slot in env._slots for k in slot) == [0, 1, 2, 3]
    for a, b in zip(env._mutable(), twin._mutable()):
        assert torch.equal(a, b)
    lo, hi = env._act_low, env._act_high
    gen = torch.Generator(device=DEV).manual_seed(2)
    acts = [lo + (hi - lo) * torch.rand(lo.shape, generator=gen, dtype=torch.float64, device=DEV) for _ in range(6)]
    buf = acts[0].clone()
    torch.cuda.synchronize()
    # capture the very first step (no warm-up step before it)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env.step(buf)
    torch.cuda.synchronize()
    before = None
    for t, a in enumerate(acts):
        buf.copy_(a)
        g.replay()
        o2, r2, t2, _, _ = twin.step(a)
        assert torch.equal(env._obs, o2) and torch.equal(env.reward, r2) and torch.equal(env.terminated, t2), t
        assert torch.equal(env.state, twin.state) and torch.equal(env._reset_count, twin._reset_count)
    # tuning under capture is refused with a message (it synchronises)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    with pytest.raises(RuntimeError, match="before capturing"):
        env.tune_streams()
    monkeypatch.undo()
    # eager steps: the allocator's counters do not move
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats(DEV)
    for a in acts:
        env.step(a)
    torch.cuda.synchronize()
    after = torch.cuda.memory_stats(DEV)
    for key in ("allocation.all.allocated", "segment.all.allocated", "allocated_bytes.all.allocated"):
        assert after[key] == before[key], key


def test_action_padding_is_ignored_by_the_box_check():
    """the columns beyond an environment's own action width are documented as ignored: a uniform sample over the
    widest network's columns (non-zero, even NaN, in the padding) must pass the Box check; a value outside the Box in an
    environment's OWN columns must not."""
    tasks = make_tasks()
    E_ = 512
    env = MixedBatchedANMEnv(tasks, np.arange(E_) % 4, device=DEV, seed=1, tol=1e-6, autoreset=True)
    env.reset(seed=1)
    a = torch.zeros((E_, env.A), dtype=torch.float64, device=DEV)
    for k in range(4):
        a[env.task_rows(k), env.action_N[k]:] = float("nan") if k % 2 else 123.0
    env.step(a)
    bad = a.clone()
    bad[env.task_rows(0)[0], 0] = 1e9
    with pytest.raises(AssertionError):
        env.step(bad)


def _list_obs_from_oracle(orc, model, spec, obs_space):
    """the reference's list-form observation (anm_env.py:562-592, units of simulator.py:559-616) from an OracleEnv's last
    transition: [(key, ids, unit)] -> vector, clipped to the Box"""
    out, b = orc.last, model.baseMVA
    V, I = out["V"], out["I"]
    ifr = out["br_i_from"]
    pu = {"bus_p": out["bus_p"], "bus_q": out["bus_q"], "bus_v_magn": np.abs(V), "bus_v_ang": np.angle(V), "bus_i_magn": np.abs(I),
          "dev_p": out["dev_p"], "dev_q": out["dev_q"], "des_soc": orc.soc, "gen_p_max": out["p_pot"],
          "branch_p": out["br_p_from"], "branch_q": out["br_q_from"], "branch_s": out["br_s"], "branch_i_magn": np.sign(ifr).real * np.abs(ifr)}
    scale = {"pu": 1.0, "rad": 1.0, "MW": b, "MVAr": b, "MVA": b, "MWh": b}
    ids = {"bus": list(model.bus_ids), "dev": list(model.dev_ids), "des": [model.dev_ids[k] for k in model.des_idx],
           "gen": [model.dev_ids[k] for k in model.gen_idx], "branch": [tuple(x) for x in model.branch_ids]}
    vals = []
    for key, nodes, unit in spec:
        if key == "aux":
            vals += [orc.state[len(orc.state) - orc.K + n] for n in nodes]
            continue
        pool = ids["des" if key == "des_soc" else ("gen" if key == "gen_p_max" else key.split("_")[0])]
        for n in nodes:
            vals.append(pu[key][pool.index(tuple(n) if key.startswith("branch") else n)] * scale[unit])
    return np.clip(np.array(vals, dtype=float), obs_space.low, obs_space.high)


def test_list_observations_and_hook_tasks_in_one_batch():
    """the mixed batch takes what the reference's per-network ANMEnv takes (anm_env.py:172-191, 497-521) --
    a list-form observation per task, gathered inside the task's step kernel through its view (ANM6 on the tree kernel,
    the 30-bus feeder on its tree kernel: wider than its state row), and hook tasks (init_state / next_vars, K = 2 through
    the general lane-group kernel, K = 1 on the thread-per-environment kernels) -- every task against its own OracleEnv:
    observation <= 1e-8, reward rtol 1e-9, terminated exact, the padding of every row untouched."""
    nets = [networks.anm6_network(), networks.synthetic_radial_network(30, 0), networks.three_bus_loop_network(gen_max=1.5),
            networks.anm6_network()]
    tab6 = anm6easy_series()
    tab30 = daily_series(nets[1], 96, 3)
    L0 = [("bus_v_magn", "all", "pu"), ("branch_s", "all", "MVA"), ("dev_p", [2, 4], "MW"), ("aux", "all")]
    L1 = [("bus_v_magn", "all", "pu"), ("branch_p", "all", "MW"), ("des_soc", "all", "MWh"), ("aux", "all")]

    def next3_np(s):      # 3-bus loop, K = 2: aux0 a 24-step clock, aux1 a counter that modulates the generation potential
        t = int((s[-2] + 1) % 24)
        w = 0.5 + 0.5 * np.cos(0.4 * (s[-1] + 1.0))
        return np.array([-0.3 - 0.2 * np.sin(2 * np.pi * t / 24), 1.2 * w, 0.9, t, s[-1] + 1.0])

    def next3(s):
        t = torch.remainder(s[:, -2] + 1, 24)
        w = 0.5 + 0.5 * torch.cos(0.4 * (s[:, -1] + 1.0))
        return torch.stack((-0.3 - 0.2 * torch.sin(2 * np.pi * t / 24), 1.2 * w, torch.full_like(t, 0.9), t, s[:, -1] + 1.0), 1)

    def next6_np(s):      # ANM6, K = 1: the ANM6Easy tables read two steps at a time
        t = int((s[-1] + 2) % 96)
        return np.concatenate((tab6[:, t], [t]))

    def next6(s):
        t = torch.remainder(s[:, -1] + 2, 96).long()
        return torch.cat((torch.as_tensor(tab6, device=s.device)[:, t].T, t[:, None].double()), 1)

    inits = {2: [], 3: []}

    def init3(n):         # state: dev_p (5), dev_q (5), soc, gen_p_max (2), aux (2)
        r = np.random.default_rng(100 + len(inits[2]))
        s = np.zeros((n, 15))
        s[:, 1], s[:, 2], s[:, 3] = -r.uniform(0.1, 0.3, n), r.uniform(0.1, 0.3, n), r.uniform(0.1, 0.3, n)
        s[:, 10] = r.uniform(10, 90, n)
        s[:, 11], s[:, 12] = 1.2, 0.9
        s[:, 13], s[:, 14] = r.integers(0, 24, n), r.integers(0, 9, n)
        inits[2].append(s)
        return s

    def init6(n):
        r = np.random.default_rng(200 + len(inits[3]))
        t0 = r.integers(0, 96, n)
        s = np.zeros((n, 18))
        s[:, [1, 3, 5]], s[:, [2, 4]], s[:, 15:17] = tab6[:3, t0].T, tab6[3:, t0].T, tab6[3:, t0].T
        s[:, 14], s[:, 17] = r.uniform(5, 95, n), t0
        inits[3].append(s)
        return s

    tasks = [dict(network=nets[0], series=tab6, observation=L0, costs_clipping=(1, 100)),
             dict(network=nets[1], series=tab30, observation=L1, costs_clipping=(1, 100)),
             dict(network=nets[2], init_state=init3, next_vars=next3, K=2, delta_t=0.5, aux_bounds=np.array([[0, 23], [0, 1e6]]), costs_clipping=(1, 100)),
             dict(network=nets[3], init_state=init6, next_vars=next6, K=1, aux_bounds=np.array([[0, 95]]), costs_clipping=(1, 100))]
    E_, T, SEED = 2048, 16, 77
    env_task = np.random.default_rng(8).integers(0, 4, E_)
    env = MixedBatchedANMEnv(tasks, env_task, device=DEV, seed=SEED, tol=1e-6, autoreset=True)
    assert env.impls == ["radial", "radial", "mesh", "thread"], env.impls
    assert env.obs_N == [14, 65, 15, 18] and env.W == 65
    env.check_actions = False
    env.state.fill_(-777.0)
    env._obs.fill_(-777.0)
    obs0, _ = env.reset(seed=SEED)
    assert not bool(env.terminated.any())
    assert len(inits[2]) == 1 and len(inits[3]) == 1    # (the mild initial states all converged: no redraw)
    state0, rc0 = env.state.clone(), env._reset_count.clone()
    lo, hi = env._act_low, env._act_high
    gen = torch.Generator(device=DEV).manual_seed(6)
    rec = {k: [] for k in ("a", "obs", "r", "term", "rc")}
    rec["obs"].append(obs0.clone())
    for t in range(T):
        a = lo + (hi - lo) * torch.rand(lo.shape, generator=gen, dtype=torch.float64, device=DEV)
        if t % 3:      # a gentler policy two steps in three: longer episodes
            a = a * 0.3
        rec["rc"].append(env._reset_count.clone())
        obs, r, term, _, _ = env.step(a)
        for k, v in zip(("a", "obs", "r", "term"), (a, obs, r, term)):
            rec[k].append(v.clone())
    for k in range(4):   # the padding of every row
        rows = env.task_rows(k)
        assert bool((env.state[rows][:, env.state_N[k]:] == -777.0).all()) and bool((env._obs[rows][:, env.obs_N[k]:] == -777.0).all()), k
    specs = {0: env.tasks[0].obs_values, 1: env.tasks[1].obs_values}
    hooks = {2: next3_np, 3: next6_np}
    n_checked = n_term = 0
    for k, task in enumerate(tasks):
        rows = env.task_rows(k).cpu().numpy()
        sample = np.random.default_rng(k).choice(rows, 8, replace=False)
        model = env.tasks[k].simulator.model
        S, NA, NO = env.state_N[k], env.action_N[k], env.obs_N[k]
        space = env.tasks[k].observation_space
        pos = {int(e): j for j, e in enumerate(rows)}
        for e in sample:
            e = int(e)
            if k < 2:
                series = tab6 if k == 0 else tab30
                orc = O.OracleEnv(task["network"], delta_t=0.25, costs_clipping=(1, 100), aux_bounds=((0, 95),), tables=series, sparse=False, tol=1e-6)
                o, conv = orc.reset_to(rng.series_init_state(model, series, SEED, e, int(rc0[e]) - 1))
                want = lambda: _list_obs_from_oracle(orc, model, specs[k], space)
            else:
                ab = ((0, 23), (0, 1e6)) if k == 2 else ((0, 95),)
                orc = O.OracleEnv(task["network"], delta_t=task.get("delta_t", 0.25), costs_clipping=(1, 100), aux_bounds=ab, sparse=False,
                                  tol=1e-6, next_vars=hooks[k])
                o, conv = orc.reset_to(inits[k][0][pos[e]])
                want = lambda: np.clip(orc.state, orc.obs_low, orc.obs_high)
            assert conv
            npt.assert_allclose(state0[e].cpu().numpy()[:S], orc.state, rtol=0, atol=1e-9)
            npt.assert_allclose(rec["obs"][0][e].cpu().numpy()[:NO], want(), rtol=0, atol=1e-8, err_msg="task %d env %d reset" % (k, e))
            for t in range(T):
                got_o, got_r, got_t = rec["obs"][t + 1][e].cpu().numpy()[:NO], float(rec["r"][t][e]), bool(rec["term"][t][e])
                if orc.terminated:
                    if k >= 2:        # hook tasks: absorbing until the caller resets them
                        assert got_t and got_r == 0.0 and not got_o.any()
                        continue
                    _, conv = orc.reset_to(rng.series_init_state(model, series, SEED, e, int(rec["rc"][t][e])))
                    assert conv == (not got_t) and got_r == 0.0
                    if not conv:
                        orc.terminated = True
                        assert not got_o.any()
                        continue
                else:
                    _, r_, term_ = orc.step(rec["a"][t][e].cpu().numpy()[:NA])
                    assert term_ == got_t, (k, e, t)
                    npt.assert_allclose(got_r, r_, rtol=1e-9, atol=1e-12)
                    if term_:
                        n_term += 1
                        assert not got_o.any()
                        continue
                npt.assert_allclose(got_o, want(), rtol=0, atol=1e-8, err_msg="task %d env %d step %d" % (k, e, t))
            n_checked += 1
    assert n_checked == 32 and n_term >= 1, (n_checked, n_term)
