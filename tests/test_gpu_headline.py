"""GPU tier, round 2: parity exactly where the headline number is taken, and on the entry points the
round-1 suite only reached indirectly.

* the benchmarked configuration (65 536 autoresetting ANM6Easy environments, Newton tolerance 1e-6,
  reference iteration cap) replayed by the oracle for hundreds of environments, autoreset draws included,
  with exact Newton iteration counts;
* the reference's ``Simulator.reset`` golden vectors (``tests/golden/reset_*.npz``, 350 cases incl.
  non-converged ones) through ``anm_reset_f64`` and ``BatchedSimulator.reset`` on both kernel families
  (simulator.py:225-293, anm_env.py:266-311);
* BASELINE.json config 5's partition rule on one GPU: two half batches keyed by ``env_offset`` are
  bit-identical to the unsharded batch;
* a callable observation and a K = 2 task with a host ``next_vars`` hook against the oracle
  (anm_env.py:176-191, 511-514);
* one captured step of the two-launch mode replayed from a HIP graph is bit-identical to eager steps.
"""
import os

import numpy as np
import numpy.testing as npt
import pytest
import torch

import parity_common as pc
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


KW = lambda net: {"device": DEV}
_uniform_actions = pc.uniform_actions


@pytest.mark.parametrize("handoff", ["auto", 0, 2])
def test_headline_config_oracle_replay(handoff):
    """bench.py's configuration (65 536 environments, tol 1e-6, autoreset, reference cap) for 10 steps; 256+
    environments replayed by the oracle.  handoff=0 forces every solve through the lane-group continuation
    from its first iteration, 2 most of them; "auto" is the shipped policy."""
    n, n_term, n_reset = pc.headline_replay(KW, 65536, 10, 224, 64, handoff_after=handoff)
    assert n >= 256 and n_term >= 32 and n_reset >= 32


@pytest.mark.parametrize("E_", [524288, 1048576])
def test_config5_single_gpu_point_oracle_replay(E_):
    """BASELINE.json config 5's load when its 524 288 environments sit on ONE GPU (and twice that): the step the
    AUTOMATIC straggler policy chooses at this size (`straggler_after="auto"`: the solves still running after 6
    iterations of the whole batch continue packed on lane groups; from 1 M environments on in two levels), 10
    autoresetting steps: 128 sampled environments plus 64 that collapsed early replayed by OracleEnv(tol=1e-6) --
    observation <= 1e-9, reward rtol 1e-9, terminated and Newton iteration counts exact, autoreset draws included --
    and every output of every step bit-identical to the one-launch step with the in-wave hand-over (the path
    test_headline_config_oracle_replay pins at 65 536).  Reference loop: gym_anm/envs/anm_env.py:333-453."""
    def policy(env):
        assert env._ws is not None and env._ws.iter_cap == 6, "the automatic policy must pick the packed continuation here"
        assert (env._ws.mid_cap > 0) == (E_ >= 1000000)

    n, n_term, n_reset = pc.headline_replay(KW, E_, 10, 128, 64, twin_kw=dict(straggler_after=None), check_env=policy)
    assert n >= 160 and n_term >= 32 and n_reset >= 32


def test_step_is_one_launch_without_allocation():
    """`step()` on the fast path launches the step kernel and nothing else: no tensor is allocated (the caching
    allocator's counters do not move) and `pfe_converged` is formed only when somebody reads it."""
    from gym_anm_amd.envs import ANM6EasyVec

    env = ANM6EasyVec(num_envs=65536, device=DEV, seed=3, tol=1e-6, autoreset=True)
    env.check_actions = False
    env.reset(seed=3)
    gen = torch.Generator(device=DEV).manual_seed(1)
    acts = [_uniform_actions(env, gen) for _ in range(4)]
    env.step(acts[0])
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats(DEV)
    for a in acts:
        obs, r, term, trunc, info = env.step(a)
    torch.cuda.synchronize()
    after = torch.cuda.memory_stats(DEV)
    for key in ("allocation.all.allocated", "segment.all.allocated", "allocated_bytes.all.allocated"):
        assert after[key] == before[key], key
    assert torch.equal(env.pfe_converged, ~term)  # formed now, from the flags of the last step


@pytest.mark.parametrize("name,impl", [("anm6", "thread"), ("anm6", "radial"), ("anm6", "mesh"), ("3bus", "thread"), ("3bus", "mesh")])
def test_reset_golden_simulator_and_env(name, impl):
    env = pc.reset_golden(name, KW, impl)
    assert env.simulator.backend.path.endswith(".so") and "gym_anm_amd/_build/libanm_" in env.simulator.backend.path


def test_sharded_batches_equal_the_unsharded_batch():
    n_term, max_resets = pc.sharded_equal_unsharded(KW, 16384, 40)
    assert n_term > 50 and max_resets >= 2


def test_two_devices_shard_like_one():
    """The same partition rule across two real devices (skipped on a single-GPU box)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    from gym_anm_amd.envs import ANM6EasyVec

    E_ = 8192
    H = E_ // 2
    whole = ANM6EasyVec(num_envs=E_, device="cuda:0", seed=7, tol=1e-6, autoreset=True)
    parts = [ANM6EasyVec(num_envs=H, device="cuda:%d" % k, seed=7, tol=1e-6, autoreset=True, env_offset=k * H)
             for k in range(2)]  # fmt: skip
    for env in [whole] + parts:
        env.check_actions = False
        env.reset(seed=7, options={"sampler": "device"})
    gen = torch.Generator(device=DEV).manual_seed(8)
    for t in range(20):
        a = _uniform_actions(whole, gen)
        o, r, term, _, _ = whole.step(a)
        for k, env in enumerate(parts):
            ok, rk, tk, _, _ = env.step(a[k * H : (k + 1) * H].to(env.device))
            assert torch.equal(ok.cpu(), o[k * H : (k + 1) * H].cpu()) and torch.equal(rk.cpu(), r[k * H : (k + 1) * H].cpu())
            assert torch.equal(tk.cpu(), term[k * H : (k + 1) * H].cpu())


@pytest.mark.parametrize("impl", ["thread", "radial", "mesh"])
def test_callable_observation_and_two_aux_variables(impl):
    pc.callable_observation_and_two_aux(lambda net: dict(KW(net), impl=impl))


def test_reference_custom_obs_space_known_answers():
    """tests/envs/custom_obs_space.py of the reference, on the HIP path (the 3-bus chain is served by the general
    lane-group kernel or a compiled library, whichever the loader picks)."""
    pc.reference_custom_obs_space(KW)


def test_reference_example_simple_env_on_the_gpu():
    """examples/simple_env.py of the reference through `gym_anm_amd.ANMEnv` (NumPy-facing, one environment) on the HIP path."""
    env = pc.reference_example_simple_env(KW)
    assert env.simulator.backend.device_type == "cuda"


def test_single_environment_class_with_callable_observation_on_the_gpu():
    env = pc.single_env_with_callable_observation(KW)
    assert env.simulator.backend.device_type == "cuda"


def test_next_vars_of_the_wrong_size_is_refused():
    from gym_anm_amd import errors, networks
    from gym_anm_amd.envs.anm_env import BatchedANMEnv

    class Bad(BatchedANMEnv):
        def __init__(self, **kw):
            super().__init__(networks.anm6_network(), "state", 1, 0.25, 0.995, 100, **kw)

        def next_vars(self, s_t):
            return torch.zeros((self.num_envs, 5), dtype=torch.float64, device=s_t.device)

    env = Bad(num_envs=4, device=DEV)
    env.reset(options={"init_state": np.zeros(env.state_N)})
    with pytest.raises(errors.EnvNextVarsError):
        env.step(torch.zeros((4, 6), dtype=torch.float64, device=DEV))


@pytest.mark.parametrize("n_capture", [1, 3])
def test_captured_two_launch_step_replays_bit_identically(n_capture):
    """One (or an odd number of) two-launch steps captured in a HIP graph and replayed: the record
    counters are handled entirely on the device, so every replay equals the eager step bit for bit."""
    from gym_anm_amd.envs import ANM6EasyVec

    E_ = 65536
    envs = [ANM6EasyVec(num_envs=E_, device=DEV, seed=9, autoreset=True, tol=1e-6, straggler_after=4, handoff_after=-1)
            for _ in range(2)]  # fmt: skip
    for env in envs:
        assert env._ws is not None
        env.check_actions = False
        env.reset(seed=9)
    eager, graphed = envs
    gen = torch.Generator(device=DEV).manual_seed(6)
    static_a = _uniform_actions(graphed, gen)
    stream = torch.cuda.Stream(device=DEV)
    stream.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(stream):
        for _ in range(2):  # warm-up on the side stream, mirrored on the eager environment
            graphed.step(static_a)
    torch.cuda.current_stream(DEV).wait_stream(stream)
    for _ in range(2):
        eager.step(static_a)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        for _ in range(n_capture):
            graphed.step(static_a)
    # capture does not execute: the eager environment is not advanced here
    n_term = 0
    for rep in range(6):
        a = _uniform_actions(eager, gen)
        static_a.copy_(a)
        graph.replay()
        for _ in range(n_capture):
            o, r, term, _, _ = eager.step(a)
        torch.cuda.synchronize()
        n_term += int(term.sum())
        for x, y in ((graphed._state_obs, o), (graphed.reward, r), (graphed.terminated, term), (graphed.state, eager.state),
                     (graphed.simulator.soc, eager.simulator.soc), (graphed.simulator.nr_iters, eager.simulator.nr_iters),
                     (graphed._reset_count, eager._reset_count), (graphed.timestep, eager.timestep)):  # fmt: skip
            assert torch.equal(x, y), rep
    assert n_term > 100


@pytest.mark.parametrize("fuse", [True, False])
def test_all_state_variables_observation(fuse):
    """Every STATE_VARIABLE in every unit, gathered inside the step kernel (fuse=True, no dump, one launch)
    and by the separate gather kernel from the dump: both against the golden transitions."""
    env = pc.all_state_variables_observation(KW, fuse_observation=fuse)
    assert env._obs_fused == fuse and env.observation_N == 2 * (6 * 6 + 2 * 7 + 1 + 2 + 4 * 5) + 5 + 1


@pytest.mark.parametrize("impl", ["radial", "mesh"])
@pytest.mark.parametrize("fuse", [True, False])
def test_all_state_variables_observation_lane_group_families(impl, fuse):
    """The same list observation through the lane-group kernels: gathered INSIDE the step kernel from the lanes that hold
    the quantities (round 5: one LDS row per environment, no dump, no second launch) and, fuse=False, by the gather kernel
    from the dump -- both against the golden transitions, and bit-equal to each other."""
    env = pc.all_state_variables_observation(lambda net: dict(KW(net), impl=impl), fuse_observation=fuse)
    assert env.simulator.impl == impl and env._obs_fused == fuse and env._need_full == (not fuse)


@pytest.mark.parametrize("impl", ["radial", "mesh"])
def test_fused_list_observation_of_a_30_bus_feeder_equals_the_gather_kernel(impl):
    """config 4's network with a custom observation (anm_env.py:497-521, 562-592): 2 048 environments, host next_vars, 12 steps
    with collapses; the fused gather of the lane-group kernel == dump + anm_gather_obs_f64, bit for bit, terminal rows zero."""
    from gym_anm_amd import networks
    from gym_anm_amd.envs.anm_env import BatchedANMEnv

    net = networks.synthetic_radial_network(30, 0)
    spec = [("bus_v_magn", "all", "pu"), ("bus_v_ang", [3, 7, 29], "degree"), ("branch_s", "all", "MVA"), ("dev_p", "all", "MW"),
            ("dev_q", [1, 5], "MVAr"), ("des_soc", "all", "MWh"), ("gen_p_max", "all", "MW"), ("bus_i_magn", [0, 12], "pu"),
            ("branch_i_ang", [(0, 1)], "rad"), ("aux", "all")]  # fmt: skip
    E_ = 2048

    class Task(BatchedANMEnv):
        def __init__(self, fuse):
            super().__init__(net, spec, 2, 0.25, 0.995, 100, aux_bounds=np.array([[0, 50], [0, 1]]), num_envs=E_, device=DEV, impl=impl,
                             fuse_observation=fuse, seed=5)

        def next_vars(self, s):
            return self._vars

    a, b = Task(True), Task(False)
    assert a._obs_fused and not b._obs_fused and a.simulator.impl == impl
    m, base = a.simulator.model, a.simulator.baseMVA
    g = torch.Generator(device="cpu").manual_seed(11)
    U = lambda lo, hi: torch.as_tensor(lo) + (torch.as_tensor(hi) - torch.as_tensor(lo)) * torch.rand((E_, len(lo)), generator=g, dtype=torch.float64)
    s0 = np.zeros((E_, a.state_N))
    for env in (a, b):
        env.check_actions = False
        env.reset(options={"init_state": torch.as_tensor(s0)})
    n_term = 0
    for t in range(12):
        scale = 1.0 + 0.35 * t   # heavier and heavier loads: collapses on the way
        pl = U(m.dev_p_min[m.load_idx] * base * scale, 0 * m.dev_p_min[m.load_idx])
        pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * base)
        aux = torch.stack((torch.full((E_,), float(t)), torch.rand(E_, generator=g, dtype=torch.float64)), 1)
        v = torch.cat((pl, pp, aux), 1).to(DEV)
        act = U(a.action_space.low, a.action_space.high).to(DEV)
        outs = []
        for env in (a, b):
            env._vars = v
            o, r, term, _, _ = env.step(act)
            outs.append((o.clone(), r.clone(), term.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]), t
        assert not bool(outs[0][0][outs[0][2]].any())
        n_term = int(outs[0][2].sum())
    assert 0 < n_term < E_


def test_fused_list_observation_equals_the_gather_kernel_with_autoreset():
    """Series-mode task with a list observation and in-kernel autoreset: the fused epilogue (general step
    kernel) and the dump + gather kernel agree bit for bit, resets and terminal rows included, and the
    "state" fast path steps the same environments identically."""
    from gym_anm_amd.envs.anm6 import ANM6Vec, anm6easy_series
    from gym_anm_amd.envs import ANM6EasyVec

    spec = [("bus_v_magn", "all", "pu"), ("branch_s", [(1, 2), (2, 4)], "MVA"), ("des_soc", "all"), ("dev_p", [0, 6], "MW"),
            ("bus_v_ang", [3], "degree"), ("branch_i_magn", "all", "pu"), ("aux", [0])]  # fmt: skip
    E_ = 8192
    mk = lambda fuse: ANM6Vec(spec, 1, 0.25, 0.995, 100, aux_bounds=np.array([[0, 95]]), costs_clipping=(1, 100), num_envs=E_,
                              device=DEV, series=anm6easy_series(), seed=3, autoreset=True, tol=1e-6, fuse_observation=fuse)
    fused, plain = mk(True), mk(False)
    ref = ANM6EasyVec(num_envs=E_, device=DEV, seed=3, autoreset=True, tol=1e-6)
    assert fused._obs_fused and not plain._obs_fused and not fused._need_full and plain._need_full
    for env in (fused, plain, ref):
        env.check_actions = False
        env.reset(seed=3, options={"sampler": "device"})
    gen = torch.Generator(device=DEV).manual_seed(1)
    n_term = 0
    for t in range(30):
        a = _uniform_actions(ref, gen)
        o1, r1, t1, _, _ = fused.step(a)
        o2, r2, t2, _, _ = plain.step(a)
        o3, r3, t3, _, _ = ref.step(a)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(t1, t2), t
        assert torch.equal(r1, r3) and torch.equal(t1, t3) and torch.equal(fused.state, ref.state), t
        assert torch.equal(fused._reset_count, ref._reset_count) and torch.equal(fused.timestep, ref.timestep)
        assert not bool(o1[t1].any())
        n_term += int(t1.sum())
    assert n_term > 20 and int(fused._reset_count.sum()) > 20


@pytest.mark.parametrize("impl", ["thread", "radial", "mesh"])
def test_track_full_follows_the_steps(impl):
    """track_full=True: simulator.state / pfe_converged are refreshed by every step (the reference's
    Simulator does that on every transition, simulator.py:529-537)."""
    from gym_anm_amd.envs import ANM6EasyVec

    E_ = 512
    env = ANM6EasyVec(num_envs=E_, device=DEV, seed=2, track_full=True, impl=impl)
    plain = ANM6EasyVec(num_envs=E_, device=DEV, seed=2, impl=impl)
    for e_ in (env, plain):
        e_.check_actions = False
        e_.reset(seed=2)
    assert env.simulator.state is not None
    gen = torch.Generator(device=DEV).manual_seed(3)
    for t in range(5):
        a = _uniform_actions(env, gen)
        o, r, term, _, _ = env.step(a)
        o2, r2, term2, _, _ = plain.step(a)
        assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(term, term2)
        live = ~term
        st = env.simulator.state
        npt.assert_allclose(st.tensor("dev_p", "MW")[live].cpu().numpy(), env.state[live, :7].cpu().numpy(), rtol=0, atol=1e-9)
        assert torch.equal(env.simulator.pfe_converged, ~term) and torch.equal(env.pfe_converged, ~term)
        assert torch.equal(plain.pfe_converged, ~term)


@pytest.mark.parametrize("impl", ["thread", "radial", "mesh"])
def test_heterogeneous_networks_64_parameter_classes(impl):
    """64 distinct networks in one batch of 4096 environments, every kernel family, against the oracle."""
    sim = pc.heterogeneous_networks(KW, impl=impl)
    assert sim.impl == impl and len(sim.variant_models) == 64


def test_parameter_classes_inside_a_block_of_64_move_the_model_to_a_lane_group_family():
    """One class per aligned block of 64 environments: every family (scalar constants per wavefront).  A class change
    inside a block: only the lane-group families can (test_a_different_network_in_every_environment); a model on the
    thread-per-environment family moves there, and cannot be moved back."""
    from gym_anm_amd import errors, networks
    from gym_anm_amd.simulator import BatchedSimulator

    base = networks.anm6_network()
    ev = np.zeros(128, dtype=np.int32)
    ev[70:] = 1  # class changes in the middle of a 64-environment block
    sim = BatchedSimulator(base, 0.25, 100, num_envs=128, device=DEV, variants=[networks.perturbed_network(base, 1)], env_variant=ev)
    assert sim.impl == "radial"
    rc = sim.backend.lib.anm_model_set_impl(sim._handle, 0)
    assert rc != 0 and b"lane-group" in sim.backend.lib.anm_last_error()
    aligned = BatchedSimulator(base, 0.25, 100, num_envs=128, device=DEV, variants=[networks.perturbed_network(base, 1)],
                               env_variant=np.repeat([0, 1], 64))
    assert aligned.impl == "thread"
    with pytest.raises(errors.UnsupportedNetworkError):
        BatchedSimulator(base, 0.25, 100, num_envs=128, device=DEV, variants=[networks.two_bus_network()])


def test_lifting_a_class_binding_is_refused_while_a_list_observation_of_the_other_family_is_set():
    """bind per-environment classes (thread -> lane-group family), set a list-form observation there (identity
    tables, no row stride), unbind: the model would be back on the thread-per-environment kernels with tables they cannot
    read (overlapping LDS rows, silently wrong observations).  Both ways back -- unbind, rebind in aligned blocks -- are
    refused until the list is cleared."""
    import ctypes as C

    from gym_anm_amd import _lib, networks
    from gym_anm_amd.simulator import BatchedSimulator

    base = networks.anm6_network()
    ev = np.zeros(128, dtype=np.int32)
    ev[70:] = 1
    sim = BatchedSimulator(base, 0.25, 100, num_envs=128, device=DEV, variants=[networks.perturbed_network(base, 1)], env_variant=ev)
    lib, h = sim.backend.lib, sim._handle
    assert sim.impl == "radial"
    cfg = _lib.EnvConfig(K=1, gamma=0.995, clip_e_loss=1.0, clip_penalty=100.0, obs_low=None, obs_high=None, series=None, period=0)
    assert lib.anm_model_set_env(h, C.byref(cfg)) == 0
    idx = np.array([2, 3, sim.full_dim], dtype=np.int32)
    one, lo, hi = np.ones(3), -np.ones(3) * 1e9, np.ones(3) * 1e9
    args = (_lib.as_c(idx, np.int32)[1], _lib.as_c(one, np.float64)[1], _lib.as_c(lo, np.float64)[1], _lib.as_c(hi, np.float64)[1])
    assert lib.anm_model_set_obs(h, 3, *args) == 0
    # an aux index beyond the task's K is refused (it would gather LDS nobody wrote)
    bad = np.array([2, 3, sim.full_dim + 1], dtype=np.int32)
    assert lib.anm_model_set_obs(h, 3, _lib.as_c(bad, np.int32)[1], *args[1:]) != 0 and b"out of range" in lib.anm_last_error()
    assert lib.anm_model_set_obs(h, 3, *args) == 0
    assert lib.anm_model_bind_env_classes(h, None, 0) != 0 and b"clear it first" in lib.anm_last_error()
    aligned = torch.as_tensor(np.repeat([0, 1], 64).astype(np.int32), device=DEV)
    assert lib.anm_model_bind_env_classes(h, aligned.data_ptr(), 128) != 0 and b"clear it first" in lib.anm_last_error()
    assert lib.anm_model_get_impl(h) == 1                      # nothing moved
    assert lib.anm_model_set_obs(h, 0, None, None, None, None) == 0
    assert lib.anm_model_bind_env_classes(h, None, 0) == 0 and lib.anm_model_get_impl(h) == 0
    assert lib.anm_model_set_obs(h, 3, *args) == 0             # set again: now with the thread family's tables


# ---------------------------------------------------------------------------------------------------
# the general lane-group family ("mesh"): any topology, block-sparse Jacobian in LDS
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(pc.golden_nets()))
def test_mesh_transition_golden(name):
    """Every golden network -- the tree ones and the 3-bus loop with its ten transformer variants -- through
    the general lane-group kernel: flags and iteration counts exact, electrical state <= 1e-9."""
    sim = pc.check_transition_against_golden(name, pc.golden_nets()[name], DEV, impl="mesh")
    assert sim.impl == "mesh"


@pytest.mark.parametrize("name", ["anm6", "3bus", "case30"])
def test_mesh_transition_golden_f32_jacobian(name):
    """The fp32 Jacobian / factorisation mode of the general lane-group kernel (mismatch, stop test and update stay
    fp64): same flags, electrical state within 2e-6 p.u. of the golden vectors."""
    pc.check_transition_against_golden(name, pc.golden_nets()[name], DEV, precision="f32", atol=2e-6, check_iters=False, impl="mesh")


def test_mesh_anm6easy_episodes():
    from gym_anm_amd.envs import ANM6EasyVec

    env = pc.run_episodes(lambda n: ANM6EasyVec(num_envs=n, device=DEV, impl="mesh"))
    assert env.simulator.impl == "mesh"


@pytest.mark.parametrize("n_bus,seed,n_chords", [(5, 2, 2), (9, 4, 3), (14, 5, 3), (14, 3, 6), (30, 6, 4), (30, 7, 8), (33, 4, 8), (48, 8, 6), (64, 9, 1),
                                                 (64, 10, 24)])
def test_mesh_random_meshed_networks_against_oracle(n_bus, seed, n_chords):
    """Random meshed networks (loops, line charging, a phase-shifting transformer) nobody compiled a library
    for: generic mode of the general lane-group kernel against the oracle, case by case.  (14, 3, 6), (33, 4, 8),
    (64, 10, 24): more branches than lanes in the group -- a lane plays two; the last one also needs more than the
    default 64 KB of LDS per workgroup.)"""
    _mesh_network_against_oracle(n_bus, seed, n_chords, 96, 1.0, 4)


@pytest.mark.parametrize("n_bus,seed,n_chords,group", [(66, 11, 10, 128), (100, 12, 12, 128), (130, 16, 0, 256), (200, 13, 30, 256),
                                                       (300, 14, 40, 512), (513, 15, 40, 512)])
def test_mesh_networks_larger_than_a_wavefront_against_oracle(n_bus, seed, n_chords, group):
    """The reference takes a network of any size (simulator.py:113-181, solve_load_flow.py:123-164).  Above 65 buses
    the environment is a WORKGROUP of 128 ... 512 lanes (k_mesh<.., WG>): barriers instead of wavefront fences, the
    step program read from global memory.  (130, 16, 0): a tree too large for the radial family.)"""
    sim = _mesh_network_against_oracle(n_bus, seed, n_chords, 24, 40.0 / n_bus, 2)
    assert sim.backend.lib.anm_model_lanes_per_env(sim._handle) == group


@pytest.mark.parametrize("shape", ["star", "ring", "path", "complete", "feeders", "ladder"])
def test_mesh_structured_topologies_against_oracle(shape):
    """Shapes the random feeders do not produce (parity_common.structured_network: a hub with 20 leaves, a ring, a path, a
    complete graph, five feeders off the slack, a ladder) through the general lane-group kernel, case by case against
    the oracle: each stresses another corner of the step program (many contributions to one block, chains, a dense
    end that is the whole network, several elimination roots)."""
    net = pc.structured_network(shape)
    _mesh_network_against_oracle(len(net["bus"]), 3, 0, 64, 0.5, 2, net=net)


def _small_tree():
    """the 8-bus feeder of parity_common.general_step_with_wide_rows (its library is compiled by that test anyway): a tree with
    a three-child bus -- no all-DPP plan, so its lane groups use the HYBRID hand-overs (codegen.hybrid_plan) in the tree kernel
    and the ds_bpermute hand-overs with children folded by height in the thread-per-environment kernels"""
    from gym_anm_amd import networks

    net = networks.synthetic_radial_network(8, 3)
    dev = [list(r) for r in net["device"]]
    while len(dev) < 9:
        dev.append([len(dev), 2 + len(dev) % 5, -1, 0.2, 0, -4.0 - len(dev)] + [None] * 9)
    net["device"] = np.array(dev, dtype=object)
    return net


@pytest.mark.parametrize("impl,handoff", [("radial", None), ("thread", 0), ("thread", 2)])
def test_small_tree_without_a_dpp_plan_against_oracle(impl, handoff):
    """The hybrid hand-overs of the tree kernel and the register hand-overs by child height (handed over from iteration 0 / 2:
    the lane groups do all / most of the work) on a tree OTHER than the stock 30-bus feeder, case by case against the oracle,
    diverging cases included."""
    from gym_anm_amd import codegen
    from gym_anm_amd.model import NetworkModel

    net = _small_tree()
    topo = NetworkModel(net, 0.25, 100).topology()
    hdr = codegen.emit_header(topo)
    assert "T_DPP = 0" in hdr and "T_HYB = 1" in hdr
    codegen.build_library(topo)   # (the tree's OWN library: without it a radial model runs the table-driven loop of any other)
    sim = _mesh_network_against_oracle(8, 3, 0, 192, 1.0, 6, net=net, impl=impl, handoff_after=handoff)
    assert sim.impl == impl and not sim.backend.generic


@pytest.mark.parametrize("n_bus,seed,handoff", [(5, 0, 0), (5, 2, 2)])
def test_small_trees_with_a_dpp_plan_other_than_the_stock_feeder_against_oracle(n_bus, seed, handoff):
    """The register hand-over variant of the lane-group loop on DPP lane layouts other than ANM6's (two 5-bus trees:
    codegen.dpp_plan picks among the equal-cost layouts one whose child moves land on parents or same-environment zeros, the
    kernel then sums children without a predicate): every solve handed over from iteration 0 / 2, case by case against the
    oracle -- diverging cases sharing wavefronts with converging ones included."""
    from gym_anm_amd import codegen, networks
    from gym_anm_amd.model import NetworkModel

    net = networks.synthetic_radial_network(n_bus, seed)
    topo = NetworkModel(net, 0.25, 100).topology()
    assert "T_DPP = 1" in codegen.emit_header(topo)
    codegen.build_library(topo)
    sim = _mesh_network_against_oracle(n_bus, seed, 0, 192, 1.0, 6, net=net, impl="thread", handoff_after=handoff)
    assert sim.impl == "thread" and not sim.backend.generic


def _mesh_network_against_oracle(n_bus, seed, n_chords, M, load_scale, n_hopeless, net=None, impl="mesh", handoff_after=None):
    import anm_oracle as O
    from gym_anm_amd import networks
    from gym_anm_amd.model import NetworkModel
    from gym_anm_amd.simulator import BatchedSimulator

    if net is None:
        net = networks.synthetic_meshed_network(n_bus, seed, n_chords) if n_chords else networks.synthetic_radial_network(n_bus, seed)
    model = NetworkModel(net, 0.25, 100)
    tree = len(net["branch"]) == len(net["bus"]) - 1
    kw = {} if handoff_after is None else {"handoff_after": handoff_after}
    sim = BatchedSimulator(net, 0.25, 100, num_envs=M, device=DEV, tol=1e-8, impl=impl if (impl != "mesh" or n_bus <= 12 or tree) else None, **kw)
    assert sim.impl == impl
    npt.assert_allclose(sim.device_ybus(), model.Y_bus, rtol=1e-15, atol=0)
    rng = np.random.default_rng(seed)
    b = model.baseMVA

    def U(lo, hi, scale=1.0):
        lo, hi = np.asarray(lo, float) * scale, np.asarray(hi, float) * scale
        return lo + (hi - lo) * rng.uniform(size=(M, lo.size))

    # (the synthetic feeders carry the same load per bus whatever their size: the large ones are loaded lightly)
    pl = U(model.dev_p_min[model.load_idx], 0 * model.dev_p_min[model.load_idx], 0.6 * b * load_scale)
    pp = U(0 * model.dev_p_max[model.gen_idx], model.dev_p_max[model.gen_idx], b)
    ps = U(model.dev_p_min[model.setp_idx], model.dev_p_max[model.setp_idx], 1.2 * b * load_scale)
    qs = U(model.dev_q_min[model.setp_idx], model.dev_q_max[model.setp_idx], 1.2 * b * load_scale)
    soc = U(model.dev_soc_min[model.des_idx], model.dev_soc_max[model.des_idx])
    pl[-n_hopeless:] *= 40.0 / load_scale  # a few hopeless cases: both sides must give up the same way
    sim.soc.copy_(torch.as_tensor(soc))
    sim.transition(pl, pp, ps, qs)
    full, sl = sim.full.cpu().numpy(), pc.full_slices(sim)
    n = O.parse_network(net, 0.25, 100)
    n_conv = 0
    for e in range(M):
        ref = O.transition(n, pl[e], pp[e], ps[e], qs[e], soc[e], tol=1e-8, sparse=False)
        assert bool(sim.pfe_converged[e]) == bool(ref["converged"]), e
        npt.assert_allclose(sim.soc[e].cpu().numpy(), ref["soc_after"], rtol=0, atol=1e-12)
        if not ref["converged"]:
            continue
        n_conv += 1
        assert int(sim.nr_iters[e]) == ref["n_iter"], e
        npt.assert_allclose(full[e, sl["bus_v_magn"]], np.abs(ref["V"]), rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["bus_v_ang"]], np.angle(ref["V"]), rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["bus_i_magn"]], np.abs(ref["I"]), rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["dev_p"]], ref["dev_p"], rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["branch_p"]], ref["br_p_from"], rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["branch_s"]], ref["br_s"], rtol=0, atol=1e-9)
        npt.assert_allclose(float(sim.reward[e]), ref["reward"], rtol=1e-9, atol=1e-9)
    assert n_conv >= M // 2
    return sim


def test_different_topologies_in_one_batch_against_their_own_oracles():
    """The reference builds one Simulator per environment from any dict (simulator.py:70-111, examples/custom_anm6.py:20).
    Here: ANM6, the 3-bus loop and a random meshed 20-bus network, dealt at random to the 4 096 environments of ONE batch
    (MixedBatchedSimulator: rows padded to the widest network, one launch of the general lane-group kernel per topology
    through anm_model_bind_view, nothing copied around them).  Every 8th environment -- and every one of the hopeless
    cases -- against the oracle of ITS OWN network: flags and Newton iteration counts exact, electrical state <= 1e-9; the
    padding of every row untouched."""
    import anm_oracle as O
    from gym_anm_amd import networks
    from gym_anm_amd.simulator import MixedBatchedSimulator

    nets = [networks.anm6_network(), networks.three_bus_loop_network(gen_max=1.5), networks.synthetic_meshed_network(20, 3, 6)]
    E_ = 4096
    rng = np.random.default_rng(5)
    env_net = rng.integers(0, len(nets), E_)
    sim = MixedBatchedSimulator(nets, env_net, 0.25, 100, device=DEV, tol=1e-8)
    # (round 5: every kernel family serves a view -- each network keeps the family that suits it)
    assert len({s.model.topology()[0] for s in sim.subs}) == 3 and [s.impl for s in sim.subs] == ["thread", "thread", "mesh"]
    w = sim.widths
    pl, pp = np.zeros((E_, w["load"])), np.zeros((E_, w["gen"]))
    ps, qs, soc = np.zeros((E_, w["setp"])), np.zeros((E_, w["setp"])), np.zeros((E_, w["des"]))
    hopeless = rng.random(E_) < 0.01
    for k, sub in enumerate(sim.subs):
        m, b = sub.model, sub.model.baseMVA
        idx = np.nonzero(env_net == k)[0]

        def U(lo, hi, scale=1.0):
            lo, hi = np.asarray(lo, float) * scale, np.asarray(hi, float) * scale
            return lo + (hi - lo) * rng.uniform(size=(idx.size, lo.size))

        pl[idx, : m.N_load] = U(m.dev_p_min[m.load_idx], 0 * m.dev_p_min[m.load_idx], 0.6 * b)
        pp[idx, : m.N_non_slack_gen] = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx], b)
        ps[idx, : len(m.setp_idx)] = U(m.dev_p_min[m.setp_idx], m.dev_p_max[m.setp_idx], 1.2 * b)
        qs[idx, : len(m.setp_idx)] = U(m.dev_q_min[m.setp_idx], m.dev_q_max[m.setp_idx], 1.2 * b)
        soc[idx, : m.N_des] = U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx])
    pl[hopeless] *= 40.0   # both sides must give up the same way
    sim.soc[:, : w["des"]].copy_(torch.as_tensor(soc))
    sim.full.fill_(-777.0)
    full, r, el, pen, conv = sim.transition(pl, pp, ps, qs)
    full, conv, iters = full.cpu().numpy(), conv.cpu().numpy(), sim.nr_iters.cpu().numpy()
    soc_after, reward = sim.soc.cpu().numpy(), r.cpu().numpy()
    parsed = [O.parse_network(n, 0.25, 100) for n in nets]
    n_checked, n_conv, n_fail = np.zeros(3, int), 0, 0
    for e in range(E_):
        k = int(env_net[e])
        sub, m = sim.layout(k), sim.layout(k).model
        assert (full[e, sub.full_dim:] == -777.0).all(), "environment %d wrote beyond the row of its network" % e
        if e % 8 and not hopeless[e]:
            continue
        ref = O.transition(parsed[k], pl[e, : m.N_load], pp[e, : m.N_non_slack_gen], ps[e, : len(m.setp_idx)], qs[e, : len(m.setp_idx)],
                           soc[e, : m.N_des], tol=1e-8, sparse=False)
        n_checked[k] += 1
        assert bool(conv[e]) == bool(ref["converged"]), e
        npt.assert_allclose(soc_after[e, : m.N_des], ref["soc_after"], rtol=0, atol=1e-12)
        if not ref["converged"]:
            n_fail += 1
            continue
        n_conv += 1
        sl = pc.full_slices(sub)
        assert int(iters[e]) == ref["n_iter"], e
        npt.assert_allclose(full[e, sl["bus_v_magn"]], np.abs(ref["V"]), rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["bus_v_ang"]], np.angle(ref["V"]), rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["bus_i_magn"]], np.abs(ref["I"]), rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["dev_p"]], ref["dev_p"], rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["dev_q"]], ref["dev_q"], rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["branch_p"]], ref["br_p_from"], rtol=0, atol=1e-9)
        npt.assert_allclose(full[e, sl["branch_s"]], ref["br_s"], rtol=0, atol=1e-9)
        npt.assert_allclose(reward[e], ref["reward"], rtol=1e-9, atol=1e-9)
    assert (n_checked >= 100).all() and n_conv >= 400 and n_fail >= 5, (n_checked, n_conv, n_fail)

    # Simulator.reset over the mixed batch (simulator.py:225-293) == BatchedSimulator.reset of every network on its own rows
    from gym_anm_amd.simulator import BatchedSimulator

    s0 = np.zeros((E_, max(2 * s.model.N_device + s.model.N_des + s.model.N_non_slack_gen for s in sim.subs)))
    for k, sub in enumerate(sim.subs):
        m, b = sub.model, sub.model.baseMVA
        idx = np.nonzero(env_net == k)[0]
        D, nd, ng = m.N_device, m.N_des, m.N_non_slack_gen
        s0[idx, :D] = rng.uniform(-1, 1, (idx.size, D)) * np.maximum(np.abs(np.nan_to_num(m.dev_p_min, posinf=1, neginf=-1)), np.abs(np.nan_to_num(m.dev_p_max, posinf=1, neginf=-1))) * b * 0.5
        s0[idx, D : 2 * D] = rng.uniform(-0.2, 0.2, (idx.size, D)) * b * 0.1
        s0[idx, 2 * D : 2 * D + nd] = rng.uniform(0.1, 0.9, (idx.size, nd)) * (m.dev_soc_max[m.des_idx] * b)
        s0[idx, 2 * D + nd : 2 * D + nd + ng] = rng.uniform(0, 1, (idx.size, ng)) * m.dev_p_max[m.gen_idx] * b
    conv_mixed = sim.reset(s0).cpu().numpy()
    for k, net in enumerate(nets):
        idx = np.nonzero(env_net == k)[0][:64]
        alone = BatchedSimulator(net, 0.25, 100, num_envs=idx.size, device=DEV, tol=1e-8, impl=sim.subs[k].impl)
        S_k = 2 * alone.model.N_device + alone.model.N_des + alone.model.N_non_slack_gen
        conv_alone = alone.reset(s0[idx, :S_k]).cpu().numpy()
        npt.assert_array_equal(conv_mixed[idx], conv_alone)
        assert torch.equal(sim.full[torch.as_tensor(idx, device=DEV)][:, : alone.full_dim], alone.full)
        assert torch.equal(sim.soc[torch.as_tensor(idx, device=DEV)][:, : alone.model.N_des], alone.soc)


@pytest.mark.parametrize("n_bus,seed,n_chords", [(30, 6, 4), (64, 10, 24), (200, 13, 30)])
def test_mesh_fused_levels_schedule_equals_the_default(monkeypatch, n_bus, seed, n_chords):
    """ANM_MESH_FUSED_LEVELS (read when the model is created): product and subtraction of an elimination level in one
    step.  Same operations on the same operands in the same order as the default schedule: bit-identical outputs."""
    from gym_anm_amd import networks
    from gym_anm_amd.simulator import BatchedSimulator

    net = networks.synthetic_meshed_network(n_bus, seed, n_chords)
    M = 256
    sims = []
    for fused in (False, True):
        if fused:
            monkeypatch.setenv("ANM_MESH_FUSED_LEVELS", "1")
        sims.append(BatchedSimulator(net, 0.25, 100, num_envs=M, device=DEV, tol=1e-8, impl="mesh"))
    monkeypatch.delenv("ANM_MESH_FUSED_LEVELS")
    model, rng = sims[0].model, np.random.default_rng(seed)
    b, f = model.baseMVA, min(1.0, 40.0 / n_bus)

    def U(lo, hi, scale=1.0):
        lo, hi = np.asarray(lo, float) * scale, np.asarray(hi, float) * scale
        return lo + (hi - lo) * rng.uniform(size=(M, lo.size))

    pl = U(model.dev_p_min[model.load_idx], 0 * model.dev_p_min[model.load_idx], 0.6 * b * f)
    pp = U(0 * model.dev_p_max[model.gen_idx], model.dev_p_max[model.gen_idx], b)
    ps = U(model.dev_p_min[model.setp_idx], model.dev_p_max[model.setp_idx], 1.2 * b * f)
    qs = U(model.dev_q_min[model.setp_idx], model.dev_q_max[model.setp_idx], 1.2 * b * f)
    soc = U(model.dev_soc_min[model.des_idx], model.dev_soc_max[model.des_idx])
    pl[-4:] *= 40.0 / f
    for sim in sims:
        sim.soc.copy_(torch.as_tensor(soc))
        sim.transition(pl, pp, ps, qs)
    assert int(sims[0].pfe_converged.sum()) >= M // 2
    assert torch.equal(sims[0].pfe_converged, sims[1].pfe_converged) and torch.equal(sims[0].nr_iters, sims[1].nr_iters)
    conv = sims[0].pfe_converged
    assert torch.equal(sims[0].full[conv], sims[1].full[conv]) and torch.equal(sims[0].reward[conv], sims[1].reward[conv])


@pytest.mark.parametrize("n_bus,seed,n_chords,switch", [(20, 3, 6, ("ANM_MESH_SIMD_WAVES", "2", "3")), (14, 5, 3, ("ANM_MESH_SIMD_WAVES", "2", "3")),
                                                        (64, 9, 20, ("ANM_MESH_TABLES", "lds", "global")), (30, 6, 4, ("ANM_MESH_TABLES", "lds", "global"))])
def test_mesh_launch_variants_are_bit_identical(monkeypatch, n_bus, seed, n_chords, switch):
    """The variants of the general lane-group kernel a plan may be launched with (mesh::Launch, read when the model is
    created): registers budgeted for two or three wavefronts per SIMD, tables staged in LDS or read from global memory.
    Same operations on the same operands in the same order: bit-identical outputs, diverging solves included."""
    from gym_anm_amd import networks
    from gym_anm_amd.simulator import BatchedSimulator

    net = networks.synthetic_meshed_network(n_bus, seed, n_chords)
    M = 512
    sims = []
    for value in switch[1:]:
        monkeypatch.setenv(switch[0], value)
        sims.append(BatchedSimulator(net, 0.25, 100, num_envs=M, device=DEV, tol=1e-8, impl="mesh"))
    monkeypatch.delenv(switch[0])
    model, rng = sims[0].model, np.random.default_rng(seed)
    b, f = model.baseMVA, min(1.0, 40.0 / n_bus)

    def U(lo, hi, scale=1.0):
        lo, hi = np.asarray(lo, float) * scale, np.asarray(hi, float) * scale
        return lo + (hi - lo) * rng.uniform(size=(M, lo.size))

    pl = U(model.dev_p_min[model.load_idx], 0 * model.dev_p_min[model.load_idx], 0.6 * b * f)
    pp = U(0 * model.dev_p_max[model.gen_idx], model.dev_p_max[model.gen_idx], b)
    ps = U(model.dev_p_min[model.setp_idx], model.dev_p_max[model.setp_idx], 1.2 * b * f)
    qs = U(model.dev_q_min[model.setp_idx], model.dev_q_max[model.setp_idx], 1.2 * b * f)
    soc = U(model.dev_soc_min[model.des_idx], model.dev_soc_max[model.des_idx])
    pl[-4:] *= 40.0 / f
    for sim in sims:
        sim.soc.copy_(torch.as_tensor(soc))
        sim.transition(pl, pp, ps, qs)
    assert int(sims[0].pfe_converged.sum()) >= M // 2
    bits = lambda a: a.view(torch.int64) if a.dtype == torch.float64 else a   # (NaN == NaN as bit patterns)
    for name in ("pfe_converged", "nr_iters", "full", "reward", "soc"):
        assert torch.equal(bits(getattr(sims[0], name)), bits(getattr(sims[1], name))), name


def test_environment_over_a_network_larger_than_a_wavefront():
    env = pc.large_network_env(KW)
    assert env.simulator.lanes_per_env == 256


def test_bounds_hook_reads_aux_bounds():
    pc.bounds_hook_reads_aux_bounds(KW)


def test_convergence_flags_after_a_masked_reset():
    pc.convergence_flags_after_a_masked_reset(KW, E_=4096)


def test_general_step_with_rows_wider_than_the_default_lds_limit():
    pc.general_step_with_wide_rows(KW)


@pytest.mark.parametrize("impl", ["radial", "mesh"])
def test_a_different_network_in_every_environment(impl):
    pc.per_environment_networks(KW, impl)


def test_a_different_feeder_in_every_environment_on_the_hybrid_tree_kernel():
    """The per-environment-constants variant of the tree kernel (k_radial<.., Topo, PG>: vector loads of each lane group's own
    network) on the stock 30-bus feeder, i.e. with the hybrid hand-overs of round 6: 8 perturbed feeders dealt at random,
    every checked transition against the oracle of its own network."""
    from gym_anm_amd import networks

    sim = pc.per_environment_networks(KW, "radial", n_variants=8, E_=128, n_check=32, base=networks.synthetic_radial_network(30, 0))
    assert sim.impl == "radial" and not sim.backend.generic and sim.lanes_per_env == 32


def test_reset_and_step_through_views_of_a_mixed_batch():
    """anm_reset_f64 / anm_step_f64 with a view bound (include/anm_mi355x.h: anm_model_bind_view): environments over three
    topologies interleaved in ONE batch whose rows are padded to the widest network, each topology's launch told which
    rows are its own -- against the same environments run on their own (same kernel family, so every number is equal),
    and the padding of every row untouched.  ANMEnv.reset / step: anm_env.py:235-311, 333-453."""
    import ctypes as C

    from gym_anm_amd import _lib, networks
    from gym_anm_amd.envs.anm_env import BatchedANMEnv, _stream_ptr

    nets = [networks.anm6_network(), networks.three_bus_loop_network(gen_max=1.5), networks.synthetic_meshed_network(20, 3, 6)]
    per = 96
    E_ = per * len(nets)
    env_net = np.arange(E_) % len(nets)

    def make(net, k):
        class Task(BatchedANMEnv):
            def __init__(self):
                super().__init__(net, "state", 1, 0.25, 0.99, 100, np.array([[0, 24]]), (10, 500), 3, num_envs=per, device=DEV,
                                 tol=1e-8, impl="mesh")

            def init_state(self):
                raise AssertionError("the test resets to explicit states")

            def next_vars(self, s_t):
                raise AssertionError("the test passes the exogenous values itself")

        return Task()

    own = [make(n, k) for k, n in enumerate(nets)]
    viewed = [make(n, k) for k, n in enumerate(nets)]      # their models serve the padded batch through a view
    dims = [e.simulator for e in own]
    W = lambda f: max(f(s) for s in dims)  # noqa: E731
    w_act = W(lambda s: s.dims.action_dim)
    w_st = max(e.state_N for e in own)
    w_exo = W(lambda s: s.N_load + s.N_non_slack_gen)
    w_des = max(1, W(lambda s: s.N_des))
    PAD = -777.0
    f64 = dict(dtype=torch.float64, device=DEV)
    state, obs, init = (torch.full((E_, w_st), PAD, **f64) for _ in range(3))
    soc = torch.full((E_, w_des), PAD, **f64)
    action = torch.full((E_, w_act), PAD, **f64)
    exo = torch.full((E_, w_exo), PAD, **f64)
    aux = torch.full((E_, 1), PAD, **f64)
    reward, e_loss, penalty = (torch.zeros(E_, **f64) for _ in range(3))
    conv, term = (torch.zeros(E_, dtype=torch.uint8, device=DEV) for _ in range(2))
    timestep, nr_iters, reset_count = (torch.zeros(E_, dtype=torch.int32, device=DEV) for _ in range(3))
    idx, views = [], []
    for k, v in enumerate(viewed):
        ix = torch.as_tensor(np.nonzero(env_net == k)[0], dtype=torch.int32, device=DEV)
        idx.append(ix)
        bv = _lib.BatchView(ix.data_ptr(), 0, 0, 0, w_des, w_act, w_st, w_exo, 1, 0)
        views.append(bv)
        sim = v.simulator
        sim.backend.check(sim.backend.lib.anm_model_bind_view(sim._handle, C.byref(bv)), "anm_model_bind_view")

    g = torch.Generator(device="cpu").manual_seed(5)

    def rows_equal(k, what):
        o, sim, ix = own[k], own[k].simulator, idx[k].long()
        sd, nd = o.state_N, sim.N_des
        torch.cuda.synchronize()
        for name, mine, theirs in (("state", state[ix, :sd], o._state_buf), ("obs", obs[ix, :sd], o._state_obs),
                                   ("terminated", term[ix], o._term_u8), ("timestep", timestep[ix], o.timestep),
                                   ("nr_iters", nr_iters[ix], sim.nr_iters)):
            assert torch.equal(mine, theirs), (what, k, name)
        if nd:
            assert torch.equal(soc[ix, :nd], sim.soc[:, :nd]), (what, k, "soc")
        assert bool((state[ix, sd:] == PAD).all()) and bool((obs[ix, sd:] == PAD).all()) and bool((soc[ix, nd:] == PAD).all())

    # ---- reset to explicit initial states (every 17th far outside the device ranges: mapped back like the others)
    for k, o in enumerate(own):
        sim, m = o.simulator, o.simulator.model
        D, nd, ng, sd = m.N_device, m.N_des, m.N_non_slack_gen, o.state_N
        s0 = torch.zeros((per, sd), dtype=torch.float64)
        u = torch.rand((per, sd), generator=g, dtype=torch.float64)
        s0[:, :D] = (2 * u[:, :D] - 1) * 3.0
        s0[:, D:2 * D] = (2 * u[:, D:2 * D] - 1) * 0.5
        s0[:, 2 * D:2 * D + nd] = u[:, 2 * D:2 * D + nd] * 5.0
        s0[:, 2 * D + nd:2 * D + nd + ng] = u[:, 2 * D + nd:2 * D + nd + ng] * 10.0
        s0[:, -1] = 7.0
        s0[::17, :D] *= 40.0
        s0 = s0.to(DEV)
        o._launch_reset(s0, None)
        init[idx[k].long(), :sd] = s0
        v = viewed[k].simulator
        rc = v.backend.lib.anm_reset_f64(
            v._handle, per, init.data_ptr(), None, 3, 0, reset_count.data_ptr(), soc.data_ptr(), state.data_ptr(), obs.data_ptr(),
            conv.data_ptr(), term.data_ptr(), timestep.data_ptr(), nr_iters.data_ptr(), None, None, C.byref(v.opts), _stream_ptr(o.device))
        v.backend.check(rc, "anm_reset_f64")
        rows_equal(k, "reset")
        assert torch.equal(conv[idx[k].long()], o._conv_u8)

    # ---- steps with the exogenous values handed over (anm_env.py:369-397)
    for t in range(5):
        for k, o in enumerate(own):
            sim, m = o.simulator, o.simulator.model
            ix = idx[k].long()
            na, ne = sim.dims.action_dim, sim.N_load + sim.N_non_slack_gen
            lo, hi = torch.as_tensor(o.action_space.low), torch.as_tensor(o.action_space.high)
            a = (lo + (hi - lo) * torch.rand((per, na), generator=g, dtype=torch.float64)).to(DEV).contiguous()
            xlo = torch.as_tensor(np.concatenate((m.dev_p_min[m.load_idx], 0 * m.dev_p_max[m.gen_idx])) * m.baseMVA)
            xhi = torch.as_tensor(np.concatenate((0 * m.dev_p_min[m.load_idx], m.dev_p_max[m.gen_idx])) * m.baseMVA)
            x = (xlo + (xhi - xlo) * torch.rand((per, ne), generator=g, dtype=torch.float64)).to(DEV).contiguous()
            ax = torch.full((per, 1), float((8 + t) % 24), **f64)
            o._step_call(a.data_ptr(), x.data_ptr(), ax.data_ptr())
            action[ix, :na], exo[ix, :ne], aux[ix] = a, x, ax
            v = viewed[k].simulator
            rc = v.backend.lib.anm_step_f64(
                v._handle, per, action.data_ptr(), exo.data_ptr(), aux.data_ptr(), soc.data_ptr(), state.data_ptr(), term.data_ptr(),
                timestep.data_ptr(), obs.data_ptr(), reward.data_ptr(), e_loss.data_ptr(), penalty.data_ptr(), nr_iters.data_ptr(),
                None, 0, 3, 0, reset_count.data_ptr(), None, None, C.byref(v.opts), _stream_ptr(o.device))
            v.backend.check(rc, "anm_step_f64")
            rows_equal(k, "step %d" % t)
            for name, mine, theirs in (("reward", reward[ix], o.reward), ("e_loss", e_loss[ix], o.e_loss),
                                       ("penalty", penalty[ix], o.penalty)):
                assert torch.equal(mine, theirs), (t, k, name)


@pytest.mark.parametrize("impl,E", [("thread", 65536), ("radial", 16384)])
def test_a_solve_that_blows_up_does_not_reach_its_neighbours(impl, E):
    """The lane groups of a wavefront hold different environments, and some hand-overs of the lane-group Newton loop
    are predicate-free (round 6: child sums through a register that is zero wherever no move writes; padding lanes
    whose products are 0 x whatever their own moves read).  None of that may carry a diverging neighbour's Inf / NaN
    into a healthy solve: the batch with every non-converging environment replaced by a benign one must leave all OTHER
    environments bit for bit as they were -- including the slow converging solves that shared lane groups with them."""
    from gym_anm_amd import networks
    from gym_anm_amd.simulator import BatchedSimulator

    g = torch.Generator(device=DEV).manual_seed(7)

    def U(lo, hi, n):
        return lo + (hi - lo) * torch.rand((E, n), generator=g, dtype=torch.float64, device=DEV)

    pl = -U(0, 1, 3) * torch.tensor([10.0, 30.0, 30.0], device=DEV)
    pp = U(0, 1, 2) * torch.tensor([30.0, 50.0], device=DEV)
    ps = torch.cat([U(0, 30, 1), U(0, 50, 1), U(-50, 50, 1)], 1)
    qs = U(-50, 50, 3)
    soc = U(0, 100, 1) / 100

    def run(pl, pp, ps, qs, soc):
        sim = BatchedSimulator(networks.anm6_network(), 0.25, 100, num_envs=E, device=DEV, tol=1e-6, max_iter=100, impl=impl)
        sim.soc.copy_(soc)
        sim.transition(pl, pp, ps, qs)
        torch.cuda.synchronize()
        return sim.full.clone(), sim.pfe_converged.clone(), sim.nr_iters.clone()

    full1, conv1, it1 = run(pl, pp, ps, qs, soc)
    bad = ~conv1
    assert int(bad.sum()) >= E // 1000 and bool(conv1[0])              # a few hundred solves that run to the cap or fail
    assert int((conv1 & (it1 > 6)).sum()) >= 8                          # ... and slow converging ones that met them on lane groups
    rep = lambda x: torch.where(bad[:, None], x[0:1].expand_as(x), x)  # noqa: E731  (environment 0 converges)
    full2, conv2, it2 = run(rep(pl), rep(pp), rep(ps), rep(qs), rep(soc))
    keep = ~bad
    assert bool(conv2.all())
    assert torch.equal(it1[keep], it2[keep]) and torch.equal(conv1[keep], conv2[keep])
    assert torch.equal(full1[keep].view(torch.int64), full2[keep].view(torch.int64))
