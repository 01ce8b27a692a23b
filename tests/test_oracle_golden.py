"""Pin the CPU oracle (oracle/anm_oracle.py) to the reference.

Sources of truth, in order:
1. golden vectors recorded from the unmodified reference (oracle/make_golden.py);
2. the reference's own known-answer tests, restated here as data:
   tests/simulator/test_simulator_basics.py:48-67 (Y_bus), :247-293 (reward),
   tests/simulator/test_devices.py:74-97, 269-295, 456-492, 523-562 (device maps).
"""
import os
import sys

import numpy as np
import numpy.testing as npt
import pytest

import anm_oracle as O
from gym_anm_amd import networks

from conftest import GOLDEN

CMP = ["dev_p", "dev_q", "soc_after", "p_pot", "bus_p", "bus_q", "V", "I", "br_p_from", "br_q_from", "br_p_to",
       "br_q_to", "br_s", "br_i_from", "br_i_to", "reward", "e_loss", "penalty"]  # fmt: skip


def _nets():
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


NETS = _nets()


def _check_transition(name, sparse, stride):
    g = np.load(os.path.join(GOLDEN, "transition_%s.npz" % name))
    n = O.parse_network(NETS[name], float(g["delta_t"]), float(g["lamb"]))
    npt.assert_array_equal(n.Y, g["Y_bus"])  # bit-exact admittance matrix
    M = len(g["n_iter"])
    n_div = 0
    for m in range(0, M, stride):
        out = O.transition(n, g["P_load"][m], g["P_pot"][m], g["P_set"][m], g["Q_set"][m], g["soc0"][m], sparse=sparse)
        # post-projection injections do not depend on the power-flow outcome
        npt.assert_allclose(out["dev_p"][1:], g["dev_p"][m][1:], rtol=0, atol=1e-12)
        npt.assert_allclose(out["dev_q"][1:], g["dev_q"][m][1:], rtol=0, atol=1e-12)
        npt.assert_allclose(out["soc_after"], g["soc_after"][m], rtol=0, atol=1e-12)
        assert out["converged"] == bool(g["converged"][m]), (name, m)
        if not g["converged"][m]:
            n_div += 1
            continue
        assert out["n_iter"] == int(g["n_iter"][m]), (name, m)
        for k in CMP:
            npt.assert_allclose(np.asarray(out[k]), g[k][m], rtol=1e-9, atol=1e-10, err_msg="%s[%d] %s" % (name, m, k))
    return n_div


@pytest.mark.parametrize("name", sorted(NETS))
def test_transition_dense_checker(name):
    """Dense-LAPACK variant of the oracle (the fast checker used by the GPU parity tests)."""
    _check_transition(name, sparse=False, stride=1 if name != "anm6" else 3)


@pytest.mark.parametrize("name", ["anm6", "3bus", "3bus_tx3", "case30"])
def test_transition_scipy_sparse_literal(name):
    """Literal scipy.sparse + spsolve variant (what the reference executes)."""
    _check_transition(name, sparse=True, stride=7)


def test_survey_known_answer():
    g = np.load(os.path.join(GOLDEN, "transition_anm6.npz"))
    # row 0 of the anm6 fixture is the hand-checked case of SURVEY.md 8(c)
    assert int(g["n_iter"][0]) == 3
    npt.assert_allclose(g["reward"][0], -7.973918826296909, rtol=1e-12)
    npt.assert_allclose(g["soc_after"][0], [0.4722222222222222], rtol=1e-14)
    npt.assert_allclose(np.abs(g["V"][0])[1], 1.006185017186, atol=1e-11)


@pytest.mark.parametrize("name,net", [("anm6", networks.anm6_network()),
                                      ("3bus", networks.three_bus_loop_network(base_mva=10, gen_max=100.0))])  # fmt: skip
def test_reset(name, net):
    g = np.load(os.path.join(GOLDEN, "reset_%s.npz" % name))
    n = O.parse_network(net, float(g["delta_t"]), float(g["lamb"]))
    for m in range(0, len(g["n_iter"]), 2):
        out = O.sim_reset(n, g["init_state"][m], sparse=False)
        assert out["converged"] == bool(g["converged"][m])
        npt.assert_allclose(out["soc_after"], g["soc_after"][m], rtol=0, atol=1e-13)
        if g["converged"][m]:
            for k in ["dev_p", "dev_q", "V", "br_s"]:
                npt.assert_allclose(np.asarray(out[k]), g[k][m], rtol=1e-9, atol=1e-10)


def test_anm6easy_episodes():
    g = np.load(os.path.join(GOLDEN, "anm6easy_episodes.npz"))
    npt.assert_array_equal(O.anm6easy_tables().shape, (5, 96))
    draw_off = np.concatenate(([0], np.cumsum(g["init_draws_count"])))
    reset_off = np.concatenate(([0], np.cumsum(g["reset_at_count"])))
    for e in range(len(g["seed"])):
        env = O.OracleEnv(networks.anm6_network(), sparse=False)
        npt.assert_array_equal(env.obs_low, g["obs_low"])
        npt.assert_array_equal(env.obs_high, g["obs_high"])
        draws = g["init_draws"][draw_off[e] : draw_off[e + 1]]
        d = 0
        obs, conv = env.reset_to(draws[d])
        d += 1
        assert conv
        npt.assert_allclose(obs, g["obs0"][e], rtol=1e-9, atol=1e-9)
        resets = list(g["reset_at"][reset_off[e] : reset_off[e + 1]])
        reset_obs = g["reset_obs"][reset_off[e] : reset_off[e + 1]]
        T = 150
        for t in range(T):
            obs, r, term = env.step(g["actions"][e][t])
            assert term == bool(g["terminated"][e][t]), (e, t)
            npt.assert_allclose(obs, g["obs"][e][t], rtol=1e-9, atol=1e-8, err_msg="ep %d step %d" % (e, t))
            npt.assert_allclose(r, g["reward"][e][t], rtol=1e-9, atol=1e-9)
            npt.assert_allclose(env.e_loss, g["e_loss"][e][t], rtol=1e-9, atol=1e-10)
            npt.assert_allclose(env.penalty, g["penalty"][e][t], rtol=1e-9, atol=1e-9)
            if term:
                j = resets.index(t)
                obs, conv = env.reset_to(draws[d])
                d += 1
                assert conv
                npt.assert_allclose(obs, reset_obs[j], rtol=1e-9, atol=1e-9)


# ------------------------------------------------------------------------------------------
# the reference's known-answer tests, as data
# ------------------------------------------------------------------------------------------
_N = None


def _basics_network():
    # tests/simulator/test_simulator_basics.py:13-27 (slack bus has ID 1, 90 deg shifter, tap=2)
    return {
        "baseMVA": 10,
        "bus": np.array([[0, 1, 50, 1.1, 0.9], [2, 1, 50, 1.1, 0.9], [1, 0, 100, 1.0, 1.0]]),
        "branch": np.array([[0, 1, 0.1, 0.2, 0.3, 20, 1, 90], [1, 2, 0.4, 0.5, 0.6, 20, 2, 0]]),
        "device": np.array(
            [
                [1, 0, -1, 0.2, 0, -10, _N, _N, _N, _N, _N, _N, _N, _N, _N],
                [0, 1, 0, _N, 200, -200, 200, -200, _N, _N, _N, _N, _N, _N, _N],
                [2, 2, 2, _N, 30, 0, 30, -30, _N, _N, _N, _N, _N, _N, _N],
                [3, 2, 3, _N, 50, -50, 50, -50, _N, _N, _N, _N, 100, 0, 0.9],
            ],
            dtype=object,
        ),
    }


def test_known_ybus():
    n = O.parse_network(_basics_network(), 1, 100)
    y01, y01_sh, t01 = 1.0 / (0.1 + 0.2j), 0.3j / 2.0, np.exp(1.0j * np.pi / 2.0)
    y12, y12_sh, t12 = 1.0 / (0.4 + 0.5j), 0.6j / 2.0, 2
    Y = np.array(
        [
            [(y01 + y01_sh) / abs(t01) ** 2, -y01 / np.conj(t01), 0],
            [-y01 / t01, (y12 + y12_sh) / 4 + (y01 + y01_sh), -y12 / np.conj(t12)],
            [0, -y12 / t12, y12 + y12_sh],
        ]
    )
    npt.assert_allclose(n.Y, Y, rtol=1e-5)


def _map(spec, typ, points, delta_t=1.0, soc=None, p_pot=None, base=10.0):
    net = {
        "baseMVA": base,
        "bus": np.array([[0, 0, 50, 1.0, 1.0], [1, 1, 50, 1.1, 0.9]]),
        "branch": np.array([[0, 1, 0.01, 0.1, 0.0, 32, 1, 0]]),
        "device": np.array([[0, 0, 0, _N, 200, -200, 200, -200, _N, _N, _N, _N, _N, _N, _N], spec], dtype=object),
    }
    n = O.parse_network(net, delta_t, 100)
    out = []
    for p, q in points:
        if typ == "gen":
            pp = np.clip(n.p_max[1] if p_pot is None else p_pot, n.p_min[1], n.p_max[1])
            out.append(O.gen_map_pq(n, 1, p, q, pp))
        else:
            out.append(O.des_map_pq(n, 1, p, q, soc))
    return n, np.array(out)


def test_known_gen_map_pq_flex():  # test_devices.py:286-295
    spec = [2, 1, 1, _N, 10, 1, 2, -2, 9, _N, 1, -1, _N, _N, _N]
    points = np.array([(-1, 0.5), (5, 5), (5, -5), (12, 0), (10, 2), (10, -2)]) / 10.0
    mapped = np.array([(1, 0.5), (5, 2), (5, -2), (10, 0), (9.5, 1.5), (9.5, -1.5)]) / 10.0
    _, out = _map(spec, "gen", points)
    npt.assert_allclose(out, mapped, atol=1e-5)


def test_known_gen_map_pq_box():  # test_devices.py:269-284
    spec = [2, 1, 1, _N, 10, 1, 2, -3, _N, _N, _N, _N, _N, _N, _N]
    rng = np.random.default_rng(0)
    for _ in range(200):
        p, q = rng.uniform(-10, 10, 2) / 10.0
        p_pot = rng.uniform(1, 10) / 10.0
        _, out = _map(spec, "gen", [(p, q)], p_pot=p_pot)
        npt.assert_allclose(out[0], [np.clip(p, 0.1, min(1.0, p_pot)), np.clip(q, -0.3, 0.2)], atol=1e-5)


def test_known_storage_map_pq():  # test_devices.py:523-562
    spec = [2, 1, 3, _N, 10, -12, 20, -30, _N, _N, _N, _N, 1000, 0, 1]
    rng = np.random.default_rng(1)
    for _ in range(100):
        p = rng.choice([rng.uniform(-20, -12), rng.uniform(10.01, 20)]) / 10.0
        q = rng.choice([rng.uniform(-40, -30), rng.uniform(20.01, 30)]) / 10.0
        _, out = _map(spec, "des", [(p, q)], soc=50.0)
        assert out[0][0] == np.clip(p, -1.2, 1.0) and out[0][1] == np.clip(q, -3.0, 2.0)  # exact, like the ref test
    spec = [2, 1, 3, _N, 10, -11, 20, -30, 5, -6, 15, -25, 1000, 0, 1]
    points = np.array([(8.5, 18.5), (8.5, -28.5), (-9.5, 18.5), (-9.5, -28.5)]) / 10.0
    mapped = np.array([(7.5, 17.5), (7.5, -27.5), (-8.5, 17.5), (-8.5, -27.5)]) / 10.0
    _, out = _map(spec, "des", points, soc=50.0)
    npt.assert_allclose(out, mapped, atol=1e-7)


def test_known_soc_update():  # test_devices.py:456-492
    spec = [2, 1, 3, _N, 10, -12, 20, -30, 5, -6, 10, -15, 100, 10, 0.9]
    n, _ = _map(spec, "des", [(0, 0)], soc=5.0)
    for dt in (1.0, 0.25):
        n.delta_t = dt
        npt.assert_allclose(O.update_soc(n, 1, 5.0, 0.4) * 10, 50 - dt * 4 / 0.9, atol=1e-9)
        npt.assert_allclose(O.update_soc(n, 1, 5.0, -0.4) * 10, 50 + dt * 4 * 0.9, atol=1e-9)
    n.delta_t = 1
    assert O.update_soc(n, 1, 9.9, -1.0) == 10.0 and O.update_soc(n, 1, 1.1, 1.0) == 1.0


def test_known_reward():
    """The reference's two reward known answers (tests/simulator/test_simulator_basics.py:247-293),
    network fixture of :10-33 restated as data, through the oracle's compute_reward."""
    N_ = None
    net = {
        "baseMVA": 10,
        "bus": np.array([[0, 1, 50, 1.1, 0.9], [2, 1, 50, 1.1, 0.9], [1, 0, 100, 1.0, 1.0]]),
        "branch": np.array([[0, 1, 0.1, 0.2, 0.3, 20, 1, 90], [1, 2, 0.4, 0.5, 0.6, 20, 2, 0]]),
        "device": np.array([
            [1, 0, -1, 0.2, 0, -10, N_, N_, N_, N_, N_, N_, N_, N_, N_],
            [0, 1, 0, N_, 200, -200, 200, -200, N_, N_, N_, N_, N_, N_, N_],
            [2, 2, 2, N_, 30, 0, 30, -30, N_, N_, N_, N_, N_, N_, N_],
            [3, 2, 3, N_, 50, -50, 50, -50, N_, N_, N_, N_, 100, 0, 0.9],
        ]),
    }  # fmt: skip
    base, lamb, dt = 10.0, 100, 0.5
    n = O.parse_network(net, dt, lamb)
    assert [int(t) for t in n.dev_type] == [O.SLACK, O.LOAD, O.RENEWABLE, O.STORAGE]  # sorted by device id
    # the reference sets devices[i].p by device id: ids 0..3 -> p = [20, -5, 20, -30] MW
    dev_p = np.array([20, -5, 20, -30]) / base
    p_pot = np.array([25 / base])
    # buses sorted by id: the test assigns v to buses[0], buses[1], buses[2]
    e, pen = O.compute_reward(n, dev_p, p_pot, np.array([1.0, 1.0, 1.0]), np.array([10, 10]) / base)
    npt.assert_allclose(e, 40 * dt / base, rtol=1e-12)
    assert pen == 0
    e, pen = O.compute_reward(n, dev_p, p_pot, np.array([1.2, 1.0, 0.8]), np.array([30, 40]) / base)
    npt.assert_allclose(e, 40 * dt / base, rtol=1e-12)
    npt.assert_allclose(pen, lamb * dt * (0.2 + 30 / base), rtol=1e-12)


def _reference_available():
    try:
        import ref_harness

        return ref_harness.reference_available()
    except Exception:
        return False


@pytest.mark.skipif(not _reference_available(), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("kind,n_bus,seed,n_chords", [("mesh", 4, 31, 1), ("mesh", 7, 32, 2), ("mesh", 11, 33, 4), ("mesh", 16, 34, 3),
                                                      ("radial", 9, 35, 0), ("radial", 18, 36, 0), ("mesh", 30, 37, 6)])
def test_oracle_against_the_live_reference_on_random_networks(kind, n_bus, seed, n_chords):
    """Beyond the committed golden vectors: the oracle and the UNMODIFIED reference, run side by side on random
    networks nobody recorded (loops, line charging, a phase shifter; feeders) -- same flags and iteration counts,
    electrical state, SoC and reward to 1e-9.  Dev container only (the reference does not travel)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import make_golden as MG

    net = networks.synthetic_meshed_network(n_bus, seed, n_chords) if kind == "mesh" else networks.synthetic_radial_network(n_bus, seed)
    P_load, P_pot, P_set, Q_set, soc0 = MG.random_inputs(net, 0.25, 100, 24, seed, wild_frac=0.15)
    g = MG.run_transitions(net, 0.25, 100, P_load, P_pot, P_set, Q_set, soc0)
    n = O.parse_network(net, 0.25, 100)
    n_conv = 0
    for m in range(len(g["n_iter"])):
        out = O.transition(n, P_load[m], P_pot[m], P_set[m], Q_set[m], soc0[m], sparse=(m % 2 == 0))
        assert out["converged"] == bool(g["converged"][m]), m
        npt.assert_allclose(out["soc_after"], g["soc_after"][m], rtol=0, atol=1e-13)
        npt.assert_allclose(np.asarray(out["dev_p"])[1:], g["dev_p"][m][1:], rtol=0, atol=1e-12)   # (slack: from the solve)
        if not out["converged"]:
            continue
        n_conv += 1
        assert out["n_iter"] == int(g["n_iter"][m]), m
        for k in ["V", "I", "dev_p", "dev_q", "br_p_from", "br_q_from", "br_s"]:
            npt.assert_allclose(np.asarray(out[k]), g[k][m], rtol=1e-9, atol=1e-10, err_msg="%s[%d]" % (k, m))
        npt.assert_allclose(out["reward"], g["reward"][m], rtol=1e-9, atol=1e-9)
    assert n_conv >= 12


@pytest.mark.skipif(not _reference_available(), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("kind,n_bus,seed,n_chords", [("mesh", 6, 41, 2), ("radial", 12, 42, 0), ("mesh", 20, 43, 5)])
def test_oracle_reset_against_the_live_reference_on_random_networks(kind, n_bus, seed, n_chords):
    """Simulator.reset(init_state) (simulator.py:225-293: SoC pre-set, one transition, SoC overwrite) -- oracle and
    live reference side by side on random networks and initial states, infeasible ones included."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import make_golden as MG

    net = networks.synthetic_meshed_network(n_bus, seed, n_chords) if kind == "mesh" else networks.synthetic_radial_network(n_bus, seed)
    g = MG.run_resets(net, 0.25, 100, 24, seed)
    n = O.parse_network(net, 0.25, 100)
    n_conv = 0
    for m in range(len(g["n_iter"])):
        out = O.sim_reset(n, g["init_state"][m], sparse=False)
        assert out["converged"] == bool(g["converged"][m]), m
        npt.assert_allclose(out["soc_after"], g["soc_after"][m], rtol=0, atol=1e-13)
        if out["converged"]:
            n_conv += 1
            for k in ["dev_p", "dev_q", "V", "br_s"]:
                npt.assert_allclose(np.asarray(out[k]), g[k][m], rtol=1e-9, atol=1e-10, err_msg="%s[%d]" % (k, m))
    assert n_conv >= 8
