"""The Newton step itself, pinned against the reference (`_dfdx`, `spsolve`).

``tests/golden/iterates_<net>.npz`` hold, per case, the iterate x_k and the mismatch ||F(x_k)||inf the UNMODIFIED
``_newton_raphson_sparse(..., lim_iter=k)`` returns for k = 1 ... 12 (``gym_anm/simulator/solve_load_flow.py:176-226``;
``oracle/make_golden_iterates.py``).  CPU tier: the oracle's restatement reproduces them bit for bit (it IS the same
scipy.sparse + spsolve path) and the host test double follows them to the GPU tier's tolerances.  GPU tier: every
kernel family through the C ABI with ``max_iter = k``."""
import os

import numpy as np
import numpy.testing as npt
import pytest

import anm_oracle as O
import parity_common as PC
from conftest import GOLDEN


@pytest.mark.parametrize("name", PC.ITER_NETS)
def test_fixture_shape_and_content(name):
    g = np.load(os.path.join(GOLDEN, "iterates_%s.npz" % name))
    M = len(g["n_iter"])
    assert M >= 200 and int(g["diverging"].sum()) >= 20
    assert np.array_equal(g["n_k"], np.minimum(g["n_iter"], 12))
    for m in range(M):   # exactly the recorded iterates are finite-or-NaN data, the rest is padding
        assert np.all(np.isnan(g["vm_k"][m, g["n_k"][m]:]))
    assert np.all(g["diff_k"][np.arange(M), g["n_k"]][g["n_iter"] <= 12] == g["diff"][g["n_iter"] <= 12])


@pytest.mark.parametrize("name", PC.ITER_NETS)
def test_oracle_reproduces_the_reference_iterates_bit_for_bit(name):
    g = np.load(os.path.join(GOLDEN, "iterates_%s.npz" % name))
    net = O.parse_network(PC.iterate_nets()[name], float(g["delta_t"]), float(g["lamb"]))
    rows = np.concatenate([np.arange(0, 180, 9), np.flatnonzero(g["diverging"])[:6]])   # a sample: spsolve is slow
    for m in rows:
        p, q = g["bus_p"][m, 1:], g["bus_q"][m, 1:]
        for k in range(1, int(g["n_k"][m]) + 1):
            x, it, diff, _ = O.newton_raphson(p, q, net.Y, tol=float(g["x_tol"]), max_iter=k, sparse=True)
            N1 = len(p)
            assert it == k
            npt.assert_array_equal(x[:N1], g["th_k"][m, k - 1])
            npt.assert_array_equal(x[N1:], g["vm_k"][m, k - 1])
            npt.assert_array_equal(diff, g["diff_k"][m, k])


@pytest.mark.parametrize("name", ["anm6", "3bus", "3bus_tx2", "3bus_tx7"])
def test_host_double_follows_the_reference_iterates(name):
    """kernel LOGIC on the host test double (g++, 1/x for the pivots, no contraction): same tolerances as the GPU tier"""
    from gym_anm_amd.model import NetworkModel
    from hostsim_backend import hostsim_backend

    net = PC.iterate_nets()[name]
    be = hostsim_backend(NetworkModel(net, 0.25, 100).topology())
    w = PC.newton_iterates_transition(name, net, "cpu", backend=be)
    assert w["conv"] <= PC.ITER_ATOL_CONV and w["div"] <= PC.ITER_RTOL_DIV
    PC.newton_iterates_step(name, net, "cpu", backend=be)
    w = PC.newton_one_step(name, net, "cpu", backend=be)
    assert w["conv"] <= PC.ITER_ATOL_CONV


# ---------------------------------------------------------------------------------------------------------------
# GPU tier: the three kernel families, and the lane-group hand-over of the thread family
# ---------------------------------------------------------------------------------------------------------------
FAMILIES = {"anm6": ("thread", "radial", "mesh"), "case30": ("radial", "mesh"), "3bus": ("thread", "mesh"),
            "3bus_tx2": ("thread", "mesh"), "3bus_tx7": ("thread", "mesh")}
CASES = [(n, f) for n in PC.ITER_NETS for f in FAMILIES[n]]


@pytest.mark.gpu
@pytest.mark.parametrize("name,impl", CASES)
def test_newton_iterates_against_the_reference_gpu(name, impl):
    w = PC.newton_iterates_transition(name, PC.iterate_nets()[name], "cuda", impl=impl)
    print("\n%s / %s: worst iterate deviation converging %.3g p.u. (allowed %.1g), diverging %.3g relative (allowed %.1g), "
          "mismatch %.3g relative (allowed %.1g + %.1g absolute)"
          % (name, impl, w["conv"], PC.ITER_ATOL_CONV, w["div"], PC.ITER_RTOL_DIV, w["diff"], PC.ITER_DIFF_RTOL, PC.ITER_DIFF_ATOL))


@pytest.mark.gpu
@pytest.mark.parametrize("name,impl", CASES)
def test_one_newton_step_from_the_reference_iterate_gpu(name, impl):
    """every recorded iterate of every case, diverging ones included: one step of this Jacobian and this linear solve from
    the reference's x_{k-1} lands on the reference's x_k"""
    w = PC.newton_one_step(name, PC.iterate_nets()[name], "cuda", impl=impl)
    print("\n%s / %s: ONE step from the reference's iterate: converging %.3g p.u. (allowed %.1g), diverging %.3g relative = at most "
          "%.2f eps cond(J) (allowed max(%.0e, %d eps cond)), mismatch %.3g relative"
          % (name, impl, w["conv"], PC.ITER_ATOL_CONV, w["div"], w.get("div_over_eps_cond", 0.0), PC.ITER_RTOL_DIV / 10, PC.ONE_STEP_COND_FACTOR, w["diff"]))


@pytest.mark.gpu
@pytest.mark.parametrize("handoff", [None, 0, 2])
def test_newton_iterates_through_the_lane_group_hand_over_gpu(handoff):
    """ANM6 through anm_step_f64: in-lane only, every solve on a lane group from its first iteration, and the mixed path"""
    w = PC.newton_iterates_step("anm6", PC.iterate_nets()["anm6"], "cuda", handoff_after=handoff)
    print("\nanm6 / thread, hand-over %r: worst iterate deviation converging %.3g p.u., diverging %.3g relative" % (handoff, w["conv"], w["div"]))
