"""CPU tier: the step program the general lane-group kernel interprets (gym_anm_amd/csrc/anm_mesh.hpp,
mesh::build_plan) is executed on the host with the kernel's semantics -- within a step every lane reads before
any lane writes, no two lanes of a step write the same place -- on a random block matrix with the network's
sparsity pattern and compared with dense Gaussian elimination (tests/hostsim/mesh_program_check.cpp).
The reference solves whatever network it is given (solve_load_flow.py:220, SciPy's sparse LU)."""
import ctypes as C
import os
import subprocess

import pytest

from gym_anm_amd import _lib, networks
from gym_anm_amd.model import NetworkModel

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def checker():
    out = os.path.join(HERE, "hostsim", "_build")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libmesh_program_check.so")
    src = os.path.join(HERE, "hostsim", "mesh_program_check.cpp")
    deps = [src] + [os.path.join(ROOT, "gym_anm_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "gym_anm_amd", "csrc")) if f.endswith(".hpp")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(p) for p in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), src, "-o", lib],
                       check=True, capture_output=True)
    h = C.CDLL(lib)
    h.mesh_program_check.argtypes = [C.POINTER(_lib.NetworkDesc), C.c_uint64, C.POINTER(C.c_double)] + [C.POINTER(C.c_int32)] * 3
    h.mesh_plan_stats.argtypes = [C.POINTER(_lib.NetworkDesc), C.POINTER(C.c_int64)]
    return h


def _launch(checker, net):
    """mesh::launch_of of the network's plan: (group, steps, LDS doubles per environment, wavefronts per workgroup, LDS bytes per
    workgroup, wavefronts per compute unit, wavefronts per SIMD of the kernel variant, tables staged in LDS)"""
    desc, keep = _lib.network_desc(NetworkModel(net, 0.25, 100))
    out = (C.c_int64 * 8)()
    assert checker.mesh_plan_stats(C.byref(desc), out) == 0
    return tuple(out)


def _check(checker, net, seed):
    desc, keep = _lib.network_desc(NetworkModel(net, 0.25, 100))
    err, steps, levels, group = C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
    rc = checker.mesh_program_check(C.byref(desc), seed, C.byref(err), C.byref(steps), C.byref(levels), C.byref(group))
    assert rc == 0, "schedule rejected (code %d)" % rc
    assert err.value < 1e-12, err.value
    return steps.value, levels.value, group.value


def test_stock_networks(checker):
    assert _check(checker, networks.anm6_network(), 1)[2] == 8
    steps, levels, group = _check(checker, networks.synthetic_radial_network(30, 0), 2)
    assert group == 32 and levels <= 8 and steps <= 7   # independent-set rounds: ~log(n) levels for a feeder, one step each
    # 30 buses, 33 branches: the group follows the buses, a lane plays two branches
    assert _check(checker, networks.synthetic_meshed_network(30, 6, 4), 3)[2] == 32


@pytest.mark.parametrize("n_bus,seed,n_chords", [(4, 0, 1), (8, 1, 2), (12, 2, 3), (20, 3, 6), (33, 4, 8), (40, 5, 10),
                                                 (50, 7, 14), (60, 8, 4), (64, 9, 20)])
def test_random_meshed_networks(checker, n_bus, seed, n_chords):
    for s in range(3):
        _check(checker, networks.synthetic_meshed_network(n_bus, seed, n_chords), 10 * seed + s)


@pytest.mark.parametrize("n_bus,seed,n_chords,group", [(66, 11, 10, 128), (130, 16, 0, 256), (200, 13, 30, 256), (300, 14, 40, 512),
                                                       (513, 15, 40, 512)])
def test_networks_larger_than_a_wavefront(checker, n_bus, seed, n_chords, group):
    """above 65 buses the environment is a workgroup of 128 ... 512 lanes: the same program, more lanes per step"""
    net = networks.synthetic_meshed_network(n_bus, seed, n_chords) if n_chords else networks.synthetic_radial_network(n_bus, seed)
    steps, levels, g = _check(checker, net, seed)
    assert g == group and levels <= 24


def test_program_lengths(checker):
    """What a Newton trip pays for is the number of fence-separated steps: products subtract the first contribution to a
    destination themselves (single-pivot levels are one step), the back substitution goes by columns along its
    dependency chain, and a zone of trailing levels is eliminated Gauss-Jordan style when that shortens the program
    (the row-wise form of rounds 2-3 took 12 / 11 / 59 / 67 steps for these four networks)"""
    for net, most in ((networks.synthetic_meshed_network(30, 6, 4), 10), (networks.synthetic_radial_network(30, 0), 7),
                      (networks.synthetic_meshed_network(64, 9, 20), 27), (networks.synthetic_meshed_network(200, 13, 30), 31)):
        assert _check(checker, net, 5)[0] <= most


def test_no_zone_and_every_zone(checker, monkeypatch):
    """the schedules the zone search chooses among: none (ANM_MESH_NO_ZONE) is valid too, and never shorter"""
    nets = [networks.synthetic_meshed_network(20, 3, 6), networks.synthetic_meshed_network(30, 6, 4), networks.synthetic_meshed_network(100, 12, 16)]
    chosen = [_check(checker, n, 7)[0] for n in nets]
    monkeypatch.setenv("ANM_MESH_NO_ZONE", "1")
    plain = [_check(checker, n, 7)[0] for n in nets]
    assert all(c <= q for c, q in zip(chosen, plain)) and any(c < q for c, q in zip(chosen, plain))


@pytest.mark.parametrize("shape", ["star", "ring", "path", "complete", "feeders", "ladder"])
def test_structured_topologies(checker, shape):
    """shapes the random feeders do not produce (tests/parity_common.py: structured_network): a hub with 20 leaves (one
    level, many contributions to ONE diagonal block), a ring and a path (chains compressed level by level), a complete
    graph (everything is the dense end), five feeders off the slack (several elimination roots: no common tail), a ladder"""
    import parity_common as pc

    net = pc.structured_network(shape)
    for seed in range(3):
        steps, levels, group = _check(checker, net, 100 + seed)
        assert steps <= 3 * levels + 6


def test_launch_variant_by_plan(checker, monkeypatch):
    """mesh::launch_of: the kernel variant budgeted for three wavefronts per SIMD where the LDS of a compute unit holds twelve
    wavefronts of the plan (groups of >= 16 lanes), the tables left in global memory where their LDS copy costs wavefronts, two
    wavefronts per SIMD and staged tables otherwise -- and never more LDS than a compute unit has (measured choices:
    profiles/r05_i_mesh_occupancy.txt, r05_j_mesh_tables.txt)"""
    for var in ("ANM_MESH_SIMD_WAVES", "ANM_MESH_TABLES", "ANM_MESH_WAVES"):
        monkeypatch.delenv(var, raising=False)
    want = {"anm6": (networks.anm6_network(), 2, 1), "mesh20": (networks.synthetic_meshed_network(20, 3, 6), 3, 1),
            "mesh30": (networks.synthetic_meshed_network(30, 6, 4), 2, 1), "case30": (networks.synthetic_radial_network(30, 0), 2, 1),
            "mesh64": (networks.synthetic_meshed_network(64, 9, 20), 2, 0), "mesh200": (networks.synthetic_meshed_network(200, 13, 30), 2, 0)}
    for name, (net, simd_waves, tables) in want.items():
        g, steps, lds_env, waves, lds_block, per_cu, sw, tl = _launch(checker, net)
        assert (sw, tl) == (simd_waves, tables), name
        assert lds_block <= 160 * 1024 and 1 <= waves <= 4 and per_cu >= waves and per_cu <= 4 * sw, name
        if sw == 3:
            assert per_cu == 12 and g >= 16, name
    # the switches override the choice (read when a model is created)
    monkeypatch.setenv("ANM_MESH_SIMD_WAVES", "2")
    assert _launch(checker, want["mesh20"][0])[6] == 2
    monkeypatch.delenv("ANM_MESH_SIMD_WAVES")
    monkeypatch.setenv("ANM_MESH_TABLES", "global")
    g, steps, lds_env, waves, lds_block, per_cu, sw, tl = _launch(checker, want["mesh20"][0])
    assert (sw, tl) == (2, 0)
    monkeypatch.setenv("ANM_MESH_TABLES", "lds")
    assert _launch(checker, want["mesh64"][0])[7] == 1
