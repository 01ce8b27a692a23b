"""CPU tier: the hipcc-built libraries load, export every symbol include/anm_mi355x.h declares,
and the product refuses to run without a GPU (no silent CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from gym_anm_amd import _lib, codegen, errors, networks
from gym_anm_amd.model import NetworkModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "anm_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(anm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.ABI)


@pytest.mark.parametrize("name", ["anm6", "case30", "2bus", "3bus"])
def test_library_exports_every_symbol(name):
    topo = codegen.stock_topologies()[name]
    path = codegen.build_library(topo)  # no-op when already built
    lib = ctypes.CDLL(path)
    for sym in _declared_symbols():
        assert hasattr(lib, sym), "%s lacks %s" % (path, sym)
    _lib.bind(lib)
    assert lib.anm_topology_signature().decode() == codegen.topology_signature(topo)
    assert lib.anm_topology_name().decode() == codegen.topology_name(topo)


def test_no_cpu_fallback():
    """Without a visible MI355X the product raises instead of computing anything on the host."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from gym_anm_amd.envs import ANM6EasyVec
    from gym_anm_amd.simulator import BatchedSimulator

    with pytest.raises(errors.HipExtensionError):
        BatchedSimulator(networks.anm6_network(), 0.25, 100, num_envs=4)
    with pytest.raises(errors.HipExtensionError):
        ANM6EasyVec(num_envs=4, device="cpu")
    from gym_anm_amd.envs import MixedBatchedANMEnv
    from gym_anm_amd.envs.anm6 import anm6easy_series

    with pytest.raises(errors.HipExtensionError):   # the mixed-topology surface: no host path either
        MixedBatchedANMEnv([dict(network=networks.anm6_network(), series=anm6easy_series())], [0, 0, 0, 0])


def test_wrong_topology_is_rejected():
    """A library only accepts networks with the structure it was compiled for."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hostsim_backend import hostsim_backend
    from gym_anm_amd.simulator import BatchedSimulator

    be = hostsim_backend(NetworkModel(networks.two_bus_network(), 0.25, 100).topology())
    with pytest.raises(errors.HipExtensionError, match="built for"):
        BatchedSimulator(networks.anm6_network(), 0.25, 100, num_envs=1, device="cpu", _backend=be)


def test_build_is_keyed_by_content_not_by_file_times(tmp_path):
    """A library is fresh iff its stamp (hash of descriptor + kernel sources + flags) matches: file
    times do not matter (the built libraries travel to the GPU box with the tree), a stale stamp
    triggers a rebuild, and two processes asking for the same library at once do not collide."""
    import subprocess
    import sys
    import time

    topo = NetworkModel(networks.two_bus_network(), 0.25, 100).topology()
    lib = codegen.build_library(topo)
    stamp = lib + ".stamp"
    t_built = os.path.getmtime(lib)
    os.utime(lib, (1, 1))  # an ancient library with a matching stamp is still fresh
    assert codegen.build_library(topo) == lib and os.path.getmtime(lib) == 1
    os.utime(lib, (t_built, t_built))
    good = open(stamp).read()
    open(stamp, "w").write("stale\n")
    code = ("import sys; sys.path.insert(0, %r); from gym_anm_amd import codegen, networks; "
            "from gym_anm_amd.model import NetworkModel; "
            "print(codegen.build_library(NetworkModel(networks.two_bus_network(), 0.25, 100).topology()))" % ROOT)
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for _ in range(3)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert open(stamp).read() == good and os.path.getmtime(lib) >= t0 - 1  # rebuilt once, stamp restored
    assert not [f for f in os.listdir(os.path.dirname(lib)) if f.endswith(".tmp")]


def test_top_level_names_of_the_reference_package():
    """gym_anm/__init__.py:5-6 exports ANMEnv and the two MPC agents at top level: so does this package (lazily)."""
    import gym_anm_amd
    from gym_anm_amd import ANMEnv, MPCAgentConstant, MPCAgentPerfect
    from gym_anm_amd.agents import mpc
    from gym_anm_amd.envs.single import ANMEnv as Single

    assert ANMEnv is Single and MPCAgentPerfect is mpc.MPCAgentPerfect and MPCAgentConstant is mpc.MPCAgentConstant
    assert {"ANMEnv", "MPCAgentPerfect", "MPCAgentConstant"} <= set(dir(gym_anm_amd))


def test_stale_mpc_library_without_a_compiler_falls_back_to_a_size_class(monkeypatch):
    """libmpc_<topology>.so built from other sources / another ABI revision must not be called; where hipcc is missing the
    precompiled size class serves the network instead of raising"""
    from gym_anm_amd import _lib, codegen, networks
    from gym_anm_amd.model import NetworkModel

    topo = NetworkModel(networks.synthetic_meshed_network(20, 3, 6), 0.25, 100).topology()
    exact = codegen.lib_path(codegen.topology_name(topo), mpc_only=True)
    if not os.path.exists(exact):
        pytest.skip("the stock MPC-only library is not built here")
    monkeypatch.setattr(_lib, "_CACHE", {})
    assert _lib.load_mpc_for_topology(topo).size_class is None          # fresh: the network's own kernel
    monkeypatch.setattr(_lib, "_CACHE", {})
    monkeypatch.setattr(codegen, "mpc_library_is_fresh", lambda t: False)
    monkeypatch.setattr(codegen, "hipcc_path", lambda: None)
    be = _lib.load_mpc_for_topology(topo)
    assert be.size_class is not None and os.path.basename(be.path).startswith("libmpc_class_")


def test_register_budgets_of_the_occupancy_critical_kernels():
    """Occupancy is decided by the register count of a few kernels, and a change elsewhere in the shared templates can move it
    silently (round 5: a sincos variant pushed config 4's tree kernel from 168 to 176 VGPRs = from three to two wavefronts per
    SIMD, 77 -> 95 us, with every test green).  Read from the code objects of the built libraries."""
    import subprocess
    import tempfile

    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-readelf")):
        pytest.skip("no llvm-readelf in this image")

    def kernels(lib):
        with tempfile.TemporaryDirectory() as td:
            sec, co = os.path.join(td, "fat.bin"), os.path.join(td, "gfx950.co")
            subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + sec, lib], check=True)
            subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + sec, "--output=" + co], check=True)
            txt = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        # scratch instructions per function: a private segment nothing addresses is a frame the compiler reserved and then
        # did not need (spill slots that ended up in vector-register lanes) -- counted as no scratch
        touches, cur = {}, None
        for line in dis.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
            elif cur and re.search(r"\b(scratch_|buffer_(load|store))", line):
                touches[cur] = touches.get(cur, 0) + 1
        out = {}
        for blk in txt.split("- .agpr_count")[1:]:
            mangled = re.search(r"\.name:\s+(\S+)", blk).group(1)
            name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))  # noqa: E731
            callee = any(n for n in touches if not n.endswith(".kd") and "k_" not in n)   # (a device function that spills: counted for everybody)
            out[name] = (g("vgpr_count"), g("private_segment_fixed_size") if touches.get(mangled, 0) or callee else 0)
        return out

    topo = codegen.stock_topologies()
    budgets = {   # kernel substring -> (max VGPRs, max scratch bytes): wavefronts per SIMD = floor(512 / VGPRs)
        "anm6": {"k_step_rows<double, false>": (256, 0), "k_step_stragglers<double, true>": (256, 0), "k_step_general<double>": (256, 0),
                 "k_transition<double, false, false>": (256, 0),   # (the anm_model_bind_nr_start path is an instantiation of its own)
                 "k_reset<double, false>": (256, 0),
                 # the 6-bus feeder through the radial family (list observations / per-environment classes of a mixed batch):
                 # three wavefronts per SIMD, with per-group classes too (that instantiation keeps the lean group loop)
                 "k_radial<double, (anonymous namespace)::Topo, false>": (168, 0), "k_radial<double, (anonymous namespace)::Topo, true>": (168, 0)},
        "case30": {"k_radial<double, (anonymous namespace)::Topo, false>": (168, 0), "k_radial<double, void, false>": (168, 0),
                   "k_mesh<double, false, false, 2, true>": (256, 32),   # (32 B: four doubles of the epilogue, outside the Newton loop)
                   "k_mesh<double, false, false, 3, true>": (168, 160),   # small networks, three wavefronts per SIMD (mesh::Launch)
                   "k_mesh<double, false, false, 2, false>": (256, 32)},  # tables left in global memory
    }
    for net, want in budgets.items():
        lib = codegen.lib_path(codegen.topology_name(topo[net]))
        if not os.path.exists(lib):
            pytest.skip("library of %s not built here" % net)
        have = kernels(lib)
        for sub_, (vmax, smax) in want.items():
            hits = [(n, v) for n, v in have.items() if sub_ in n]
            assert hits, sub_
            for n, (vg, sc) in hits:
                assert vg <= vmax and sc <= smax, "%s: %d VGPRs, %d B scratch (budget %d / %d)" % (n[:90], vg, sc, vmax, smax)


def test_mixed_batch_slot_plan():
    """envs/mixed.py: plan_slots -- which launches of a mixed step share a stream, from the time each takes alone (a fork / join
    pair costs ~20 us per side stream): the longest alone on the current stream, the others packed behind each other as long
    as they stay inside it, never more slots than hardware queues"""
    from gym_anm_amd.envs.mixed import plan_slots

    assert plan_slots([87, 72, 48, 37]) == [[0], [1], [2, 3]]
    assert plan_slots([87.0, 87.0]) == [[0], [1]]
    assert plan_slots([500, 480, 470, 30, 20, 10]) == [[0, 5], [1, 4], [2, 3]]
    slots = plan_slots([50, 40, 30, 20, 10], max_slots=2)
    assert len(slots) == 2 and sorted(i for s in slots for i in s) == [0, 1, 2, 3, 4] and slots[0][0] == 0
    assert plan_slots([10]) == [[0]] and plan_slots([]) == [[]]
    for n in range(1, 9):   # every launch exactly once, at most four slots, slot 0 starts with the longest
        d = [((7 * i) % 11 + 1) * 10.0 for i in range(n)]
        slots = plan_slots(d)
        assert len(slots) <= 4 and sorted(i for s in slots for i in s) == list(range(n)) and d[slots[0][0]] == max(d)
