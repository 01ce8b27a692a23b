"""The C ABI is usable from plain C: examples/transition_from_c.c is compiled with gcc against
include/anm_mi355x.h and the 2-bus library (CPU tier: builds and fails cleanly without a GPU;
GPU tier: its output is checked against the oracle)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from gym_anm_amd import codegen, networks  # noqa: E402
from gym_anm_amd.model import NetworkModel  # noqa: E402


def build_example(tmp_path):
    lib = codegen.build_library(NetworkModel(networks.two_bus_network(), 0.25, 100).topology())
    exe = str(tmp_path / "transition_from_c")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", rocm + "/include",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "transition_from_c.c"), lib,
           "-L", rocm + "/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath," + rocm + "/lib",
           "-o", exe]  # fmt: skip
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def test_c_example_builds_and_needs_a_gpu(tmp_path):
    import torch

    exe = build_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-tier test")
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 1 and "no GPU visible" in res.stderr  # loud failure, no CPU path


@pytest.mark.gpu
def test_c_example_matches_oracle(tmp_path):
    import anm_oracle as O

    exe = build_example(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    rows = re.findall(r"env (\d) load (\S+) MW converged (\d) iterations (\d+) \|V1\| (\S+) theta1 (\S+) slack_p (\S+) "
                      r"reward (\S+)", res.stdout)  # fmt: skip
    assert len(rows) == 4
    net = O.parse_network(networks.two_bus_network(), 0.25, 100)
    z = np.zeros(0)
    for _, load, conv, iters, vm, th, sp, rew in rows:
        ref = O.transition(net, np.array([float(load)]), z, z, z, z, tol=1e-8)
        assert int(conv) == int(ref["converged"]) and int(iters) == ref["n_iter"]
        if ref["converged"]:
            assert abs(float(vm) - abs(ref["V"][1])) < 1e-9
            assert abs(float(th) - np.angle(ref["V"][1])) < 1e-9
            assert abs(float(sp) - ref["dev_p"][0]) < 1e-9
            assert abs(float(rew) - ref["reward"]) < 1e-9 * max(1.0, abs(ref["reward"]))
    assert [int(r[2]) for r in rows] == [1, 1, 1, 0]  # 5 MW over this line has no power-flow solution
