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
