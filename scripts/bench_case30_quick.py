"""case30 (BASELINE.json config 4), 16384 environments, Simulator.transition: lane-group family with the
reference cap and with a cap of 20; set ANM_RADIAL_GENERIC=1 for the table-driven Newton loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_case30 import run, DEV
from gym_anm_amd import networks
import gym_anm_amd.simulator as S
n30 = networks.synthetic_radial_network(30, 0)
for cap in (100, 20):
    orig = S.BatchedSimulator.__init__
    def init(self, *a, _o=orig, **k):
        k.setdefault("max_iter", cap); k.setdefault("tol", 1e-6)
        _o(self, *a, **k)
    S.BatchedSimulator.__init__ = init
    print("cap", cap, "generic" if os.environ.get("ANM_RADIAL_GENERIC") else "specialised")
    run("case30", n30, 16384, "radial")
    run("case30", n30, 16384, "radial", full=False)
    S.BatchedSimulator.__init__ = orig
