"""Meshed 30-bus network through the general lane-group kernel with Newton caps 100 / 20 / 8: separates the cost
of the handful of diverging solves (one serial chain of trips) from the cost of the batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_case30 as B
from gym_anm_amd import networks
import gym_anm_amd.simulator as S
net = networks.synthetic_meshed_network(30, 6, 4)
orig = S.BatchedSimulator.__init__
for cap in (100, 20, 8):
    def init(self, *a, _o=orig, **k):
        k.setdefault("max_iter", cap); k.setdefault("tol", 1e-6)
        _o(self, *a, **k)
    S.BatchedSimulator.__init__ = init
    print("cap", cap)
    for E in (16384, 65536):
        B.run("mesh30", net, E, "mesh", n=10)
S.BatchedSimulator.__init__ = orig
