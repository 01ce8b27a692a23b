"""One configuration of the general lane-group kernel for profiling: mesh30, 16384 environments, cap from argv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_case30 as B
from gym_anm_amd import networks
import gym_anm_amd.simulator as S
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 8
orig = S.BatchedSimulator.__init__
def init(self, *a, _o=orig, **k):
    k.setdefault("max_iter", cap); k.setdefault("tol", 1e-6)
    _o(self, *a, **k)
S.BatchedSimulator.__init__ = init
B.run("mesh30", networks.synthetic_meshed_network(30, 6, 4), 16384, "mesh", n=10)
