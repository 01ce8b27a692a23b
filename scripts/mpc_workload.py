"""Workload for profiling k_mpc: 65 536 ANM6Easy programs, perfect forecasts, N stages.

    python scripts/mpc_workload.py 1 [solves]      # N = 1: k_mpc<Topo, true>; N > 1: k_mpc<Topo, false>
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_anm_amd.agents import MPCAgentPerfect
from gym_anm_amd.envs import ANM6EasyVec

DEV = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
E = 65536
env = ANM6EasyVec(num_envs=E, device=DEV, seed=3)
env.reset(seed=3)
ag = MPCAgentPerfect(env.simulator, env.action_space, env.gamma, safety_margin=0.92, planning_steps=N)
pl, pg = ag.forecast(env)
soc = ag._soc(env)
for _ in range(n):
    ag.solver.solve(pl, pg, soc)
torch.cuda.synchronize()
it = ag.solver.iters.double()
print("N %d: iterations mean %.2f max %d; rows per stage %d" % (N, float(it.mean()), int(it.max()), ag.solver.dims.n_stage_rows))
