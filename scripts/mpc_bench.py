"""Batched MPC DC-OPF policy on ANM6Easy: time of one solve of all environments (one launch of k_mpc: HIP events on the
launch stream and wall clock around act()), interior-point iterations, closed-loop return.

    python scripts/mpc_bench.py [--quick]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_anm_amd.agents import MPCAgentConstant, MPCAgentPerfect
from gym_anm_amd.envs import ANM6EasyVec

DEV = "cuda:0"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
CASES = [(MPCAgentConstant, 1, 65536), (MPCAgentPerfect, 1, 65536), (MPCAgentPerfect, 4, 65536), (MPCAgentPerfect, 10, 65536),
         (MPCAgentConstant, 20, 65536), (MPCAgentPerfect, 10, 4096), (MPCAgentPerfect, 64, 16384)]
if "--quick" in sys.argv:
    CASES = CASES[:4]
for Agent, N, E in CASES:
    env = ANM6EasyVec(num_envs=E, device=DEV, seed=3)
    env.reset(seed=3)
    ag = Agent(env.simulator, env.action_space, env.gamma, safety_margin=0.92, planning_steps=N)
    tot = torch.zeros(E, dtype=torch.float64, device=DEV)
    wall, kern, its = [], [], []
    for s in range(8):
        pl, pg = ag.forecast(env)
        soc = ag._soc(env)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        ag.solver.solve(pl, pg, soc)      # (layout change of the forecasts + the launch)
        e1.record()
        torch.cuda.synchronize()
        kern.append(e0.elapsed_time(e1))
        torch.cuda.synchronize()
        t = time.perf_counter()
        a = ag.act(env)
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t)
        its.append(ag.solver.iters.double())
        bad = ~ag.last_converged
        if bool(bad.any()) and not os.path.exists(os.path.join(OUT, "mpc_unconverged_N%d.npz" % N)):
            import numpy as np

            os.makedirs(OUT, exist_ok=True)
            np.savez(os.path.join(OUT, "mpc_unconverged_N%d.npz" % N), pl=pl[bad].cpu().numpy(), pg=pg[bad].cpu().numpy(),
                     soc=soc[bad].cpu().numpy(), iters=ag.solver.iters[bad].cpu().numpy(), info=ag.solver.info[bad].cpu().numpy(),
                     objective=ag.solver.objective[bad].cpu().numpy())
            print("  %d environments did not converge (step %d): dumped" % (int(bad.sum()), s), flush=True)
        _, r, term, _, _ = env.step(a)
        tot += r
    it = torch.stack(its[2:])
    d = ag.solver.dims
    print("%-17s N=%2d E=%6d  %d rows x %d variables per stage  solve %.3f ms (events)  act() %.3f ms (wall)  iterations mean %.1f max %d"
          "  %.2e programs/s  return/step %.3f  terminated %d"
          % (Agent.__name__, N, E, d.n_stage_rows, d.n_stage_vars, min(kern[2:]), 1e3 * min(wall[2:]), float(it.mean()), int(it.max()),
             E / (1e-3 * min(kern[2:])), float(tot.mean()) / 8, int(env.terminated.sum())), flush=True)
