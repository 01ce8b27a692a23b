"""Golden vectors of the MPC DC-OPF policies, recorded from the UNMODIFIED reference agents.

TEST INFRASTRUCTURE ONLY (dev container: needs /root/reference; see oracle/ref_harness.py).

    python oracle/make_golden_mpc.py        # (re)writes tests/golden/mpc_anm6.npz

``gym_anm/agents/mpc.py`` states the N-stage DC-OPF with cvxpy's modelling operators and calls
``Problem.solve()``.  cvxpy is not in this image; the harness stands in for the *modelling API only* and hands
the linear program the reference's code builds to scipy's HiGHS (``ref_harness.solve_lp``).  The reference's own
tests of this path (``tests/test_dcopf_agent.py``: 3 horizons x 1000 closed-loop steps, constraints C1-C7 to
1e-5) pass under the stand-in -- ``--reference-tests`` runs them.  Recorded per case: what the agent saw (state,
forecasts, SoC), the optimal VALUE of the program (solver-independent), its status, the first-stage minimiser
HiGHS returned (one of possibly several) and the action ``act()`` returned.  Configurations follow the
reference's examples (``examples/mpc_perfect.py``, ``mpc_constant.py``: safety margin 0.96, N = 10) and tests
(N = 1, 3, 20).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.load_reference()
from gym_anm import MPCAgentConstant, MPCAgentPerfect  # noqa: E402
from gym_anm.envs import ANM6Easy  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
CONFIGS = [("perfect", 1, 0.96), ("perfect", 3, 0.96), ("perfect", 10, 0.96), ("constant", 3, 0.9), ("constant", 20, 0.9)]


def record(kind, N, margin, seed, T):
    env = ANM6Easy()
    env.reset(seed=seed)
    cls = MPCAgentPerfect if kind == "perfect" else MPCAgentConstant
    agent = cls(env.simulator, env.action_space, env.gamma, safety_margin=margin, planning_steps=N)
    rng = np.random.default_rng(100 + seed)
    out = {k: [] for k in ["state", "load", "gen", "soc", "objective", "p_dev", "theta", "action"]}
    for t in range(T):
        Pl, Pg = agent.forecast(env)
        raw = agent._solve(env.simulator, Pl, Pg)
        assert agent.dc_opf.status == "optimal"
        a = np.clip(raw, env.action_space.low, env.action_space.high)  # mpc.py:338-341
        out["state"].append(env.state.copy())
        out["load"].append(np.array(Pl, dtype=np.float64))
        out["gen"].append(np.array(Pg, dtype=np.float64))
        out["soc"].append(np.array(agent.init_soc.value, dtype=np.float64))
        out["objective"].append(agent.dc_opf.value)
        out["p_dev"].append(np.array(agent.P_dev.value))
        out["theta"].append(np.array(agent.V_bus_ang.value))
        out["action"].append(a)
        if t % 3 == 2:  # every third step somebody else drives: the storage unit leaves the MPC's own trajectory
            a = rng.uniform(env.action_space.low, env.action_space.high)
        _, _, term, _, _ = env.step(a)
        if term:
            env.reset()
    return {k: np.array(v) for k, v in out.items()}


def main():
    if "--reference-tests" in sys.argv:
        import unittest

        import tests.test_dcopf_agent as T  # the reference's, from /root/reference

        res = unittest.TextTestRunner(verbosity=2).run(unittest.defaultTestLoader.loadTestsFromModule(T))
        sys.exit(0 if res.wasSuccessful() else 1)
    out = {"kind": np.array([c[0] for c in CONFIGS]), "N": np.array([c[1] for c in CONFIGS]),
           "safety_margin": np.array([c[2] for c in CONFIGS]), "gamma": np.array(ANM6Easy().gamma)}
    for k, (kind, N, margin) in enumerate(CONFIGS):
        rec = record(kind, N, margin, seed=40 + k, T=60)
        for name, v in rec.items():
            out["c%d_%s" % (k, name)] = v
        print("%-8s N=%-2d margin=%.2f  cases=%d  objective in [%.4f, %.4f]" % (kind, N, margin, len(rec["objective"]),
                                                                                 rec["objective"].min(), rec["objective"].max()))
    np.savez_compressed(os.path.join(GOLDEN, "mpc_anm6.npz"), **out)


if __name__ == "__main__":
    main()
