"""Step time by observation mode: "state" (fast-path kernel), a list-form observation gathered inside the
general step kernel ("list"), the same through the electrical-state dump + gather kernel ("list-unfused"), and
"state" with the dump of every step (track_full).  ANM6Easy, random agent."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec

import numpy as np
from gym_anm_amd.envs.anm6 import ANM6Vec, anm6easy_series


class ListObs(ANM6EasyVec):
    """ANM6Easy with a list-form observation instead of "state"."""

    def __init__(self, observation, num_envs=1, device="cuda", seed=None, **kw):
        self.P_loads = anm6easy_series()[:3]
        self.P_maxs = anm6easy_series()[3:]
        ANM6Vec.__init__(self, observation, 1, 0.25, 0.995, 100, aux_bounds=np.array([[0, 95]]), costs_clipping=(1, 100),
                         seed=seed, num_envs=num_envs, device=device, series=anm6easy_series(), **kw)

dev = torch.device("cuda", 0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for cap in (20, 100):
    for mode in ("state", "list", "list-soc-aux", "list-unfused", "state+dump"):
        kw = dict(num_envs=E, device=dev, seed=1, tol=1e-6, max_iter=cap, autoreset=True)
        if mode.startswith("list"):
            obs = [("bus_v_magn", "all", "pu"), ("branch_s", "all", "MVA"), ("des_soc", "all", "MWh"), ("aux", "all", None)]
            if mode == "list-soc-aux":  # nothing electrical: what the general step kernel costs by itself
                obs = obs[2:]
            env = ListObs(obs, fuse_observation=(mode != "list-unfused"), **kw)
        else:
            env = ANM6EasyVec(track_full=(mode == "state+dump"), **kw)
        env.check_actions = False
        env.reset(seed=1)
        g = torch.Generator(device=dev).manual_seed(0)
        lo = torch.as_tensor(env.action_space.low, device=dev); hi = torch.as_tensor(env.action_space.high, device=dev)
        pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=dev) for _ in range(8)]
        for i in range(10): env.step(pool[i % 8])
        torch.cuda.synchronize()
        t = time.perf_counter(); n = 100
        for i in range(n): env.step(pool[i % 8])
        torch.cuda.synchronize()
        print("E=%d cap=%3d obs=%-12s %.1f us/step" % (E, cap, mode, (time.perf_counter() - t) / n * 1e6), flush=True)


# ---- the lane-group families (round 5: anm_model_set_obs is accepted by them): config 4's 30-bus feeder, 16 384
# environments, host next_vars (exo handed over), cap 20 -- "state", a list gathered in the kernel, the same through dump + gather
from gym_anm_amd import networks
from gym_anm_amd.envs.anm_env import BatchedANMEnv

net30 = networks.synthetic_radial_network(30, 0)
E30 = 16384


class Feeder(BatchedANMEnv):
    def __init__(self, observation, impl, fuse):
        super().__init__(net30, observation, 1, 0.25, 0.995, 100, aux_bounds=np.array([[0, 95]]), num_envs=E30, device=dev, impl=impl,
                         fuse_observation=fuse, tol=1e-6, max_iter=20, seed=2)

    def next_vars(self, s):
        return self._vars


for impl in ("radial", "mesh"):
    for mode in ("state", "list", "list-unfused"):
        obs = "state" if mode == "state" else [("bus_v_magn", "all", "pu"), ("branch_s", "all", "MVA"), ("des_soc", "all", "MWh"), ("aux", "all")]
        env = Feeder(obs, impl, mode != "list-unfused")
        env.check_actions = False
        m, b = env.simulator.model, env.simulator.baseMVA
        g = torch.Generator(device=dev).manual_seed(3)
        U = lambda lo, hi: (torch.as_tensor(lo, device=dev) + (torch.as_tensor(hi, device=dev) - torch.as_tensor(lo, device=dev))
                            * torch.rand((E30, len(lo)), generator=g, dtype=torch.float64, device=dev))
        env._vars = torch.cat((U(m.dev_p_min[m.load_idx] * b, 0 * m.dev_p_min[m.load_idx]), U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * b),
                               torch.zeros((E30, 1), dtype=torch.float64, device=dev)), 1)
        env.reset(options={"init_state": torch.zeros((E30, env.state_N), dtype=torch.float64, device=dev)})
        act = U(env.action_space.low, env.action_space.high)
        for i in range(5):
            env._term_u8.zero_(); env.step(act)
        torch.cuda.synchronize()
        t = time.perf_counter(); n = 50
        for i in range(n):
            env._term_u8.zero_(); env.step(act)
        torch.cuda.synchronize()
        print("case30 E=%d cap= 20 impl=%-6s obs=%-12s %.1f us/step (incl. one memset)" % (E30, impl, mode, (time.perf_counter() - t) / n * 1e6), flush=True)
