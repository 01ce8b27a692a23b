"""NumPy-facing vector-environment adapter.

RL libraries that drive ``gymnasium.vector.VectorEnv``-style environments from the host (NumPy actions
in, NumPy observations out) can wrap any batched environment of this package with it.  The simulator
step itself stays on the GPU; the adapter adds one host-to-device copy of the actions and one
device-to-host copy of (obs, reward, terminated) per step, i.e. the PCIe round trip DESIGN.md prices.
Trainers that keep their policy on the GPU should use the batched environment directly.

The reference has no vector environment (one ``ANMEnv`` object per environment, anm_env.py:17-156); the
attribute names below are Gymnasium's (``num_envs``, ``single_action_space``,
``single_observation_space``), the autoreset convention is Gymnasium's "next-step" mode, which is what
the in-kernel autoreset implements.
"""

from __future__ import annotations

import numpy as np
import torch


class NumpyVectorEnv:
    metadata = {"autoreset_mode": "next_step", "render_modes": []}

    def __init__(self, env):
        if not getattr(env, "autoreset", False):
            raise ValueError("NumpyVectorEnv needs an environment built with autoreset=True")
        self.env = env
        self.num_envs = env.num_envs
        self.single_action_space = env.action_space
        self.single_observation_space = env.observation_space
        self.action_space = env.action_space          # per-environment Box; actions are [num_envs, A]
        self.observation_space = env.observation_space
        self.render_mode = None
        env.check_actions = False                     # checked here on the host, where the data already is
        self._pinned = None

    def reset(self, *, seed=None, options=None):
        obs, info = self.env.reset(seed=seed, options=options)
        return obs.cpu().numpy(), info

    def step(self, actions):
        actions = np.asarray(actions, dtype=np.float64)
        lo, hi = self.single_action_space.low, self.single_action_space.high
        if actions.shape != (self.num_envs, lo.shape[0]) or not ((actions >= lo) & (actions <= hi)).all():
            raise AssertionError("Action %r (%s) invalid." % (actions, type(actions)))  # anm_env.py:356-357
        obs, rew, term, trunc, info = self.env.step(torch.from_numpy(actions).to(self.env.device))
        return (obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy(), trunc.cpu().numpy(), info)

    def close(self):
        self.env.close()
