"""``ANMEnv`` with the reference's single-environment, NumPy-facing contract (gym_anm/envs/anm_env.py:19-592).

Subclass it exactly like ``gym_anm.ANMEnv``::

    class MyEnv(ANMEnv):
        def __init__(self):
            super().__init__(network, observation, K, delta_t, gamma, lamb, aux_bounds, costs_clipping, seed)
        def init_state(self): ...        # -> 1-D state vector            (anm_env.py:158-174)
        def next_vars(self, s_t): ...    # 1-D state -> 1-D [P_load.., P_pot.., aux..]   (anm_env.py:176-191)
        def observation_bounds(self): ...  # optional                      (anm_env.py:193-233)

``reset()`` / ``step()`` take and return 1-D NumPy arrays, a float reward and Python bools, and raise what the
reference raises.  Underneath runs a one-environment :class:`BatchedANMEnv` on the GPU (``device``, ``**kw`` are
the only additions to the reference's constructor); for throughput use :class:`BatchedANMEnv` directly."""

from __future__ import annotations

import numpy as np
import torch

from ..spaces import GymEnv
from .anm_env import BatchedANMEnv


class ANMEnv(GymEnv):  # gymnasium.Env when gymnasium is installed (wrappers, gym.make), a minimal stand-in otherwise
    metadata = {"render_modes": []}

    def __init__(self, network, observation, K, delta_t, gamma, lamb, aux_bounds=None, costs_clipping=(None, None),
                 seed=None, device="cuda", **kw):  # fmt: skip
        outer = self
        user_bounds = type(self).observation_bounds is not ANMEnv.observation_bounds
        obs_fn = observation if callable(observation) else None

        class _Vec(BatchedANMEnv):
            def init_state(self):
                return np.asarray(outer.init_state(), dtype=np.float64).reshape(1, -1)

            def next_vars(self, s_t):
                v = np.asarray(outer.next_vars(torch.as_tensor(s_t)[0].cpu().numpy()), dtype=np.float64)
                return v.reshape(1, -1)

            def observation_bounds(self):
                if user_bounds:
                    # what the reference's hook may look at exists at this point of its constructor (anm_env.py:124-139)
                    for name in ("simulator", "obs_values", "state_values", "state_N", "action_space"):
                        if hasattr(self, name):
                            setattr(outer, name, getattr(self, name))
                    return outer.observation_bounds()
                return BatchedANMEnv.observation_bounds(self)

        def batched_obs(s_t):  # callable observation: 1-D state -> 1-D vector (anm_env.py:511-514)
            o = np.asarray(obs_fn(torch.as_tensor(s_t)[0].cpu().numpy()), dtype=np.float64)
            return torch.as_tensor(o.reshape(1, -1), device=s_t.device if hasattr(s_t, "device") else None)

        # the attributes a constructor hook of the user (observation_bounds) may read exist before the batched
        # environment calls it
        self.K, self.gamma, self.lamb, self.delta_t = K, gamma, lamb, delta_t
        # (anm_env.py:122-130: aux_bounds and costs_clipping exist before observation_bounds() is called -- the
        # reference's own default implementation reads aux_bounds)
        self.aux_bounds = aux_bounds
        self.costs_clipping = costs_clipping
        kw.setdefault("track_full", True)  # `simulator.state` / `.devices[i].p` follow every step, as in the reference
        self.vec = self._make_vec(_Vec, network, batched_obs if obs_fn is not None else observation, K, delta_t, gamma,
                                  lamb, aux_bounds, costs_clipping, seed, device, kw)
        v = self.vec
        self.simulator = v.simulator
        self.action_space, self.observation_space = v.action_space, v.observation_space
        self.state_values, self.obs_values = v.state_values, v.obs_values
        self.state_N, self.observation_N = v.state_N, getattr(v, "observation_N", None)
        self.costs_clipping = v.costs_clipping
        self.timestep_length = getattr(v, "timestep_length", None)
        self.render_mode = None
        self.terminated = False
        self.timestep = 0
        self.e_loss = self.penalty = 0.0
        self.state = None
        self.pfe_converged = None

    def _make_vec(self, vec_cls, network, observation, K, delta_t, gamma, lamb, aux_bounds, costs_clipping, seed, device, kw):
        """The one-environment batched environment underneath (a subclass with a fused task overrides this)."""
        return vec_cls(network, observation, K, delta_t, gamma, lamb, aux_bounds, costs_clipping, seed, num_envs=1,
                       device=device, **kw)

    # ---- hooks (anm_env.py:158-233) ----------------------------------------------------------------------------
    def init_state(self):
        raise NotImplementedError

    def next_vars(self, s_t):
        raise NotImplementedError

    def observation_bounds(self):
        return BatchedANMEnv.observation_bounds(self.vec)

    def observation(self, s_t):
        o = self.vec.observation(torch.as_tensor(np.asarray(s_t, dtype=np.float64).reshape(1, -1), device=self.vec.device))
        return o[0].cpu().numpy()

    @property
    def np_random(self):
        return self.vec.np_random

    def _sync(self):
        v = self.vec
        self.state = v.state[0].cpu().numpy()
        self.terminated = bool(v.terminated[0])
        self.timestep = int(v.timestep[0])
        self.e_loss = float(v.e_loss[0])
        self.penalty = float(v.penalty[0])
        self.pfe_converged = bool(v.pfe_converged[0]) if v.pfe_converged is not None else None
        self.observation_space = v.observation_space
        self.observation_N = getattr(v, "observation_N", self.observation_N)

    # ---- gym.Env surface (anm_env.py:235-453) ---------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        obs, info = self.vec.reset(seed=seed, options=options)
        self._sync()
        o = obs[0].cpu().numpy()
        assert self.observation_space.contains(o), "Observation %r (%s) invalid." % (o, type(o))
        return o, info

    def step(self, action):
        action = np.asarray(action, dtype=np.float64)
        assert self.action_space.contains(action), "Action %r (%s) invalid." % (action, type(action))
        if self.terminated:  # absorbing terminal state (anm_env.py:365-367)
            return np.zeros(self.observation_N), 0.0, True, False, {}
        self.vec.check_actions = False
        obs, r, term, trunc, info = self.vec.step(torch.as_tensor(action).unsqueeze(0))
        self._sync()
        return obs[0].cpu().numpy(), float(r[0]), bool(term[0]), False, info

    def render(self, mode="human"):
        raise NotImplementedError("the web renderer of the reference is out of scope for this build")

    def close(self):
        pass


class ANM6(ANMEnv):
    """``gym_anm.envs.ANM6`` (anm6_env/anm6.py:13-141): the 6-bus network behind the single-environment surface, with
    the date bookkeeping of the reference (the web rendering it serves is out of scope)."""

    def __init__(self, observation, K, delta_t, gamma, lamb, aux_bounds=None, costs_clipping=(None, None), seed=None,
                 **kw):  # fmt: skip
        import datetime as dt

        from .. import networks

        super().__init__(networks.anm6_network(), observation, K, delta_t, gamma, lamb, aux_bounds, costs_clipping, seed, **kw)
        self.timestep_length = dt.timedelta(minutes=int(60 * delta_t))
        self.date = self.date_init = None
        self.year_count = 0

    def reset(self, *, seed=None, options=None):
        from .anm6 import random_date

        opts = dict(options or {})
        date_init = opts.pop("date_init", None)
        obs, info = super().reset(seed=seed, options=opts or None)
        self.year_count = 0  # anm6.py:124-141
        self.date_init = date_init if date_init is not None else random_date(self.np_random, 2020)
        self.date = self.date_init
        return obs, info

    def step(self, action):
        out = super().step(action)
        if self.date is not None:
            self.date += self.timestep_length  # anm6.py:113-122
            self.year_count = (self.date - self.date_init).days // 365
        return out


class ANM6Easy(ANM6):
    """``ANM6Easy-v0`` (anm6_env/anm6_easy.py): the reference's single-environment, NumPy-facing task.  Underneath runs
    :class:`ANM6EasyVec` with the daily series fused into the step kernel; ``init_state`` / ``next_vars`` are the
    batched class's (environment 0 consumes ``np_random`` in the reference's order)."""

    def __init__(self, device="cuda", **kw):
        self._kw = kw
        super().__init__("state", 1, 0.25, 0.995, 100, np.array([[0, 24 / 0.25 - 1]]), (1, 100), kw.pop("seed", None),
                         device=device, **kw)
        self.P_loads, self.P_maxs = self.vec.P_loads, self.vec.P_maxs

    def _make_vec(self, vec_cls, network, observation, K, delta_t, gamma, lamb, aux_bounds, costs_clipping, seed, device, kw):
        from .anm6 import ANM6EasyVec

        return ANM6EasyVec(num_envs=1, device=device, seed=seed, **kw)

    def init_state(self):
        return self.vec.init_state()[0]

    def next_vars(self, s_t):
        return self.vec.next_vars(np.asarray(s_t, dtype=np.float64).reshape(1, -1))[0].cpu().numpy()

    def reset(self, *, seed=None, options=None):
        obs, info = super().reset(seed=seed, options=options)
        # anm6_easy.py:67-74: the displayed date starts at the sampled time of day
        self.date_init = self.date = self.date + self.state[-1] * self.timestep_length
        return obs, info
