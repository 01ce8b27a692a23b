"""The ANM6 family: the 6-bus case and the ``ANM6Easy-v0`` task, batched and single.

Restates ``gym_anm/envs/anm6_env/anm6.py`` (date bookkeeping only; the web renderer is out of
scope) and ``gym_anm/envs/anm6_env/anm6_easy.py`` (task constants :12-19, ``init_state`` :25-52,
``next_vars`` :54-65, the fixed daily series :77-132).

* :class:`ANM6EasyVec` -- ``num_envs`` copies on one MI355X.  The daily series live in HBM and the
  ``next_vars`` lookup is fused into the step kernel ("series mode"), so a step is one launch.
* :class:`ANM6Easy` -- the reference's single-environment NumPy-facing surface
  (``reset() -> (ndarray(18,), {})``, ``step(ndarray(6,)) -> (ndarray(18,), float, bool, False, {})``)
  on top of a 1-environment batch; its ``reset(seed=...)`` consumes the NumPy generator exactly
  like the reference does, so seeded episodes are reproducible against it.
"""

from __future__ import annotations

import datetime as dt

import numpy as np
import torch

from .. import networks
from .anm_env import BatchedANMEnv


def _daily_series(s1, ramp12, s2, ramp23, s3):
    """One 96-step day (anm6_easy.py:77-132): plateau s1 (25 steps), ramp, plateau s2 (13), ramp,
    plateau s3 (13), then the mirror image, closed by 4 more steps of s1."""
    r12 = np.linspace(ramp12[0], ramp12[1], 7)
    r23 = np.linspace(ramp23[0], ramp23[1], 7)
    a, b, c = s1 * np.ones(25), s2 * np.ones(13), s3 * np.ones(13)
    return np.concatenate((a, r12, b, r23, c, r23[::-1], b, r12[::-1], a[:4]))


def anm6easy_series() -> np.ndarray:
    """[5, 96] MW: loads (devices 1, 3, 5) then generator potentials (devices 2, 4)."""
    rows = [
        _daily_series(-1, (-1.5, -4.5), -5, (-4.625, -2.375), -2),  # residential load
        _daily_series(-4, (-4.75, -9.25), -10, (-11.25, -18.75), -20),  # industrial load
        _daily_series(0, (-3.125, -21.875), -25, (-21.875, -3.125), 0),  # EV charging
        _daily_series(0, (0.5, 3.5), 4, (7.25, 36.75), 30),  # PV
        _daily_series(40, (36.375, 14.625), 11, (14.725, 36.375), 40),  # wind
    ]
    out = np.vstack(rows)
    assert out.shape == (5, 96)
    return out


def random_date(np_random, year):
    """anm6_env/utils.py:5-23."""
    return dt.datetime(year, 1, 1) + dt.timedelta(days=float(np_random.integers(1, 365)))


class ANM6Vec(BatchedANMEnv):
    """Batched ``ANM6``: the 6-bus network behind the ``ANMEnv`` surface (anm6.py:13-141)."""

    metadata = {"render_modes": []}

    def __init__(self, observation, K, delta_t, gamma, lamb, aux_bounds=None, costs_clipping=(None, None), seed=None,
                 **kw):  # fmt: skip
        super().__init__(networks.anm6_network(), observation, K, delta_t, gamma, lamb, aux_bounds, costs_clipping,
                         seed, **kw)  # fmt: skip
        self.network_specs = self.simulator.get_rendering_specs()
        self.timestep_length = dt.timedelta(minutes=int(60 * delta_t))


class ANM6EasyVec(ANM6Vec):
    """``num_envs`` ANM6Easy-v0 environments stepped by one kernel launch."""

    def __init__(self, num_envs=1, device="cuda", seed=None, **kw):
        delta_t = 0.25
        self.P_loads = anm6easy_series()[:3]
        self.P_maxs = anm6easy_series()[3:]
        super().__init__(
            "state", 1, delta_t, 0.995, 100, aux_bounds=np.array([[0, 24 / delta_t - 1]]), costs_clipping=(1, 100),
            seed=seed, num_envs=num_envs, device=device, series=anm6easy_series(), **kw,
        )  # fmt: skip

    def init_state(self):
        """[num_envs, 18] initial states (anm6_easy.py:25-52).  Environment 0 consumes ``np_random``
        in the reference's order (t_0, q of device 2, q of device 4, SoC of device 6); the other
        environments are drawn in bulk from the same generator afterwards."""
        n_dev, n_gen, n_des = 7, 2, 1
        E_ = self.num_envs
        m = self.simulator.model
        rng = self.np_random
        state = np.zeros((E_, 2 * n_dev + n_des + n_gen + self.K))
        t0 = np.zeros(E_, dtype=np.int64)
        q = np.zeros((E_, 2))
        soc = np.zeros(E_)
        period = int(24 / self.delta_t)
        t0[0] = rng.integers(0, period)
        for j, dev in enumerate((2, 4)):
            k = m.dev_ids.index(dev)
            q[0, j] = rng.uniform(m.dev_q_min[k], m.dev_q_max[k])
        k6 = m.dev_ids.index(6)
        soc[0] = rng.uniform(m.dev_soc_min[k6], m.dev_soc_max[k6])
        if E_ > 1:
            t0[1:] = rng.integers(0, period, size=E_ - 1)
            for j, dev in enumerate((2, 4)):
                k = m.dev_ids.index(dev)
                q[1:, j] = rng.uniform(m.dev_q_min[k], m.dev_q_max[k], size=E_ - 1)
            soc[1:] = rng.uniform(m.dev_soc_min[k6], m.dev_soc_max[k6], size=E_ - 1)
        state[:, -1] = t0
        for dev, series in zip((1, 3, 5), self.P_loads):
            k = m.dev_ids.index(dev)
            state[:, dev] = series[t0]
            state[:, n_dev + dev] = series[t0] * m.dev_qp[k]
        for j, (dev, series) in enumerate(zip((2, 4), self.P_maxs)):
            state[:, 2 * n_dev + n_des + j] = series[t0]
            state[:, dev] = series[t0]
            state[:, n_dev + dev] = q[:, j]  # drawn in p.u., stored in the MVAr slot (reference quirk)
        state[:, 2 * n_dev] = soc  # drawn in p.u., stored in the MWh slot (reference quirk)
        return state

    def next_vars(self, s_t):
        """Host restatement of the fused lookup (anm6_easy.py:54-65); the kernel does this itself."""
        s_t = torch.as_tensor(s_t)
        aux = torch.remainder(s_t[:, -1] + 1, 24 / self.delta_t).long()
        tab = torch.as_tensor(anm6easy_series(), dtype=torch.float64, device=s_t.device)
        return torch.cat([tab[:, aux].T, aux.unsqueeze(1).to(torch.float64)], dim=1)


class ANM6Easy:
    """Single-environment ``ANM6Easy-v0`` with the reference's NumPy-facing contract."""

    metadata = {"render_modes": []}

    def __init__(self, device="cuda", **kw):
        kw.setdefault("track_full", True)  # `simulator.state` follows every step, as in the reference
        self.vec = ANM6EasyVec(num_envs=1, device=device, **kw)
        v = self.vec
        self.action_space, self.observation_space = v.action_space, v.observation_space
        self.K, self.gamma, self.lamb, self.delta_t = v.K, v.gamma, v.lamb, v.delta_t
        self.costs_clipping = v.costs_clipping
        self.simulator = v.simulator
        self.state_N, self.observation_N = v.state_N, v.observation_N
        self.timestep_length = v.timestep_length
        self.date = self.date_init = None
        self.year_count = 0
        self.render_mode = None
        self.terminated = False
        self.timestep = 0
        self.e_loss = self.penalty = 0.0
        self.state = None

    @property
    def np_random(self):
        return self.vec.np_random

    def _sync_attrs(self):
        v = self.vec
        self.state = v.state[0].cpu().numpy()
        self.terminated = bool(v.terminated[0])
        self.timestep = int(v.timestep[0])
        self.e_loss = float(v.e_loss[0])
        self.penalty = float(v.penalty[0])

    def reset(self, *, seed=None, options=None):
        opts = dict(options or {})
        date_init = opts.pop("date_init", None)
        obs, info = self.vec.reset(seed=seed, options=opts or None)
        self._sync_attrs()
        # ANM6.reset (anm6.py:124-141) then ANM6Easy.reset (anm6_easy.py:67-74)
        self.year_count = 0
        self.date_init = date_init if date_init is not None else random_date(self.vec.np_random, 2020)
        self.date = self.date_init
        self.date_init = self.date = self.date + self.state[-1] * self.timestep_length
        o = obs[0].cpu().numpy()
        assert self.observation_space.contains(o), "Observation %r (%s) invalid." % (o, type(o))
        return o, info

    def step(self, action):
        action = np.asarray(action, dtype=np.float64)
        assert self.action_space.contains(action), "Action %r (%s) invalid." % (action, type(action))
        if self.terminated:
            return np.zeros(self.observation_N), 0.0, True, False, {}
        self.vec.check_actions = False
        obs, r, term, trunc, info = self.vec.step(torch.as_tensor(action).unsqueeze(0))
        self._sync_attrs()
        self.date += self.timestep_length  # anm6.py:113-122
        self.year_count = (self.date - self.date_init).days // 365
        return obs[0].cpu().numpy(), float(r[0]), bool(term[0]), False, info

    def render(self, mode="human", skip_frames=0):
        raise NotImplementedError("the web renderer of the reference is out of scope for this build")

    def close(self):
        pass
