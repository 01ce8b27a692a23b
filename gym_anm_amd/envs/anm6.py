"""The ANM6 family: the 6-bus case and the ``ANM6Easy-v0`` task, batched and single.

Restates ``gym_anm/envs/anm6_env/anm6.py`` (date bookkeeping only; the web renderer is out of
scope) and ``gym_anm/envs/anm6_env/anm6_easy.py`` (task constants :12-19, ``init_state`` :25-52,
``next_vars`` :54-65, the fixed daily series :77-132).

* :class:`ANM6EasyVec` -- ``num_envs`` copies on one MI355X.  The daily series live in HBM and the
  ``next_vars`` lookup is fused into the step kernel ("series mode"), so a step is one launch.
* the reference's single-environment NumPy-facing ``ANM6`` / ``ANM6Easy`` live in ``single.py``
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
