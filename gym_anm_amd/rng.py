"""Host restatement of the device-side counter-based RNG (Philox4x32-10, csrc/anm_device.hpp).

The reference samples initial states on the host with NumPy's PCG64 (``anm6_easy.py:25-52``); for
in-kernel autoreset of tens of thousands of environments this build draws them on the device
instead, keyed by ``(seed, environment index, reset count)`` so that any environment's stream can
be reproduced independently.  This module is the specification the kernel is tested against
(bit-exact integers, identical doubles).
"""

from __future__ import annotations

import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32(seed: int, env: int, epoch: int, draw: int):
    c = [env & MASK, (env >> 32) & MASK, epoch & MASK, draw & MASK]
    k0, k1 = seed & MASK, (seed >> 32) & MASK
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c


def u01(hi: int, lo: int) -> float:
    return float(((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0)


def series_init_state(model, series, seed, env, epoch):
    """Initial state the step kernel draws for environment ``env`` at its ``epoch``-th autoreset
    (series-mode tasks, layout and quirks of ``ANM6Easy.init_state``)."""
    period = series.shape[1]
    D, nd, ng = model.N_device, model.N_des, model.N_non_slack_gen
    s0 = np.zeros(2 * D + nd + ng + 1)
    r = philox4x32(seed, env, epoch, 0)
    aux = (r[0] * period) >> 32
    s0[-1] = aux

    def uniform(u):
        q = philox4x32(seed, env, epoch, 1 + u // 2)
        return u01(q[2 * (u % 2)], q[2 * (u % 2) + 1])

    for s, k in enumerate(model.load_idx):
        s0[k] = series[s, aux]
    for g, k in enumerate(model.gen_idx):
        pm = series[model.N_load + g, aux]
        s0[k] = pm
        s0[2 * D + nd + g] = pm
        s0[D + k] = model.dev_q_min[k] + (model.dev_q_max[k] - model.dev_q_min[k]) * uniform(g)
    for e, k in enumerate(model.des_idx):
        s0[2 * D + e] = model.dev_soc_min[k] + (model.dev_soc_max[k] - model.dev_soc_min[k]) * uniform(ng + e)
    return s0
