"""Golden sample of the reference's initial-state distribution: tests/golden/init_state_anm6easy.npz.

TEST INFRASTRUCTURE ONLY (dev container; needs /root/reference).  Usage::

    python oracle/make_golden_init_state.py

10 000 calls of the UNMODIFIED ``ANM6Easy.init_state()`` (``gym_anm/envs/anm6_env/anm6_easy.py:25-52``) after
``reset(seed=2024)`` -- no power flow involved: ``init_state`` only draws.  Recorded per draw: the time index ``t_0``, the
generators' reactive power (devices 2 and 4; the reference draws them in the devices' p.u. range and stores them in the
MVAr slot), the storage unit's state of charge (device 6; p.u. range, MWh slot), and -- to pin the table look-up -- the loads'
and generators' active power at ``t_0``.  The device-side sampler (Philox4x32-10, csrc/anm_device.hpp + anm_env_ops.hpp:
sample_series_init_state) cannot reproduce NumPy's PCG64 stream draw for draw; the GPU tier tests that its draws have THIS
distribution (two-sample Kolmogorov-Smirnov / chi-square on the marginals, tests/test_gpu_sampler.py) and that the
deterministic part (P at t_0) is the same function of t_0.
"""

from __future__ import annotations

import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_harness  # noqa: E402

ref_harness.load_reference()

from gym_anm.envs import ANM6Easy  # noqa: E402

warnings.simplefilter("ignore")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def main(n=10000, seed=2024):
    env = ANM6Easy()
    env.reset(seed=seed)
    S = np.array([env.init_state() for _ in range(n)])
    out = dict(
        seed=seed, t0=S[:, -1].astype(np.int64), gen_q=S[:, [7 + 2, 7 + 4]].copy(), des_soc=S[:, 14].copy(),
        load_p=S[:, [1, 3, 5]].copy(), gen_p=S[:, [2, 4]].copy(), gen_p_max=S[:, [15, 16]].copy(),
        q_range=np.array([[env.simulator.devices[d].q_min, env.simulator.devices[d].q_max] for d in (2, 4)]),
        soc_range=np.array([env.simulator.devices[6].soc_min, env.simulator.devices[6].soc_max]),
    )
    np.savez_compressed(os.path.join(GOLDEN, "init_state_anm6easy.npz"), **out)
    print("init_state_anm6easy.npz: %d draws; t0 in [%d, %d], q ranges %s, soc range %s"
          % (n, out["t0"].min(), out["t0"].max(), out["q_range"].tolist(), out["soc_range"].tolist()))


if __name__ == "__main__":
    main()
