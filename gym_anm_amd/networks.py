"""Network input dictionaries (case files) used by the batched environments, tests and bench.

A network dictionary has the reference's format (``gym_anm/envs/anm6_env/network.py:49-82``,
column maps ``gym_anm/simulator/components/constants.py:1-27``):

* ``baseMVA``: float
* ``bus``    rows ``[BUS_ID, BUS_TYPE(0 slack / 1 PQ), BASE_KV, VMAX, VMIN]``
* ``device`` rows ``[DEV_ID, BUS_ID, DEV_TYPE(-1 load, 0 slack, 1 classical gen, 2 renewable,
  3 storage), Q/P, PMAX, PMIN, QMAX, QMIN, P+, P-, Q+, Q-, SOC_MAX, SOC_MIN, EFF]`` (MW / MVAr /
  MWh; ``None`` = unspecified)
* ``branch`` rows ``[F_BUS, T_BUS, BR_R, BR_X, BR_B, RATE(MVA), TAP, SHIFT(deg)]``
"""

from __future__ import annotations

import numpy as np

_N = None


def anm6_network() -> dict:
    """The 6-bus / 7-device / 5-branch ANM6 case (values: ``network.py:49-82`` of the reference)."""
    bus = [[0, 0, 132, 1.0, 1.0]] + [[i, 1, 33, 1.1, 0.9] for i in range(1, 6)]
    load = lambda d, b, pmin: [d, b, -1, 0.2, 0, pmin, _N, _N, _N, _N, _N, _N, _N, _N, _N]  # noqa: E731
    device = [
        [0, 0, 0, _N, 200, -200, 200, -200, _N, _N, _N, _N, _N, _N, _N],  # slack
        load(1, 3, -10),  # residential load
        [2, 3, 2, _N, 30, 0, 30, -30, 20, _N, 15, -15, _N, _N, _N],  # PV
        load(3, 4, -30),  # industrial load
        [4, 4, 2, _N, 50, 0, 50, -50, 35, _N, 20, -20, _N, _N, _N],  # wind
        load(5, 5, -30),  # EV charging
        [6, 5, 3, _N, 50, -50, 50, -50, 30, -30, 25, -25, 100, 0, 0.9],  # storage
    ]
    branch = [
        [0, 1, 0.0036, 0.1834, 0.0, 32, 1, 0],
        [1, 2, 0.03, 0.022, 0.0, 25, 1, 0],
        [1, 3, 0.0307, 0.0621, 0.0, 18, 1, 0],
        [2, 4, 0.0303, 0.0611, 0.0, 18, 1, 0],
        [2, 5, 0.0159, 0.0502, 0.0, 18, 1, 0],
    ]
    return {
        "baseMVA": 100.0,
        "bus": np.array(bus, dtype=float),
        "device": np.array(device, dtype=object),
        "branch": np.array(branch, dtype=float),
    }


def two_bus_network(base_mva=1) -> dict:
    """Slack + one load over a single line (topology of ``test_simulator_transitions.py:18-32``)."""
    return {
        "baseMVA": base_mva,
        "bus": np.array([[0, 0, 50, 1.0, 1.0], [1, 1, 50, 1.1, 0.9]]),
        "branch": np.array([[0, 1, 0.01, 0.1, 0.0, 32, 1, 0]]),
        "device": np.array(
            [
                [0, 0, 0, _N, 200, -200, 200, -200, _N, _N, _N, _N, _N, _N, _N],
                [1, 1, -1, 0.2, 0, -10, _N, _N, _N, _N, _N, _N, _N, _N, _N],
            ],
            dtype=object,
        ),
    }


def three_bus_loop_network(tap=1.0, shift=0.0, base_mva=1, gen_max=200.0) -> dict:
    """3-bus meshed network with line charging and an optional off-nominal transformer on
    branch (1,2) (topology of ``test_simulator_transitions.py:48-69`` / ``:93-121``)."""
    g = gen_max
    return {
        "baseMVA": base_mva,
        "bus": np.array([[0, 0, 50, 1.0, 1.0], [1, 1, 50, 1.1, 0.9], [2, 1, 50, 1.1, 0.9]]),
        "branch": np.array(
            [
                [0, 1, 0.01, 0.1, 0.0, 30, 1, 0],
                [1, 2, 0.02, 0.3, 0.2, 30, tap, shift],
                [2, 0, 0.05, 0.2, 0.1, 30, 1, 0],
            ]
        ),
        "device": np.array(
            [
                [0, 0, 0, _N, 200, -200, 200, -200, _N, _N, _N, _N, _N, _N, _N],
                [1, 1, -1, 0.2, 0, -10, _N, _N, _N, _N, _N, _N, _N, _N, _N],
                [2, 1, 1, _N, g, 0, g, -g, _N, _N, _N, _N, _N, _N, _N],
                [3, 2, 2, _N, g, 0, g, -g, _N, _N, _N, _N, _N, _N, _N],
                [4, 2, 3, _N, g, -g, g, -g, _N, _N, _N, _N, 100, 0, 0.9],
            ],
            dtype=object,
        ),
    }


def synthetic_radial_network(n_bus: int = 30, seed: int = 0) -> dict:
    """Seeded random radial (tree) distribution feeder (BASELINE.json config 4).

    Bus 0 is the 132 kV slack; bus 1 hangs off it through a transformer-like reactance; every
    further bus picks a random parent among the earlier non-slack buses.  Each bus >= 2 carries
    at most one device: load (55 %), renewable generator (20 %), storage (10 %) or nothing (15 %).
    """
    rng = np.random.default_rng(seed)
    bus = [[0, 0, 132, 1.0, 1.0]] + [[i, 1, 33, 1.1, 0.9] for i in range(1, n_bus)]
    branch = [[0, 1, float(rng.uniform(0.003, 0.03)), 0.18, 0.0, 400.0, 1, 0]]
    for i in range(2, n_bus):
        parent = int(rng.integers(1, i))
        branch.append(
            [
                parent,
                i,
                float(rng.uniform(0.003, 0.03)),
                float(rng.uniform(0.02, 0.06)),
                0.0,
                float(np.round(rng.uniform(15, 35), 1)),
                1,
                0,
            ]
        )
    device = [[0, 0, 0, _N, 1000, -1000, 1000, -1000, _N, _N, _N, _N, _N, _N, _N]]
    dev_id = 1
    for i in range(2, n_bus):
        u = rng.uniform()
        pmax = float(np.round(rng.uniform(2, 20), 1))
        pplus = float(np.round(pmax * rng.uniform(0.6, 0.7), 2))
        if u < 0.55:
            device.append([dev_id, i, -1, 0.2, 0, -pmax, _N, _N, _N, _N, _N, _N, _N, _N, _N])
        elif u < 0.75:
            device.append([dev_id, i, 2, _N, pmax, 0, pmax, -pmax, pplus, _N, 0.5 * pmax, -0.5 * pmax, _N, _N, _N])
        elif u < 0.85:
            device.append(
                [dev_id, i, 3, _N, pmax, -pmax, pmax, -pmax, pplus, -pplus, 0.5 * pmax, -0.5 * pmax, 50, 0, 0.9]
            )
        else:
            continue
        dev_id += 1
    return {
        "baseMVA": 100.0,
        "bus": np.array(bus, dtype=float),
        "device": np.array(device, dtype=object),
        "branch": np.array(branch, dtype=float),
    }


def perturbed_network(net: dict, seed: int, rel=0.2) -> dict:
    """A variant of ``net`` with the same topology and other numbers: branch r, x, rating, tap and (small)
    line charging, the voltage limits of the PQ buses and every finite device limit are scaled by factors
    drawn uniformly in [1 - rel, 1 + rel] (limits keep their sign and their ordering).  Used for
    per-environment heterogeneous networks (``BatchedSimulator(variants=...)``)."""
    rng = np.random.default_rng(seed)
    f = lambda n=None: rng.uniform(1 - rel, 1 + rel, size=n)  # noqa: E731
    out = {"baseMVA": net["baseMVA"], "bus": np.array(net["bus"], dtype=float).copy(),
           "device": np.array(net["device"], dtype=object).copy(), "branch": np.array(net["branch"], dtype=float).copy()}
    br = out["branch"]
    br[:, 2] *= f(len(br))
    br[:, 3] *= f(len(br))
    br[:, 4] = rng.uniform(0, 0.01, len(br))
    br[:, 5] *= f(len(br))
    br[:, 6] = np.where(rng.uniform(size=len(br)) < 0.3, rng.uniform(0.97, 1.03, len(br)), br[:, 6])
    bus = out["bus"]
    pq = bus[:, 1] != 0
    bus[pq, 3] = 1.0 + 0.1 * f(int(pq.sum()))
    bus[pq, 4] = 1.0 - 0.1 * f(int(pq.sum()))
    for row in out["device"]:
        s = f()
        for col in (4, 5, 6, 7, 8, 9, 10, 11, 12):  # PMAX PMIN QMAX QMIN P+ P- Q+ Q- SOC_MAX: one factor per device
            if row[col] is not None:
                row[col] = float(row[col]) * s
        if row[14] is not None:
            row[14] = float(min(1.0, float(row[14]) * rng.uniform(0.9, 1.05)))
    return out


def synthetic_meshed_network(n_bus: int, seed: int, n_chords: int) -> dict:
    """The random feeder of :func:`synthetic_radial_network` plus ``n_chords`` extra branches that close
    loops, with line charging; the first chord is an off-nominal transformer with a phase shift.  Exercises
    fill-in of the block LU, asymmetric admittance entries and both ends of every branch."""
    net = synthetic_radial_network(n_bus, seed)
    rng = np.random.default_rng(1000 + seed)
    have = {(int(min(f, t)), int(max(f, t))) for f, t in net["branch"][:, :2]}
    extra = []
    while len(extra) < n_chords:
        f, t = sorted(int(x) for x in rng.choice(np.arange(1, n_bus), size=2, replace=False))
        if (f, t) in have:
            continue
        have.add((f, t))
        tap, shift = (1.0, 0.0) if extra else (0.97, 3.0)
        extra.append([f, t, float(rng.uniform(0.005, 0.03)), float(rng.uniform(0.03, 0.08)), float(rng.uniform(0, 0.02)),
                      30.0, tap, shift])  # fmt: skip
    net["branch"] = np.vstack([net["branch"], np.array(extra)])
    return net
