#!/usr/bin/env python
"""CPU baseline leg of bench.py (TEST/MEASUREMENT INFRASTRUCTURE, never the product path).

Times the oracle (oracle/anm_oracle.py: the NumPy/SciPy restatement of the reference algorithm,
literal scipy.sparse Jacobian + spsolve path of gym_anm/simulator/solve_load_flow.py:176-226) on
the headline workload: ANM6Easy, uniform random actions in the action Box, reset on collapse
(anm6_easy.py:25-52 sampler), one environment stepped sequentially per process.

    python oracle/cpu_bench.py --seconds 8 --seed 3      -> one JSON line {"steps": n, "seconds": t}

bench.py runs one in-process sample for the 1-core figure and one subprocess per host core for
the all-cores figure (independent environments: the CPU analogue of the sharded batch).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def run(budget_s, seed=0):
    import anm_oracle as O
    from gym_anm_amd import networks

    env = O.OracleEnv(networks.anm6_network(), sparse=True, tol=1e-5)
    tables = O.anm6easy_tables()
    rng = np.random.default_rng(seed)
    lo = np.array([0, 0, -30, -50, -50, -50.0])
    hi = np.array([30, 50, 30, 50, 50, 50.0])

    def reset():
        t0 = int(rng.integers(0, 96))
        s0 = np.zeros(18)
        s0[[1, 3, 5]] = tables[:3, t0]
        s0[[2, 4]] = tables[3:, t0]
        s0[[15, 16]] = tables[3:, t0]
        s0[9], s0[11] = rng.uniform(-0.3, 0.3), rng.uniform(-0.5, 0.5)
        s0[14] = rng.uniform(0, 1)
        s0[17] = t0
        env.reset_to(s0)

    reset()
    n, t0, c0 = 0, time.perf_counter(), time.process_time()
    while time.perf_counter() - t0 < budget_s:
        _, _, term = env.step(rng.uniform(lo, hi))
        n += 1
        if term:
            reset()
    return n, time.perf_counter() - t0, time.process_time() - c0


def run_reference(budget_s, seed=0):
    """The UNMODIFIED reference (gym_anm.envs.ANM6Easy from /root/reference, imported through oracle/ref_harness.py: stubs
    for gymnasium / websocket, the exact-projection stand-in for the one cvxpy QP) on the same workload: uniform random
    actions in the action Box, reset on collapse.  Dev container only (the reference does not travel to the GPU box)."""
    import ref_harness

    if not ref_harness.reference_available():
        raise SystemExit("the reference is not available here (/root/reference)")
    ref_harness.load_reference()
    from gym_anm.envs import ANM6Easy

    env = ANM6Easy()
    env.reset(seed=seed)
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(env.action_space.low, float), np.asarray(env.action_space.high, float)
    n, t0, c0 = 0, time.perf_counter(), time.process_time()
    while time.perf_counter() - t0 < budget_s:
        _, _, term, _, _ = env.step(rng.uniform(lo, hi))
        n += 1
        if term:
            env.reset()
    return n, time.perf_counter() - t0, time.process_time() - c0


def run_cpp(budget_s, seed=0, num_envs=2048):
    """Second CPU figure: the kernel templates themselves compiled for the host with g++ -O2
    (tests/hostsim, the test double of the CPU tier), single-threaded loop over `num_envs` environments
    per step, random agent with in-"kernel" autoreset -- an upper bound for a compiled CPU port."""
    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from hostsim_backend import hostsim_backend
    from gym_anm_amd import networks
    from gym_anm_amd.envs import ANM6EasyVec
    from gym_anm_amd.model import NetworkModel

    torch.set_num_threads(1)
    be = hostsim_backend(NetworkModel(networks.anm6_network(), 0.25, 100).topology())
    env = ANM6EasyVec(num_envs=num_envs, device="cpu", seed=seed, autoreset=True, tol=1e-5, _backend=be)
    env.check_actions = False
    env.reset(seed=seed)
    g = torch.Generator().manual_seed(seed)
    lo, hi = torch.as_tensor(env.action_space.low), torch.as_tensor(env.action_space.high)
    pool = [lo + (hi - lo) * torch.rand((num_envs, 6), generator=g, dtype=torch.float64) for _ in range(4)]
    env.step(pool[0])
    n, t0, c0 = 0, time.perf_counter(), time.process_time()
    while time.perf_counter() - t0 < budget_s:
        env.step(pool[n % 4])
        n += 1
    return n * num_envs, time.perf_counter() - t0, time.process_time() - c0


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota if any."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / period + 0.5)))
            break
        except Exception:
            continue
    return n


def run_all_cores(budget_s, n_proc=None, impl="numpy"):
    """One independent process per usable host core, each stepping its own environment(s) for `budget_s`."""
    import subprocess

    n_proc = n_proc or usable_cores()
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--seconds", str(budget_s), "--impl", impl]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd + ["--seed", str(100 + i)], stdout=subprocess.PIPE, env=env) for i in range(n_proc)]
    total, slowest, cpu = 0, 0.0, 0.0
    for p in procs:
        out, _ = p.communicate(timeout=budget_s * 10 + 120)
        r = json.loads(out.decode().strip().splitlines()[-1])
        total += r["steps"]
        slowest = max(slowest, r["seconds"])
        cpu += r["cpu_seconds"]
    # cpu / slowest = cores' worth of CPU time the processes really got (a cgroup quota shows up here)
    return total, slowest, n_proc, time.perf_counter() - t0, cpu / slowest


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--impl", choices=["numpy", "cpp", "reference"], default="numpy")
    ap.add_argument("--reference", action="store_true",
                    help="dev container: the unmodified reference and the oracle on the same workload, one after the other")
    a = ap.parse_args()
    if a.reference:
        nr, tr, _ = run_reference(a.seconds, a.seed)
        no, to, _ = run(a.seconds, a.seed)
        print(json.dumps({"reference_env_steps_per_s": nr / tr, "oracle_env_steps_per_s": no / to, "ratio_oracle_over_reference": (no / to) / (nr / tr),
                          "seconds_each": a.seconds, "cores": 1, "host": os.uname().nodename,
                          "workload": "ANM6Easy, uniform random actions in the action Box, reset on collapse, tol 1e-5, one environment"}))
    else:
        n, dt, cpu = {"numpy": run, "cpp": run_cpp, "reference": run_reference}[a.impl](a.seconds, a.seed)
        print(json.dumps({"steps": n, "seconds": dt, "cpu_seconds": cpu}))
