"""N > 1 path on CPU: world_size-2 gloo.  Each rank owns a contiguous shard of the environment
batch (the way bench.py shards it over GPUs: same seed, env_offset = rank * num_envs) and runs it
on the host test double; the gathered shards must equal the unsharded batch bit for bit --
including the environments re-initialised in-kernel by the counter-based RNG -- and the only
collectives are the ones bench.py uses (all_reduce MAX of the time, SUM of the step count)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

E_TOTAL, T, SEED = 512, 40, 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_env(num_envs, offset):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from hostsim_backend import hostsim_backend
    from gym_anm_amd import networks
    from gym_anm_amd.envs import ANM6EasyVec
    from gym_anm_amd.model import NetworkModel

    be = hostsim_backend(NetworkModel(networks.anm6_network(), 0.25, 100).topology())
    env = ANM6EasyVec(num_envs=num_envs, device="cpu", seed=SEED, autoreset=True, env_offset=offset, _backend=be)
    env.check_actions = False
    return env


def _inputs():
    rng = np.random.default_rng(SEED)
    full = _make_env(E_TOTAL, 0)
    full.reset(seed=SEED)
    s0 = full.state.numpy().copy()  # common initial states
    lo, hi = full.action_space.low, full.action_space.high
    acts = rng.uniform(lo, hi, size=(T, E_TOTAL, 6))
    acts[:, :, 2:] *= 1.0
    return s0, acts


def _run(env, s0, acts):
    env.reset(options={"init_state": torch.as_tensor(s0)})
    obs_l, r_l, t_l = [], [], []
    for t in range(acts.shape[0]):
        o, r, term, _, _ = env.step(torch.as_tensor(acts[t]))
        obs_l.append(o.clone())
        r_l.append(r.clone())
        t_l.append(term.clone())
    return torch.stack(obs_l), torch.stack(r_l), torch.stack(t_l)


def _worker(rank, world, port, s0, acts, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = E_TOTAL // world
    sl = slice(rank * n, (rank + 1) * n)
    env = _make_env(n, rank * n)
    obs, r, term = _run(env, s0[sl], acts[:, sl])
    # data path: no collective.  Reporting path (bench.py): MAX of elapsed, SUM of steps.
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    steps = torch.tensor([float(n * T)])
    dist.all_reduce(steps, op=dist.ReduceOp.SUM)
    g_obs = [torch.zeros_like(obs) for _ in range(world)]
    g_r = [torch.zeros_like(r) for _ in range(world)]
    g_t = [torch.zeros_like(term) for _ in range(world)]
    dist.all_gather(g_obs, obs)
    dist.all_gather(g_r, r)
    dist.all_gather(g_t, term)
    if rank == 0:
        out["obs"] = torch.cat(g_obs, dim=1).numpy()
        out["r"] = torch.cat(g_r, dim=1).numpy()
        out["term"] = torch.cat(g_t, dim=1).numpy()
        out["t_max"] = float(t)
        out["steps"] = float(steps)
        out["resets"] = int(env._reset_count.sum())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_equals_unsharded_gloo():
    s0, acts = _inputs()
    ref_env = _make_env(E_TOTAL, 0)
    obs, r, term = _run(ref_env, s0, acts)
    assert int(ref_env._reset_count.sum()) > 0, "workload must exercise the in-kernel autoreset"
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), s0, acts, out), nprocs=2, join=True)
    np.testing.assert_array_equal(out["obs"], obs.numpy())
    np.testing.assert_array_equal(out["r"], r.numpy())
    np.testing.assert_array_equal(out["term"], term.numpy())
    assert out["t_max"] == 2.0 and out["steps"] == float(E_TOTAL * T)
    assert out["resets"] > 0
