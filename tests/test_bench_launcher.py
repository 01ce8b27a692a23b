"""bench.py's N > 1 machinery in the CPU tier: `python bench.py --gpus N` typed without a launcher starts its own ranks
(torch.distributed.run, 127.0.0.1), the batch is sharded contiguously with `env_offset`, the communicator's rank count is
asserted, the time is the slowest rank's and the step count the sum, per-rank side figures (BASELINE config 4) are
reduced as sum of rates / slowest launch, rank 0 prints ONE JSON line.  Ranks run on the host test double over gloo
(tests/bench_double.py); the reference loop being sharded is gym_anm/envs/anm_env.py:333-453."""
import json
import os
import socket
import subprocess
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _args(**kw):
    a = types.SimpleNamespace(num_envs=65536, global_envs=0)
    a.__dict__.update(kw)
    return a


def test_shard_weak_and_strong():
    assert bench.shard(_args(), 8, 3) == (65536, 3 * 65536, "weak")
    assert bench.shard(_args(global_envs=524288), 8, 3) == (65536, 3 * 65536, "strong")
    assert bench.shard(_args(global_envs=524288), 1, 0) == (524288, 0, "strong")
    with pytest.raises(SystemExit):
        bench.shard(_args(global_envs=1001), 8, 0)


def test_rank_info_and_launch_command():
    assert bench.rank_info({}) == (0, 0, 1, False)
    assert bench.rank_info({"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8"}) == (3, 3, 8, True)
    cmd = bench.self_launch_command(4, "/x/bench.py", ["--gpus", "4", "--steps", "7"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "7"]


def test_single_rank_comm_is_the_identity():
    c = bench.Comm("gloo", 0, 1, torch.device("cpu"), launched=False)
    assert c.ranks_seen is None and c.reduce([1.5, 2.5], "max") == [1.5, 2.5]
    fig = {"case30_radial_16384_cap100": {"env_steps_per_s": 9.0e7, "us_per_launch": 180.0, "roofline": {"note": "x"}}}
    out = bench.reduce_side_figure(c, fig)
    assert out["case30_radial_16384_cap100"]["env_steps_per_s"] == 9.0e7 and out["case30_radial_16384_cap100"]["ranks"] == 1
    c.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reduce_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    c = bench.Comm("gloo", rank, world, torch.device("cpu"), launched=True)
    fig = {"case30_radial_16384_cap100": {"env_steps_per_s": 1.0e8 + rank, "us_per_launch": 160.0 + 10 * rank, "impl": "radial",
                                          "roofline": {"note": "n"}},
           "case30_radial_16384_cap20": {"env_steps_per_s": 2.0e8, "us_per_launch": 70.0 - rank}}
    red = bench.reduce_side_figure(c, fig)
    t = c.reduce([1.0 + rank], "max")[0]
    n = c.reduce([100.0], "sum")[0]
    if rank == 0:
        out["red"], out["t"], out["n"], out["seen"] = red, t, n, c.ranks_seen
    c.close()


@pytest.mark.timeout(300)
def test_side_figures_reduce_over_two_gloo_ranks():
    out = mp.Manager().dict()
    mp.spawn(_reduce_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = out["red"]
    assert out["seen"] == 2 and out["t"] == 2.0 and out["n"] == 200.0
    assert r["case30_radial_16384_cap100"]["env_steps_per_s"] == 2.0e8 + 1 and r["case30_radial_16384_cap100"]["us_per_launch"] == 170.0
    assert r["case30_radial_16384_cap20"]["env_steps_per_s"] == 4.0e8 and r["case30_radial_16384_cap20"]["us_per_launch"] == 70.0
    assert r["case30_radial_16384_cap100"]["ranks"] == 2 and r["case30_radial_16384_cap100"]["impl"] == "radial"
    assert "summed over the ranks" in r["case30_radial_16384_cap100"]["roofline"]["note"]


@pytest.mark.timeout(600)
def test_bench_typed_with_gpus_2_starts_its_own_two_ranks():
    """the command as a user types it -- no torchrun in front -- on the host test double"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(HERE, "bench_double.py"), "--gpus", "2", "--num-envs", "256", "--steps", "6",
                          "--warmup", "2", "--headline-only", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=540)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints, once: %r" % res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and out["scaling"] == "weak"
    assert out["config"]["num_envs_per_gpu"] == 256 and out["config"]["global_num_envs"] == 512
    assert out["steps"] == 6 and out["warmup"] == 2
    # value = the units ALL ranks processed / the slowest rank's time
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 512.0) < 1e-6 * 512
    # strong scaling: the same total split over the ranks
    res = subprocess.run([sys.executable, os.path.join(HERE, "bench_double.py"), "--gpus", "2", "--global-envs", "512", "--steps", "4",
                          "--warmup", "1", "--headline-only", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=540)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["num_envs_per_gpu"] == 256 and out["config"]["global_num_envs"] == 512


def test_more_gpus_asked_than_visible_exits_with_one_line_before_spawning():
    """`bench.py --gpus N` with fewer than N visible devices must say so and start nothing"""
    assert bench.preflight_devices(1, 1) is None and bench.preflight_devices(8, 8) is None
    assert "--gpus 2 but only 1 GPU is visible" in bench.preflight_devices(2, 1)
    assert "--gpus 8 but only 0 GPUs are visible" in bench.preflight_devices(8, 0)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        pytest.skip("a box with 64 GPUs")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=280)
    assert res.returncode != 0 and res.stdout.strip() == ""
    err = [ln for ln in res.stderr.splitlines() if ln.strip()]
    assert len(err) == 1 and "--gpus 64 but only" in err[0] and "nothing started" in err[0], res.stderr[-2000:]


@pytest.mark.timeout(300)
def test_a_missing_peer_does_not_hang_the_rendezvous():
    """one rank of two whose peer never arrives (it could not set its device): init_process_group gives up after
    ANM_BENCH_RDZV_TIMEOUT and the rank leaves with a message, it does not wait for ever"""
    import time

    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               ANM_BENCH_RDZV_TIMEOUT="4")
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(HERE, "bench_double.py"), "--gpus", "2", "--num-envs", "64", "--steps", "1",
                          "--warmup", "0", "--headline-only", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=240)
    assert res.returncode != 0
    assert "did not complete within 4 s" in res.stderr and "a peer rank is missing" in res.stderr, res.stderr[-2000:]
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 200
