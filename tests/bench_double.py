"""Entry script of tests/test_bench_launcher.py -- TEST INFRASTRUCTURE, not a benchmark.

``bench.main`` with the host test double (tests/hostsim: the kernel templates compiled with g++), the ``gloo`` backend
and CPU tensors, so that the launcher and the rank bookkeeping of ``bench.py`` -- self-launch under
``torch.distributed.run``, contiguous sharding, ``env_offset``, the rank-count assertion, MAX / SUM reductions, the one
JSON line from rank 0 -- run in the CPU tier with two ranks.  ``bench.py`` itself has no switch that selects this."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402


def make_env(num_envs, device, args, env_offset, seed):
    from hostsim_backend import hostsim_backend

    from gym_anm_amd import networks
    from gym_anm_amd.envs import ANM6EasyVec
    from gym_anm_amd.model import NetworkModel

    be = hostsim_backend(NetworkModel(networks.anm6_network(), 0.25, 100).topology())
    return ANM6EasyVec(num_envs=num_envs, device="cpu", seed=seed, tol=args.tol, max_iter=args.max_iter, autoreset=True,
                       env_offset=env_offset, _backend=be)


if __name__ == "__main__":
    bench.main(make_env=make_env, backend="gloo", device_type="cpu", script=os.path.abspath(__file__))
