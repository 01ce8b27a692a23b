"""CPU tier: the counter-based sampler is Philox4x32-10 (Random123's known answers), and the stand-alone sampler entry point
(anm_sample_init_state_f64), through the host test double of the kernel templates, draws exactly what gym_anm_amd/rng.py
specifies.  The GPU tier (tests/test_gpu_sampler.py) holds the kernels to the same specification and to the distribution of
the reference's own ``ANM6Easy.init_state`` (gym_anm/envs/anm6_env/anm6_easy.py:25-52)."""
import numpy as np
import numpy.testing as npt
import pytest

from gym_anm_amd import networks, rng
from gym_anm_amd.model import NetworkModel

# Random123 (D. E. Shaw Research) kat_vectors, philox4x32 with 10 rounds: counter[4], key[2] -> output[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


@pytest.mark.parametrize("ctr,key,out", KAT)
def test_philox4x32_10_known_answers(ctr, key, out):
    # the kernel's key layout: counter = (env lo, env hi, epoch, draw), key = (seed lo, seed hi)
    got = rng.philox4x32(key[0] | (key[1] << 32), ctr[0] | (ctr[1] << 32), ctr[2], ctr[3])
    assert tuple(got) == out


def test_u01_is_the_53_bit_uniform():
    assert rng.u01(0, 0) == 0.0
    assert rng.u01(0xFFFFFFFF, 0xFFFFFFFF) == 1.0 - 2.0 ** -53
    assert rng.u01(0x80000000, 0) == 0.5
    assert rng.u01(0, 0x7FF) == 0.0          # the low 11 bits are dropped


def test_host_double_of_the_sampler_entry_point_draws_the_specification():
    """anm_sample_init_state_f64 on the host-compiled kernel templates (the thread family's sample_series_init_state):
    rows and raw Philox words against rng.py for 2 000 (seed, environment, epoch) keys"""
    import ctypes as C

    from hostsim_backend import hostsim_backend

    from gym_anm_amd import _lib
    from gym_anm_amd.envs import ANM6EasyVec

    net = networks.anm6_network()
    model = NetworkModel(net, 0.25, 100)
    E_, SEED, OFF = 2000, 0x1234567890ABCDEF, (1 << 33) + 5
    env = ANM6EasyVec(num_envs=E_, device="cpu", seed=SEED & (2**62 - 1), env_offset=OFF, _backend=hostsim_backend(model.topology()))
    epochs = np.random.default_rng(1).integers(0, 2**31 - 1, E_).astype(np.int32)
    env._reset_count.copy_(__import__("torch").as_tensor(epochs))
    rows, raw = env.sample_init_state(raw=True)
    rows, raw = rows.numpy(), raw.numpy()
    seed = env.rng_seed
    for e in range(E_):
        for b in range(raw.shape[1]):
            assert tuple(raw[e, b]) == tuple(rng.philox4x32(seed, OFF + e, int(epochs[e]), b)), (e, b)
        want = rng.series_init_state(model, env._series, seed, OFF + e, int(epochs[e]))
        npt.assert_array_equal(rows[e], want)      # (g++ -ffp-contract=off: the same unfused arithmetic as Python)
    # the time index really is floor(word * period / 2^32)
    npt.assert_array_equal(rows[:, -1], (raw[:, 0, 0] * 96) >> 32)
