"""The headline step kernel launched ALONE (synchronize before and after every launch) and back to back (each launch queued
behind the previous one on the stream), timed by HIP events inside the library (anm_time_step_launches).
usage: python scripts/isolated_vs_queued.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec

dev = torch.device("cuda", 0)
E = 65536
env = ANM6EasyVec(num_envs=E, device=dev, seed=1, tol=1e-6, max_iter=100, autoreset=True)
env.check_actions = False
env.reset(seed=1)
g = torch.Generator(device=dev).manual_seed(0)
lo = torch.as_tensor(env.action_space.low, device=dev); hi = torch.as_tensor(env.action_space.high, device=dev)
pool = [lo + (hi - lo) * torch.rand((E, 6), generator=g, dtype=torch.float64, device=dev) for _ in range(8)]
for i in range(30): env.step(pool[i % 8])
sim = env.simulator

def timed(n_launch):
    ms = C.c_float(0.0)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = sim.backend.lib.anm_time_step_launches(
            sim._handle, E, pool[0].data_ptr(), sim.soc.data_ptr(), env._state_buf.data_ptr(), env._term_u8.data_ptr(),
            env.timestep.data_ptr(), env._state_obs.data_ptr(), env.reward.data_ptr(), env.e_loss.data_ptr(),
            env.penalty.data_ptr(), 1, env.rng_seed, env.env_offset, env._reset_count.data_ptr(), env._aux_index_ptr,
            env._ws_ref, C.byref(sim.opts), stream, n_launch, C.byref(ms))
    sim.backend.check(rc, "anm_time_step_launches")
    return ms.value * 1e3   # us per launch

for rnd in range(3):
    torch.cuda.synchronize()
    iso = []
    for _ in range(50):
        torch.cuda.synchronize()
        iso.append(timed(1))
    q = timed(200)
    print("isolated launches: mean %.1f us (min %.1f, max %.1f);  200 launches back to back: %.1f us per launch" % (sum(iso) / len(iso), min(iso), max(iso), q), flush=True)
