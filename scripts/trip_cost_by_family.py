"""Cost of one Newton trip of the slowest wave, both kernel families, ANM6Easy at 65 536 environments:
(step time with cap 100 - step time with cap 20) / 80."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd.envs import ANM6EasyVec
DEV="cuda:0"; E=65536
for impl in ("thread","radial"):
    res=[]
    for cap in (20,100):
        env = ANM6EasyVec(num_envs=E, device=DEV, seed=1, autoreset=True, tol=1e-6, max_iter=cap, impl=impl, straggler_after=None)
        env.check_actions=False; env.reset(seed=1)
        g=torch.Generator(device=DEV).manual_seed(0)
        lo=torch.as_tensor(env.action_space.low,device=DEV); hi=torch.as_tensor(env.action_space.high,device=DEV)
        pool=[lo+(hi-lo)*torch.rand((E,6),generator=g,dtype=torch.float64,device=DEV) for _ in range(8)]
        for i in range(10): env.step(pool[i%8])
        torch.cuda.synchronize(); t=time.perf_counter(); n=100
        for i in range(n): env.step(pool[i%8])
        torch.cuda.synchronize(); res.append((time.perf_counter()-t)/n*1e6)
    print(impl, "cap20 %.1f us cap100 %.1f us -> %.2f us per trip of the slowest wave" % (res[0],res[1],(res[1]-res[0])/80))
