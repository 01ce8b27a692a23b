import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from gym_anm_amd.envs import ANM6EasyVec
DEV="cuda:0"; E=8192
envs=[ANM6EasyVec(num_envs=E, device=DEV, seed=5, autoreset=True, impl="radial") for _ in range(2)]
for e in envs: e.check_actions=False; e.reset(seed=5)
gen=torch.Generator(device=DEV).manual_seed(4)
lo=torch.as_tensor(envs[0].action_space.low,device=DEV); hi=torch.as_tensor(envs[0].action_space.high,device=DEV)
for t in range(12):
    a=lo+(hi-lo)*torch.rand((E,6),generator=gen,dtype=torch.float64,device=DEV)
    was=envs[0].terminated.clone()
    os.environ["ANM_RADIAL_GENERIC"]="1"; o0,r0,t0,_,_=envs[0].step(a); torch.cuda.synchronize()
    del os.environ["ANM_RADIAL_GENERIC"]; o1,r1,t1,_,_=envs[1].step(a); torch.cuda.synchronize()
    bad=((o0-o1).abs()>1e-9).any(dim=1) | (t0!=t1)
    idx=torch.nonzero(bad)[:,0]
    print("step",t,"mismatching envs",len(idx),"resetting",int(was.sum()),"term",int(t0.sum()))
    for i in idx[:6].tolist():
        print("  env",i,"lane-in-wave",i%8,"wave",i//8,"was_term",bool(was[i]),"term",bool(t0[i]),bool(t1[i]),"iters",int(envs[0].simulator.nr_iters[i]),int(envs[1].simulator.nr_iters[i]))
        print("   cols", torch.nonzero((o0[i]-o1[i]).abs()>1e-9)[:,0].tolist(), (o0[i]-o1[i])[[0,2,4,6,14]].tolist())
        nb=[j for j in range((i//8)*8,(i//8)*8+8)]
        print("   wave neighbours was_term", was[nb].int().tolist(), "term", t0[nb].int().tolist(), "iters", envs[1].simulator.nr_iters[nb].tolist())
    if len(idx): 
        envs[1].state.copy_(envs[0].state); envs[1]._state_obs.copy_(envs[0]._state_obs); envs[1].simulator.soc.copy_(envs[0].simulator.soc)
        envs[1]._term_u8.copy_(envs[0]._term_u8); envs[1]._reset_count.copy_(envs[0]._reset_count); envs[1].timestep.copy_(envs[0].timestep)
