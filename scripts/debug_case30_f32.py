import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_anm_amd import networks
from gym_anm_amd.simulator import BatchedSimulator
from gym_anm_amd import _lib
import ctypes
if len(sys.argv) > 1:
    path = sys.argv[1]
    _lib.load_for_topology = lambda topo: _lib.Backend(ctypes.CDLL(path), "cuda", path)
    print("using", path)
g = np.load('tests/golden/transition_case30.npz')
net = networks.synthetic_radial_network(30, 0)
M = len(g['n_iter'])
out = {}
for prec in ('f64', 'f32'):
    for mi in (1, 100):
        sim = BatchedSimulator(net, 0.25, 100, num_envs=M, device='cuda:0', precision=prec, max_iter=mi)
        sim.soc.copy_(torch.as_tensor(g['soc0']))
        st, r, e, p, conv = sim.transition(g['P_load'], g['P_pot'], g['P_set'], g['Q_set'])
        torch.cuda.synchronize()
        it = sim.nr_iters.cpu().numpy()
        print(prec, mi, 'conv', int(conv.sum()), 'iters', np.bincount(it)[:8], it.max())
        out['%s_%d_full' % (prec, mi)] = sim.full.cpu().numpy()
        out['%s_%d_it' % (prec, mi)] = it

