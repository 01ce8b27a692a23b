"""Where the time of a Simulator.transition call goes beyond its kernel: events around (a) the bare C launch, (b) the bare launch
followed by the converged-flag conversion, (c) the Python method.  usage: python scripts/transition_host_path.py [cap]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_anm_amd import networks
from gym_anm_amd.simulator import BatchedSimulator, _stream_ptr

dev = torch.device("cuda", 0)
E, cap = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 100
sim = BatchedSimulator(networks.anm6_network(), 0.25, 100, num_envs=E, device=dev, tol=1e-6, max_iter=cap)
m, b = sim.model, sim.model.baseMVA
g = torch.Generator(device=dev).manual_seed(0)
U = lambda lo, hi: (torch.as_tensor(lo, device=dev) + (torch.as_tensor(hi, device=dev) - torch.as_tensor(lo, device=dev))
                    * torch.rand((E, len(lo)), generator=g, dtype=torch.float64, device=dev))
pl = U(m.dev_p_min[m.load_idx] * b, 0 * m.dev_p_min[m.load_idx]); pp = U(0 * m.dev_p_max[m.gen_idx], m.dev_p_max[m.gen_idx] * b)
ps = U(m.dev_p_min[m.setp_idx] * b, m.dev_p_max[m.setp_idx] * b); qs = U(m.dev_q_min[m.setp_idx] * b, m.dev_q_max[m.setp_idx] * b)
soc = U(m.dev_soc_min[m.des_idx], m.dev_soc_max[m.des_idx])

def raw():
    with sim._device_ctx():
        return sim.backend.lib.anm_transition_f64(sim._handle, E, pl.data_ptr(), pp.data_ptr(), ps.data_ptr(), qs.data_ptr(), sim.soc.data_ptr(),
                                                  sim.full.data_ptr(), sim.reward.data_ptr(), sim.e_loss.data_ptr(), sim.penalty.data_ptr(),
                                                  sim._conv_u8.data_ptr(), sim.nr_iters.data_ptr(), C.byref(sim.opts), _stream_ptr(sim.device))
def raw_bool():
    raw(); return sim._conv_u8.bool()
def method():
    return sim.transition(pl, pp, ps, qs)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rnd in range(2):
    for name, fn in (("bare launch", raw), ("bare launch + .bool()", raw_bool), ("Simulator.transition", method)):
        t = 0.0
        for _ in range(20):
            sim.soc.copy_(soc); torch.cuda.synchronize()
            ev0.record(); fn(); ev1.record(); torch.cuda.synchronize()
            t += ev0.elapsed_time(ev1)
        print("cap %d  %-24s %.1f us (iterations %d..%d)" % (cap, name, t / 20 * 1e3, int(sim.nr_iters.min()), int(sim.nr_iters.max())), flush=True)

# back to back, no synchronize in between: (a) the bare launch alone, (b) followed by .bool(), (c) followed by an in-place torch op on
# another tensor, (d) the bare launch WITHOUT the dump followed by .bool()
import time
def raw_nodump():
    with sim._device_ctx():
        return sim.backend.lib.anm_transition_f64(sim._handle, E, pl.data_ptr(), pp.data_ptr(), ps.data_ptr(), qs.data_ptr(), sim.soc.data_ptr(),
                                                  None, sim.reward.data_ptr(), sim.e_loss.data_ptr(), sim.penalty.data_ptr(),
                                                  sim._conv_u8.data_ptr(), sim.nr_iters.data_ptr(), C.byref(sim.opts), _stream_ptr(sim.device))
scratch = torch.zeros(16, device=dev)
for name, fn in (("bare x50", lambda: raw()), ("bare + .bool() x50", lambda: raw_bool()), ("bare + tiny add_ x50", lambda: (raw(), scratch.add_(1.0))),
                 ("bare(no dump) x50", lambda: raw_nodump()), ("bare(no dump) + .bool() x50", lambda: (raw_nodump(), sim._conv_u8.bool()))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(50):
        fn()
    ev1.record(); torch.cuda.synchronize()
    print("cap %d  back to back: %-28s %.1f us per iteration" % (cap, name, ev0.elapsed_time(ev1) / 50 * 1e3), flush=True)

big = torch.zeros(65536, device=dev); big8 = torch.zeros(65536, dtype=torch.uint8, device=dev); outb = torch.empty(65536, dtype=torch.bool, device=dev)
for name, fn in (("bare + add_ on 65536 floats", lambda: (raw(), big.add_(1.0))), ("bare + nr_iters.clone()", lambda: (raw(), sim.nr_iters.clone())),
                 ("bare + conv_u8.clone()", lambda: (raw(), sim._conv_u8.clone())), ("bare + other_u8.bool()", lambda: (raw(), big8.bool())),
                 ("bare + torch.ne(conv, 0, out=)", lambda: (raw(), torch.ne(sim._conv_u8, 0, out=outb))),
                 ("bare + conv.view(bool)", lambda: (raw(), sim._conv_u8.view(torch.bool))), ("bare + outb.copy_(conv)", lambda: (raw(), outb.copy_(sim._conv_u8)))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(50):
        fn()
    ev1.record(); torch.cuda.synchronize()
    print("cap %d  back to back: %-32s %.1f us per iteration" % (cap, name, ev0.elapsed_time(ev1) / 50 * 1e3), flush=True)
