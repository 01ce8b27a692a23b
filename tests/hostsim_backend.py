"""Test double loader: the kernel templates compiled for the HOST (tests/hostsim/hostsim.cpp).

Used only by the CPU test tier to exercise the Python host layer and the kernel logic without a
GPU.  The product never imports this module.
"""
import ctypes
import os
import subprocess

from gym_anm_amd import codegen
from gym_anm_amd._lib import Backend

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "hostsim", "_build")
_CACHE = {}


def hostsim_backend(topo):
    name = codegen.topology_name(topo)
    if name in _CACHE:
        return _CACHE[name]
    os.makedirs(OUT, exist_ok=True)
    hdr = os.path.join(OUT, "topo_%s.h" % name)
    text = codegen.emit_header(topo, name)
    if not os.path.exists(hdr) or open(hdr).read() != text:
        open(hdr, "w").write(text)
    lib = os.path.join(OUT, "libhostsim_%s.so" % name)
    srcs = [os.path.join(HERE, "hostsim", "hostsim.cpp")] + [
        os.path.join(codegen.CSRC, f) for f in os.listdir(codegen.CSRC) if f.endswith(".hpp")
    ]
    newest = max(os.path.getmtime(p) for p in srcs + [hdr])
    if not os.path.exists(lib) or os.path.getmtime(lib) < newest:
        cmd = ["g++", "-pthread", "-O2", "-std=c++17", "-fPIC", "-shared", "-ftemplate-depth=4096", "-ffp-contract=off",
               '-DANM_TOPO_HEADER="%s"' % hdr, "-I", os.path.join(ROOT, "include"), srcs[0], "-o", lib]  # fmt: skip
        subprocess.run(cmd, check=True, capture_output=True)
    be = Backend(ctypes.CDLL(lib), "cpu", lib)
    assert be.signature() == codegen.topology_signature(topo)
    _CACHE[name] = be
    return be
