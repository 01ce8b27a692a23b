"""General lane-group kernel, wavefronts per compute unit: Simulator.transition launches of meshed networks (cap 100)
under whatever ANM_MESH_SIMD_WAVES / ANM_MESH_TABLES / ANM_MESH_WAVES / ANM_BUILD_TAG the environment names (ANM_IMPL=mesh puts ANM6 there).
usage: python scripts/mesh_occupancy_bench.py [anm6 mesh12 mesh20 mesh30 mesh50 mesh64 mesh200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import transition_figure
from gym_anm_amd import networks

NETS = {"anm6": (networks.anm6_network, 65536, 1.0), "mesh12": (lambda: networks.synthetic_meshed_network(12, 3, 3), 32768, 1.0),
        "mesh20": (lambda: networks.synthetic_meshed_network(20, 3, 6), 16384, 1.0), "mesh30": (lambda: networks.synthetic_meshed_network(30, 6, 4), 16384, 1.0),
        "mesh50": (lambda: networks.synthetic_meshed_network(50, 5, 10), 8192, 1.0), "mesh64": (lambda: networks.synthetic_meshed_network(64, 9, 20), 8192, 1.0), "mesh200": (lambda: networks.synthetic_meshed_network(200, 13, 30), 4096, 40.0 / 200)}
dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["mesh20", "mesh30", "mesh64"]):
    mk, E, scale = NETS[name]
    r = transition_figure(dev, mk(), E, 100, 20, scale)
    print("%s simd_waves=%s waves=%s: %.1f us per launch of %d (impl %s, converged %.4f)" % (
        name, os.environ.get("ANM_MESH_SIMD_WAVES", "auto"), os.environ.get("ANM_MESH_WAVES", "auto"), r["us_per_launch"], E, r["impl"], r["converged_frac"]), flush=True)
