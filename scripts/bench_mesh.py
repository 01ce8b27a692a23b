"""Meshed 30-bus network (random feeder + 4 chords, line charging, a phase shifter), 16384 environments, one
Simulator.transition per launch: general lane-group kernel ("mesh", generic mode) vs the thread-per-environment
kernels compiled for that topology (the round-1 path for meshed networks above 12 buses)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_case30 import run
from gym_anm_amd import networks
net = networks.synthetic_meshed_network(30, 6, 4)
for E in (16384, 65536):
    run("mesh30", net, E, "mesh")
run("mesh30", net, 16384, "mesh", "f32")
if "--thread" in sys.argv:
    run("mesh30", net, 16384, "thread", n=3)
run("case30", networks.synthetic_radial_network(30, 0), 16384, "mesh")
run("anm6", networks.anm6_network(), 65536, "mesh")
