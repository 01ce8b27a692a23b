import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_case30 import run, networks
n30 = networks.synthetic_radial_network(30, 0)
for E in (128, 2048, 4096, 6144, 8192, 12288, 16384, 32768):
    run("case30", n30, E, "radial", n=50)
