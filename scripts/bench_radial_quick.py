"""case30 lane-group kernel, quick timing (tuning builds via ANM_BUILD_TAG)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_case30 import run, networks
n30 = networks.synthetic_radial_network(30, 0)
print("tag", os.environ.get("ANM_BUILD_TAG", "base"))
run("case30", n30, 16384, "radial")
run("case30", n30, 65536, "radial")
