"""Same-box A/B of whole TREES: run from the root of a tree (the repository, or a worktree of an earlier commit under
_ab/<name> whose ANM6 library was built there): prints the launch time of the workloads named on the command line.

    python scripts/ab_trees.py mesh30 mesh200 case30 case30_20

mesh30 / mesh200: the general lane-group kernel (bench.mesh_side_figure's two workloads); case30 / case30_20: config 4 at the
reference's cap and at 20.  scripts/ab_trees.sh interleaves the trees."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench
from gym_anm_amd import networks

dev = torch.device("cuda", 0)
for what in sys.argv[1:]:
    if what == "mesh30":
        f = bench.transition_figure(dev, networks.synthetic_meshed_network(30, 6, 4), 16384, 100, 40)
    elif what == "mesh200":
        f = bench.transition_figure(dev, networks.synthetic_meshed_network(200, 13, 30), 4096, 100, 10, 40.0 / 200)
    elif what == "case30":
        f = bench.transition_figure(dev, networks.synthetic_radial_network(30, 0), 16384, 100, 40)
    elif what == "case30_20":
        f = bench.transition_figure(dev, networks.synthetic_radial_network(30, 0), 16384, 20, 40)
    else:
        raise SystemExit("unknown workload " + what)
    print("%-10s %8.1f us  (%s, %d lanes)" % (what, f["us_per_launch"], f["impl"], f["lanes_per_env"]), flush=True)
