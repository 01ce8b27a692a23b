"""bench.mixed_side_figure alone (four topologies, 16 384 environments, per-topology streams, then one stream): the workload of
a kernel-sequence trace.  usage: python scripts/mixed_workload.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import mixed_side_figure
print(json.dumps(mixed_side_figure(torch.device("cuda", 0))))
