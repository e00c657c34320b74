"""Path B legs of bench.py on their own (one GPU): the segmented sampler pass and the dense top-k pass on a 1.25 M-row
shard.  tools/collect_profiles.sh runs this under rocprofv3 (--kernel-trace --stats; FETCH_SIZE; WRITE_SIZE) to produce
profiles/<tag>_retrieval_kernel_stats.csv and profiles/<tag>_retrieval_pmc.json.

    python3 tools/perf_retrieval.py [--cpu] [--d 512] [--steps 5]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--cpu", action="store_true", help="also time the CPU baselines (oracle) and check the GPU lists against them")
ap.add_argument("--d", type=int, default=512)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for rec in (bench.retrieval_secondary(dev, d=a.d, steps=a.steps, with_cpu=a.cpu),
            bench.retrieval_dense_secondary(dev, steps=a.steps, with_cpu=a.cpu)):
    print(json.dumps(rec), flush=True)
