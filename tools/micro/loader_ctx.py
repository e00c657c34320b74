"""Dev experiment: the bench's loader leg alone, and after the bench's CPU-baseline legs have run in the same process."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
r = bench.decode_secondary(dev, with_cpu=False)
print("alone:", r["loader"])
if os.environ.get("CPU", "1") == "1":
    bench.cpu_baseline(32)
    print("after cpu_baseline (torch threads %d):" % torch.get_num_threads(), bench.decode_secondary(dev, with_cpu=False)["loader"])
