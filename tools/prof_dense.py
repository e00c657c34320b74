"""The dense retrieval pass of bench.py alone (run under rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
print(bench.retrieval_dense_secondary(torch.device("cuda", 0), steps=5))
