"""Dev tool: what the vendor library (torch.mm -> hipBLASLt / rocBLAS) reaches on the step's plain GEMM shapes — a yardstick for
the hand-written kernels (tools/perf_gemm.py), not a product path."""
import os, sys
import torch
dev = torch.device("cuda:0")
M, d = int(os.environ.get("SWAT_PG_M", 8224)), int(os.environ.get("SWAT_PG_D", 1024))
shapes = [("fwd qkv", "nt", M, 3 * d, d), ("fwd out", "nt", M, d, d), ("fwd fc", "nt", M, 4 * d, d), ("fwd proj", "nt", M, d, 4 * d),
          ("dgrad proj->4d", "nn", M, 4 * d, d), ("dgrad fc->d", "nn", M, d, 4 * d), ("dgrad out", "nn", M, d, d), ("dgrad qkv->d", "nn", M, d, 3 * d),
          ("wgrad proj", "tn", d, 4 * d, M), ("wgrad fc", "tn", 4 * d, d, M), ("wgrad out", "tn", d, d, M), ("wgrad qkv", "tn", 3 * d, d, M)]
tot = 0.0
for name, lay, m, n, k in shapes:
    if lay == "nt": A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16(); f = lambda: A @ B.t()
    elif lay == "nn": A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); f = lambda: A @ B
    else: A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); f = lambda: A.t() @ B
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    print(f"{name:16s} {m:6d} x {n:5d} x {k:5d}  {us:8.1f} us  {2.0 * m * n * k / us / 1e6:6.0f} TF")
print(f"sum {tot:.1f} us")
