#!/usr/bin/env python3
"""tools/dense_cold.py — a few COLD dense top-k calls (new query tensor each) on a 1.25 M x 512 shard, for a kernel trace of one call:
rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/dense_cold.py ; python3 tools/trace_window.py <dir> 0.97 900"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swat_amd.retrieval import dense  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
n, d, nq, k = 1_250_000, 512, 200, 500
db = torch.randn(n, d, device=dev, generator=g)
db /= db.norm(dim=1, keepdim=True)
sh = dense.DenseShadow(db)
qs = []
for i in range(12):
    q = torch.randn(nq, d, device=dev, generator=g)
    qs.append(q / q.norm(dim=1, keepdim=True))
for q in qs[:4]:
    dense.topk_dense(db, q, k, shadow=sh)
torch.cuda.synchronize()
t0 = time.perf_counter()
for q in qs[4:]:
    dense.topk_dense(db, q, k, shadow=sh)
torch.cuda.synchronize()
print("cold call %.3f ms" % (1e3 * (time.perf_counter() - t0) / 8))
