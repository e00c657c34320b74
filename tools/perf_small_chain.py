import os, sys
sys.path.insert(0, "/root/repo")
import torch
from swat_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
for M, N, K in ((32, 200, 512), (32, 768, 512)):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
    def f1(): _lib.check(L.swat_sgemm_small(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), M, N, K, K, 1, 1, K, N, 1.0, 0, _lib.stream()), "s")
    def f2(): _lib.check(L.swat_sim_topk_candidates(_lib.ptr(a), M, _lib.ptr(b), N, K, None, 0, None, 0, None, _lib.ptr(c), N, _lib.stream()), "m")
    for name, f in (("vector chain", f1), ("matrix-core chain", f2)):
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): f()
        e1.record(); torch.cuda.synchronize()
        print(M, N, K, name, round(e0.elapsed_time(e1) * 10, 2), "us")
