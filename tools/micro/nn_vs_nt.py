"""Dev experiment: the c_proj data gradient dF = (dx W) * act'(F) as the NN product the step runs against the same product as NT over a
transposed copy of W (K contiguous in both operands), default dispatch, at the ViT-B/16 / ViT-L/14 / ViT-B/32 row counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from swat_amd import ops, _lib
dev = torch.device("cuda:0"); L = _lib.lib()
def t(fn, n=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M, d in ((6304, 768), (8224, 1024), (1600, 768)):
    mlp = 4 * d
    dx = torch.randn(M, d, device=dev).bfloat16(); W = (torch.randn(d, mlp, device=dev) * 0.02).bfloat16(); Wt = W.t().contiguous()
    F = torch.randn(M, mlp, device=dev).bfloat16(); o1 = torch.empty(M, mlp, device=dev, dtype=torch.bfloat16); o2 = torch.empty_like(o1)
    nn = lambda: ops.gemm(ops.NN, dx, W, M, mlp, d, d, mlp, o1, mlp, actin=F, ldact=mlp, act="quick_gelu")
    nt = lambda: ops.gemm(ops.NT, dx, Wt, M, mlp, d, d, d, o2, mlp, actin=F, ldact=mlp, act="quick_gelu")
    a, b = t(nn), t(nt)
    print(f"M={M} d={d}: NN {a:.1f} us, NT over W^T {b:.1f} us; equal bits: {torch.equal(o1, o2)}; max |diff| {(o1.float() - o2.float()).abs().max().item():.3e}")
    # the other data gradients for reference (K = 4 d, 3 d, d; N = d)
    for K in (mlp, 3 * d, d):
        dy = torch.randn(M, K, device=dev).bfloat16(); W2 = (torch.randn(K, d, device=dev) * 0.02).bfloat16(); W2t = W2.t().contiguous()
        o = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
        a = t(lambda: ops.gemm(ops.NN, dy, W2, M, d, K, K, d, o, d)); b = t(lambda: ops.gemm(ops.NT, dy, W2t, M, d, K, K, K, o, d))
        print(f"      N={d} K={K}: NN {a:.1f} us, NT {b:.1f} us")
