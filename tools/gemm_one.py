"""Dev tool: launch ONE GEMM shape under ONE tile configuration n times (to sit under rocprofv3 --pmc / --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import ops, _lib
dev = torch.device("cuda:0"); L = _lib.lib()
M, d = 1600, 768
shapes = {"qkv": (ops.NT, M, 3 * d, d), "fc": (ops.NT, M, 4 * d, d), "proj": (ops.NT, M, d, 4 * d), "out": (ops.NT, M, d, d),
          "dproj": (ops.NN, M, 4 * d, d), "dfc": (ops.NN, M, d, 4 * d), "dout": (ops.NN, M, d, d), "dqkv": (ops.NN, M, d, 3 * d),
          "wproj": (ops.TN, d, 4 * d, M), "wfc": (ops.TN, 4 * d, d, M), "wout": (ops.TN, d, d, M), "wqkv": (ops.TN, 3 * d, d, M)}
names = sys.argv[1].split(","); cfgs = [int(c) for c in sys.argv[2].split(",")]; n_it = int(sys.argv[3]) if len(sys.argv) > 3 else 10
exp = int(sys.argv[4]) if len(sys.argv) > 4 else 0
for name in names:
    layout, m, n, k = shapes[name]
    if layout == ops.NT: A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16(); lda, ldb = k, k
    elif layout == ops.NN: A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = k, n
    else: A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = m, n
    C = torch.empty(m, n, device=dev, dtype=torch.float32 if layout == ops.TN else torch.bfloat16)
    for cfg in cfgs:
        L.swat_gemm_set_tile(cfg); L.swat_gemm_set_exp(exp)
        for _ in range(n_it):
            ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
        torch.cuda.synchronize()
L.swat_gemm_set_tile(-1); L.swat_gemm_set_exp(0)
