import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import ops, _lib
layout, m, n, k, cfg, exp = [int(x) for x in sys.argv[1:7]]
dev = torch.device("cuda:0")
if layout == 0: A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16(); lda, ldb = k, k
elif layout == 1: A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = k, n
else: A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = m, n
C = torch.empty(m, n, device=dev, dtype=torch.float32)
_lib.lib().swat_gemm_set_tile(cfg); _lib.lib().swat_gemm_set_exp(exp)
for _ in range(5): ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
torch.cuda.synchronize()
