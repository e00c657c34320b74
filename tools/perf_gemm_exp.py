import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.lib()
def t(layout, m, n, k, cfg, exp, swz=-1):
    if layout == ops.NT: A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16(); lda, ldb = k, k
    elif layout == ops.NN: A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = k, n
    else: A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = m, n
    C = torch.empty(m, n, device=dev, dtype=torch.float32)
    lib.swat_gemm_set_tile(cfg); lib.swat_gemm_set_exp(exp); lib.swat_gemm_set_swizzle(swz)
    for _ in range(3): ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
    torch.cuda.synchronize()
    lib.swat_gemm_prof_begin(64)
    for _ in range(20): ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)(); fl = (ctypes.c_double * 6)()
    lib.swat_gemm_prof_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p))
    lib.swat_gemm_set_tile(-1); lib.swat_gemm_set_exp(0); lib.swat_gemm_set_swizzle(-1)
    return sum(ms) / 20 * 1e3
for name, layout, m, n, k in [("fwd fc", ops.NT, 1600, 3072, 768), ("fwd proj", ops.NT, 1600, 768, 3072), ("wgrad fc", ops.TN, 3072, 768, 1600)]:
    for cfg in (14, 15, 16):
        full = t(layout, m, n, k, cfg, 0); dma = t(layout, m, n, k, cfg, 2); nosk = t(layout, m, n, k, cfg, 8); noskd = t(layout, m, n, k, cfg, 10)
        print(f"{name:10s} cfg{cfg}: skew: full {full:6.1f}us dma-only {dma:6.1f}us | no skew: full {nosk:6.1f} dma-only {noskd:6.1f}   ({2.0*m*n*k/full/1e6:.0f} TF)")
