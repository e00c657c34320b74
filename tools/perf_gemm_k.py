import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.lib()
def t(layout, m, n, k, cfg, exp=0):
    if layout == ops.NT: A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16(); lda, ldb = k, k
    elif layout == ops.NN: A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = k, n
    else: A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = m, n
    C = torch.empty(m, n, device=dev, dtype=torch.float32)
    lib.swat_gemm_set_tile(cfg); lib.swat_gemm_set_exp(exp)
    for _ in range(3): ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
    torch.cuda.synchronize()
    lib.swat_gemm_prof_begin(64)
    for _ in range(20): ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)(); fl = (ctypes.c_double * 6)()
    lib.swat_gemm_prof_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p))
    lib.swat_gemm_set_tile(-1); lib.swat_gemm_set_exp(0)
    return sum(ms) / 20 * 1e3
for (m, n) in [(1600, 3072), (1600, 768), (256, 256)]:
    for cfg in (6, 9):
        print(f"M={m} N={n} cfg{cfg}: " + "  ".join(f"K={k}: {t(ops.NT, m, n, k, cfg):6.1f}us" for k in (64, 128, 256, 768, 1536, 3072)) + f"   | K=768 noepi {t(ops.NT, m, n, 768, cfg, 4):6.1f}")
