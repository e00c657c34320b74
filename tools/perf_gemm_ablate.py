"""Dev tool: one GEMM shape x tile configuration under the tuning ablations (exp bits) — where does the time go?"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import ops, _lib

dev = torch.device("cuda:0")
L = _lib.lib()
M, d = 1600, 768
shapes = {"qkv": (ops.NT, M, 3 * d, d), "fc": (ops.NT, M, 4 * d, d), "proj": (ops.NT, M, d, 4 * d), "out": (ops.NT, M, d, d),
          "dproj": (ops.NN, M, 4 * d, d), "dfc": (ops.NN, M, d, 4 * d), "wfc": (ops.TN, 4 * d, d, M), "wqkv": (ops.TN, 3 * d, d, M)}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["qkv"]
cfgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [50]
exps = [0, 8, 4, 2, 2 | 4, 1, 1 | 4, 1 | 2 | 4]
def timeit(layout, A, B, m, n, k, lda, ldb, C):
    for _ in range(3):
        ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
    torch.cuda.synchronize()
    L.swat_gemm_prof_begin(64)
    for _ in range(20):
        ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_int64 * 16)(); fl = (ctypes.c_double * 16)()
    L.swat_gemm_prof_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p))
    return sum(ms) / 20 * 1e3
print("exp bits: 1 = no loads after the prologue, 2 = no LDS reads / MFMA, 4 = no epilogue, 8 = no k-skew; times in us")
for name in names:
    layout, m, n, k = shapes[name]
    if layout == ops.NT:
        A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16(); lda, ldb = k, k
    elif layout == ops.NN:
        A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = k, n
    else:
        A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = m, n
    for odt in (torch.float32, torch.bfloat16):
        C = torch.empty(m, n, device=dev, dtype=odt)
        for cfg in cfgs:
            L.swat_gemm_set_tile(cfg)
            row = []
            for e in exps:
                L.swat_gemm_set_exp(e)
                row.append(timeit(layout, A, B, m, n, k, lda, ldb, C))
            L.swat_gemm_set_exp(0)
            print(f"{name:6s} out={str(odt)[6:]:9s} cfg {cfg:3d}: " + "  ".join(f"e{e}={t:6.2f}" for e, t in zip(exps, row)), flush=True)
    L.swat_gemm_set_tile(-1)
