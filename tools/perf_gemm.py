"""Dev tool: time the finetuning step's GEMM shapes under every tile configuration (on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import ops, _lib

dev = torch.device("cuda:0")
M, d = 1600, 768
shapes = [  # (name, layout, M, N, K)  in the kernel's own (rows, cols, reduction) terms
    ("fwd qkv", ops.NT, M, 3 * d, d), ("fwd out", ops.NT, M, d, d), ("fwd fc", ops.NT, M, 4 * d, d), ("fwd proj", ops.NT, M, d, 4 * d),
    ("dgrad proj->4d", ops.NN, M, 4 * d, d), ("dgrad fc->d", ops.NN, M, d, 4 * d), ("dgrad out", ops.NN, M, d, d), ("dgrad qkv->d", ops.NN, M, d, 3 * d),
    ("wgrad proj", ops.TN, d, 4 * d, M), ("wgrad fc", ops.TN, 4 * d, d, M), ("wgrad out", ops.TN, d, d, M), ("wgrad qkv", ops.TN, 3 * d, d, M),
    ("patch", ops.NT, 1568, d, 3072), ("wgrad conv", ops.TN, d, 3072, 1568),
]
cfgs = {14: "d64x64s3", 30: "R64x64d4", 50: "P128x128", 51: "P160x128", 52: "P160x96", 53: "P96x64", 54: "P96x96", 55: "P128x64", 56: "P64x64", 57: "P192x128",
        58: "P128sq8c", 59: "P128sqL2", 60: "P128sqL8", 61: "P128sqS5", 62: "P80x64", 63: "P48x48"}
if len(sys.argv) > 1:
    cfgs = {int(c): str(c) for c in sys.argv[1].split(",")}
print(f"{'shape':18s} " + " ".join(f"{v:>11s}" for v in cfgs.values()) + "   (TFLOP/s)")
for name, layout, m, n, k in shapes:
    if layout == ops.NT:
        A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16(); lda, ldb = k, k
    elif layout == ops.NN:
        A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = k, n
    else:
        A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = m, n
    C = torch.empty(m, n, device=dev, dtype=torch.float32)
    row = []
    for cfg in cfgs:
        _lib.lib().swat_gemm_set_tile(cfg)
        for _ in range(3):
            ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
        torch.cuda.synchronize()
        import ctypes
        _lib.lib().swat_gemm_prof_begin(64)
        for _ in range(20):
            ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
        torch.cuda.synchronize()
        ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)(); fl = (ctypes.c_double * 6)()
        _lib.lib().swat_gemm_prof_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p))
        us = sum(ms) / 20 * 1e3
        row.append(2.0 * m * n * k / us / 1e6)
    _lib.lib().swat_gemm_set_tile(-1)
    print(f"{name:18s} " + " ".join(f"{v:11.0f}" for v in row))
