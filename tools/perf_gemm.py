"""Dev tool: time the finetuning step's GEMM shapes, with the epilogues the step uses, under tile configurations."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import ops, _lib

dev = torch.device("cuda:0"); L = _lib.lib()
M, d = int(os.environ.get("SWAT_PG_M", 1600)), int(os.environ.get("SWAT_PG_D", 768))   # ViT-L/14: 8224 1024; ViT-B/16: 6304 768
shapes = [  # (name, layout, M, N, K, epilogue)  in the kernel's own (rows, cols, reduction) terms
    ("fwd qkv", ops.NT, M, 3 * d, d, "bias"), ("fwd out", ops.NT, M, d, d, "res"), ("fwd fc", ops.NT, M, 4 * d, d, "fc"), ("fwd proj", ops.NT, M, d, 4 * d, "res"),
    ("dgrad proj->4d", ops.NN, M, 4 * d, d, "actin"), ("dgrad fc->d", ops.NN, M, d, 4 * d, "bf16"), ("dgrad out", ops.NN, M, d, d, "bf16"), ("dgrad qkv->d", ops.NN, M, d, 3 * d, "bf16"),
    ("wgrad proj", ops.TN, d, 4 * d, M, "f32"), ("wgrad fc", ops.TN, 4 * d, d, M, "f32"), ("wgrad out", ops.TN, d, d, M, "f32"), ("wgrad qkv", ops.TN, 3 * d, d, M, "f32"),
    ("patch", ops.NT, 1568, d, 3072, "f32"),
]
PC_LAYOUTS = {50: (0, 1, 2), 51: (0, 1), 53: (0, 1, 2), 54: (2,), 55: (0, 1, 2), 56: (0, 1, 2), 58: (0, 1), 60: (0, 1), 61: (2,), 62: (0, 1, 2), 63: (0,), 64: (0, 1), 65: (0, 1)}
cfgs = [14, 30, 43, 33, 50, 51, 53, 54, 55, 56, 58]
if len(sys.argv) > 1:
    cfgs = [int(c) for c in sys.argv[1].split(",")]
print(f"{'shape':18s} " + " ".join(f"{c:>7d}" for c in cfgs) + "   (us per launch; * = best)")
tot_best = 0.0
for name, layout, m, n, k, epi in shapes:
    if layout == ops.NT: A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16(); lda, ldb = k, k
    elif layout == ops.NN: A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = k, n
    else: A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = m, n
    kw = {}
    C = torch.empty(m, n, device=dev, dtype=torch.float32 if epi in ("res", "f32") else torch.bfloat16)
    if epi in ("fc", "bias", "res"): kw["bias"] = torch.randn(n, device=dev)
    if epi == "fc": kw["C2"] = torch.empty(m, n, device=dev, dtype=torch.bfloat16); kw["act"] = "quick_gelu"
    if epi == "res": C.normal_(); kw["residual"] = C
    if epi == "actin": kw["actin"] = torch.randn(m, n, device=dev).bfloat16(); kw["ldact"] = n; kw["act"] = "quick_gelu"
    row = []
    for cfg in cfgs:
        if cfg >= 50 and layout not in PC_LAYOUTS.get(cfg, (0, 1, 2)):
            row.append(float("nan")); continue
        L.swat_gemm_set_tile(cfg)
        for _ in range(3):
            ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n, **kw)
        torch.cuda.synchronize()
        L.swat_gemm_prof_begin(64)
        for _ in range(20):
            ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n, **kw)
        torch.cuda.synchronize()
        ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_int64 * 16)(); fl = (ctypes.c_double * 16)()
        L.swat_gemm_prof_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p))
        row.append(sum(ms) / 20 * 1e3)
    L.swat_gemm_set_tile(-1)
    best = min(v for v in row if v == v)
    tot_best += best
    print(f"{name:18s} " + " ".join((f"{v:6.1f}" + ("*" if v == best else " ")) if v == v else "    -  " for v in row) +
          f"   best {2.0 * m * n * k / best / 1e6:5.0f} TF")
print(f"sum of best (12 block GEMMs + patch): {tot_best:.1f} us")
