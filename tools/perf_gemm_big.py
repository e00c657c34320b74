"""Dev tool: the many-row products of ViT-L/14 (M = 8 224) and ViT-B/16 (M = 6 304) under the 256 x 256 tile (cfg 80,
csrc/gemm_big.hip) against the tile pick_cfg takes by default (the 256 x 256 tile is opt-in: swat_gemm_set_exp bit 2048)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import ops, _lib

dev = torch.device("cuda:0")
L = _lib.lib()


def run(layout, m, n, k, epi, cfg):
    if layout == ops.NT:
        A = torch.randn(m, k, device=dev).bfloat16(); B = (torch.randn(n, k, device=dev) * 0.05).bfloat16(); lda, ldb = k, k
    elif layout == ops.TN:
        A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = m, n
    else:
        A = torch.randn(m, k, device=dev).bfloat16(); B = (torch.randn(k, n, device=dev) * 0.05).bfloat16(); lda, ldb = k, n
    kw = {}
    C = torch.empty(m, n, device=dev, dtype=torch.float32 if epi in ("res", "f32") else torch.bfloat16)
    if epi in ("fc", "bias", "res"):
        kw["bias"] = torch.randn(n, device=dev)
    if epi == "fc":
        kw["C2"] = torch.empty(m, n, device=dev, dtype=torch.bfloat16); kw["act"] = "gelu"
    if epi == "res":
        C.normal_(); kw["residual"] = C
    if epi == "actin":
        kw["actin"] = torch.randn(m, n, device=dev).bfloat16(); kw["ldact"] = n; kw["act"] = "gelu"
    L.swat_gemm_set_tile(cfg)
    for _ in range(3):
        ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n, **kw)
    torch.cuda.synchronize()
    L.swat_gemm_prof_begin(64)
    for _ in range(10):
        ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n, **kw)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_int64 * 16)(); fl = (ctypes.c_double * 16)()
    L.swat_gemm_prof_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p))
    L.swat_gemm_set_tile(-1)
    return sum(ms) / 10 * 1e3


for arch, M, d in (("ViT-L/14", 8224, 1024), ("ViT-B/16", 6304, 768)):
    Mp = (M + 63) // 64 * 64
    shapes = [("fwd qkv", ops.NT, M, 3 * d, d, "bias"), ("fwd out", ops.NT, M, d, d, "res"), ("fwd fc", ops.NT, M, 4 * d, d, "fc"),
              ("fwd proj", ops.NT, M, d, 4 * d, "res"), ("dgrad proj->4d", ops.NN, M, 4 * d, d, "actin"), ("dgrad fc->d", ops.NN, M, d, 4 * d, "bf16"),
              ("dgrad out", ops.NN, M, d, d, "bf16"), ("dgrad qkv->d", ops.NN, M, d, 3 * d, "bf16"),
              ("wgrad proj", ops.TN, d, 4 * d, Mp, "f32"), ("wgrad fc", ops.TN, 4 * d, d, Mp, "f32"), ("wgrad out", ops.TN, d, d, Mp, "f32"),
              ("wgrad qkv", ops.TN, 3 * d, d, Mp, "f32")]
    print(f"{arch}: M = {M}, d = {d}   (us per launch, TFLOP/s)")

    t_old = t_new = 0.0
    for name, layout, m, n, k, epi in shapes:
        old = run(layout, m, n, k, epi, -1)
        new = run(layout, m, n, k, epi, 80) if layout != ops.TN else float("nan")
        fl = 2.0 * m * n * k
        t_old += old; t_new += new if new == new else old
        extra = ""
        for c in [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else []:
            try:
                t = run(layout, m, n, k, epi, c)
                extra += f"   cfg{c} {t:7.1f} us {fl / t / 1e6:5.0f} TF"
            except Exception:
                L.swat_gemm_set_tile(-1)
                extra += f"   cfg{c}       -          "
        print(f"  {name:16s} before {old:8.1f} us {fl / old / 1e6:6.0f} TF   256x256 {new:8.1f} us {fl / new / 1e6:6.0f} TF" + extra)
    print(f"  sum: {t_old:.0f} -> {t_new:.0f} us per block")
