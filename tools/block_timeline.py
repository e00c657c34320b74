"""Per-workgroup timeline of one GEMM launch (exp bit 16): rounds, main-loop vs epilogue time, CU residency."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swat_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.lib()
def run(name, lay, m, n, k, epi, cfg, exp=0):
    layout = {"NT": ops.NT, "NN": ops.NN, "TN": ops.TN}[lay]
    if lay == "NT": A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16(); lda, ldb = k, k
    elif lay == "NN": A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = k, n
    else: A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = m, n
    kw = {}
    C = torch.empty(m, n, device=dev, dtype=torch.float32 if epi in ("res", "f32") else torch.bfloat16)
    if epi in ("fc", "bias", "res"): kw["bias"] = torch.randn(n, device=dev)
    if epi == "fc": kw["C2"] = torch.empty(m, n, device=dev, dtype=torch.bfloat16); kw["act"] = "quick_gelu"
    if epi == "res": kw["residual"] = C
    if epi == "actin": kw["actin"] = torch.randn(m, n, device=dev).bfloat16(); kw["ldact"] = n; kw["act"] = "quick_gelu"
    buf = torch.zeros(4 * 8192, dtype=torch.int64, device=dev)
    lib.swat_gemm_set_trace(buf.data_ptr()); lib.swat_gemm_set_tile(cfg); lib.swat_gemm_set_exp(16 | exp)
    for _ in range(5):
        buf.zero_()
        ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n, **kw)
    torch.cuda.synchronize()
    lib.swat_gemm_set_tile(-1); lib.swat_gemm_set_exp(0); lib.swat_gemm_set_trace(None)
    t = buf.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    st, ml, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0  # us
    hw = t[:, 3] & 0xffffffff; xcc = (t[:, 3] >> 32) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (xcc << 7)
    print(f"{name} cfg {cfg} exp {exp}: {len(t)} workgroups on {len(np.unique(cu))} CUs, span {en.max():.1f} us")
    print(f"   start times  : " + " ".join(f"{np.sum((st >= a) & (st < a + 2)):4d}" for a in range(0, 30, 2)) + "   (per 2 us bin)")
    print(f"   end times    : " + " ".join(f"{np.sum((en >= a) & (en < a + 2)):4d}" for a in range(0, 30, 2)))
    first = st < 1.5
    for lab, sel in (("first round", first), ("later rounds", ~first)):
        if sel.sum():
            print(f"   {lab:12s}: n={sel.sum():4d}  main loop {np.median(ml[sel] - st[sel]):5.2f} us (p90 {np.percentile(ml[sel] - st[sel], 90):5.2f})"
                  f"  epilogue {np.median(en[sel] - ml[sel]):5.2f} us (p90 {np.percentile(en[sel] - ml[sel], 90):5.2f})")
    ids = np.nonzero(buf.cpu().numpy().reshape(-1, 4)[:, 0])[0]
    print("   xcc of the first 24 workgroups:", xcc[:24].tolist(), " fraction with xcc == id % 8:", float(np.mean(xcc == ids % 8)))
    per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
    print(f"   workgroups per CU: min {per_cu.min()} max {per_cu.max()}  hist {np.bincount(per_cu)}")
run("fc fwd NT 1600x3072x768 dual", "NT", 1600, 3072, 768, "fc", 14)
run("fc fwd NT 1600x3072x768 dual", "NT", 1600, 3072, 768, "fc", 43)
run("qkv fwd NT 1600x2304x768", "NT", 1600, 2304, 768, "bias", 43)
run("proj fwd NT 1600x768x3072 res", "NT", 1600, 768, 3072, "res", 14)
run("proj dgrad NN 1600x3072x768 act'", "NN", 1600, 3072, 768, "actin", 43)
run("fc dgrad NN 1600x768x3072", "NN", 1600, 768, 3072, "f32", 14)
run("fc wgrad TN 3072x768x1600", "TN", 3072, 768, 1600, "f32", 14)
