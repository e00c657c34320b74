"""GEMM ablations measured by the rocprofv3 kernel trace (kernel durations, no event overhead).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/x -- python3 tools/exp_trace.py run
    python3 tools/exp_trace.py parse /tmp/x
exp bits: 1 no loads after the prologue, 2 no LDS reads / MFMA, 4 no epilogue, 8 no k-skew.
"""
import sys, os, glob, csv, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = 12
SHAPES = [("fc fwd  NT 1600x3072x768 bias+gelu dual bf16", "NT", 1600, 3072, 768, "fc"),
          ("qkv fwd NT 1600x2304x768 bias bf16", "NT", 1600, 2304, 768, "bias"),
          ("proj fwd NT 1600x768x3072 bias+res f32", "NT", 1600, 768, 3072, "res"),
          ("out fwd NT 1600x768x768 bias+res f32", "NT", 1600, 768, 768, "res"),
          ("fc dgrad NN 1600x768x3072 f32", "NN", 1600, 768, 3072, "f32"),
          ("qkv dgrad NN 1600x768x2304 f32", "NN", 1600, 768, 2304, "f32"),
          ("proj dgrad NN 1600x3072x768 act' bf16", "NN", 1600, 3072, 768, "actin"),
          ("out dgrad NN 1600x768x768 bf16", "NN", 1600, 768, 768, "bf16"),
          ("fc wgrad TN 3072x768x1600 f32", "TN", 3072, 768, 1600, "f32"),
          ("proj wgrad TN 768x3072x1600 f32", "TN", 768, 3072, 1600, "f32"),
          ("qkv wgrad TN 2304x768x1600 f32", "TN", 2304, 768, 1600, "f32"),
          ("out wgrad TN 768x768x1600 f32", "TN", 768, 768, 1600, "f32")]
CFGS = {"NT": (14, 16, 43, 30), "NN": (14, 30, 43, 32), "TN": (14, 30, 33, 16)}
EXPS = (0,)

def plan():
    out = []
    for si, sh in enumerate(SHAPES):
        for cfg in CFGS[sh[1]]:
            for e in EXPS:
                if cfg != 14 and e != 0:
                    continue
                out.append((si, cfg, e))
    return out

if sys.argv[1] == "run":
    import torch
    from swat_amd import ops, _lib
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    for si, cfg, e in plan():
        name, lay, m, n, k, epi = SHAPES[si]
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
        lib.swat_gemm_set_tile(cfg); lib.swat_gemm_set_exp(e)
        for _ in range(REPS):
            ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n, **kw)
        if e == 0 and epi in ("bias", "f32") and lay != "TN":
            want = A.float() @ (B.float().t() if lay == "NT" else B.float())
            if "bias" in kw: want = want + kw["bias"]
            err = (C.float() - want).abs().max().item() / want.abs().max().item()
            print(f"check {name} cfg {cfg}: rel err {err:.2e}", flush=True)
        torch.cuda.synchronize()
    lib.swat_gemm_set_tile(-1); lib.swat_gemm_set_exp(0)
else:
    path = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(path))
            if "gemm_" in r["Kernel_Name"]]
    rows.sort()
    durs = [d for _, d in rows]
    pl = plan()
    assert len(durs) == REPS * len(pl), (len(durs), len(pl))
    last = None
    for i, (si, cfg, e) in enumerate(pl):
        d = sorted(durs[i * REPS + 2:(i + 1) * REPS])
        med = d[len(d) // 2] / 1e3
        name, lay, m, n, k, epi = SHAPES[si]
        if si != last:
            print(name); last = si
        tf = 2.0 * m * n * k / med / 1e6
        print(f"    cfg {cfg:2d} exp {e}: {med:6.1f} us" + (f"  ({tf:.0f} TF)" if e == 0 else ""))
