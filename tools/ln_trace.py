"""LayerNorm backward timing by split count (run under rocprofv3 --kernel-trace; parse prints per-group medians)."""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = 12
if sys.argv[1] == "run":
    import torch
    from swat_amd import _lib
    lib = _lib.lib(); dev = torch.device("cuda:0")
    M, d = 1600, 768
    dy = torch.randn(4, M, d, device=dev); x = torch.randn(M, d, device=dev); g = torch.randn(d, device=dev)
    mean = torch.randn(M, device=dev); rstd = torch.rand(M, device=dev) + 0.5
    dx = torch.randn(M, d, device=dev); dxb = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    dg = torch.empty(d, device=dev); db = torch.empty(d, device=dev); dcs = torch.empty(d, device=dev)
    ws = torch.empty(lib.swat_layernorm_bwd_workspace_bytes(M, d), dtype=torch.uint8, device=dev)
    p = lambda t: t.data_ptr()
    for ns in (1, 2, 3, 4):
        for _ in range(REPS):
            lib.swat_layernorm_bwd_split(p(dy), d, ns, M * d, p(x), d, p(g), p(mean), p(rstd), p(dx), d, p(dx), d, p(dxb), d,
                                         p(dg), p(db), p(dcs), p(ws), M, d, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
else:
    path = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:40])
                  for r in csv.DictReader(open(path)) if "ln_bwd" in r["Kernel_Name"])
    main = [d for _, d, n in rows if "reduce" not in n]; red = [d for _, d, n in rows if "reduce" in n]
    for i, ns in enumerate((1, 2, 3, 4)):
        a = sorted(main[i * REPS + 2:(i + 1) * REPS]); b = sorted(red[i * REPS + 2:(i + 1) * REPS])
        byts = (ns + 3) * 1600 * 768 * 4 + 1600 * 768 * 2
        print(f"nsplit {ns}: ln_bwd {a[len(a)//2]/1e3:6.2f} us ({byts/a[len(a)//2]/1e3:.2f} TB/s)   reduce {b[len(b)//2]/1e3:5.2f} us")
