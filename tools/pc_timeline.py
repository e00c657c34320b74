"""Dev tool: per-workgroup time stamps of the producer / consumer GEMM (exp bit 16) against the launch's event time."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swat_amd import ops, _lib
dev = torch.device("cuda:0"); L = _lib.lib()
M, d = 1600, 768
shapes = {"qkv": (ops.NT, M, 3 * d, d), "fc": (ops.NT, M, 4 * d, d), "proj": (ops.NT, M, d, 4 * d), "out": (ops.NT, M, d, d),
          "dproj": (ops.NN, M, 4 * d, d), "dfc": (ops.NN, M, d, 4 * d), "wfc": (ops.TN, 4 * d, d, M), "wqkv": (ops.TN, 3 * d, d, M)}
PAD = int(os.environ.get('PAD', '0'))
def run(name, cfg, exp, odt=torch.bfloat16):
    layout, m, n, k = shapes[name]
    if layout == ops.NT: A = torch.randn(m, k + PAD, device=dev).bfloat16(); B = torch.randn(n, k + PAD, device=dev).bfloat16(); lda, ldb = k + PAD, k + PAD
    elif layout == ops.NN: A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = k, n
    else: A = torch.randn(k, m, device=dev).bfloat16(); B = torch.randn(k, n, device=dev).bfloat16(); lda, ldb = m, n
    C = torch.empty(m, n, device=dev, dtype=odt)
    buf = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
    L.swat_gemm_set_trace(buf.data_ptr()); L.swat_gemm_set_tile(cfg); L.swat_gemm_set_exp(16 | exp)
    for _ in range(4):
        ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
    torch.cuda.synchronize(); buf.zero_()
    L.swat_gemm_prof_begin(4)
    ops.gemm(layout, A, B, m, n, k, lda, ldb, C, n)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_int64 * 16)(); fl = (ctypes.c_double * 16)()
    L.swat_gemm_prof_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p))
    L.swat_gemm_set_tile(-1); L.swat_gemm_set_exp(0); L.swat_gemm_set_trace(None)
    t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 0] != 0]
    t0 = t[:, 0].min(); s = (t[:, :6] - t0) / 100.0
    med = lambda a: float(np.median(a))
    print(f"{name} cfg {cfg} exp {exp} out {str(odt)[6:]}: event {sum(ms) * 1e3:6.2f} us | {len(t)} WGs; entry spread {s[:,0].max():.2f}; "
          f"setup {med(s[:,1]-s[:,0]):.2f}; to barrier0 {med(s[:,2]-s[:,1]):.2f} (max {np.max(s[:,2]-s[:,1]):.2f}); main loop {med(s[:,3]-s[:,2]):.2f} (max {np.max(s[:,3]-s[:,2]):.2f}); "
          f"loop cycles {med(t[:,7]):.0f} / ticks {med(t[:,3]-t[:,2]):.0f}; epilogue issue {med(s[:,4]-s[:,3]):.2f}; store ack {med(s[:,5]-s[:,4]):.2f} (max {np.max(s[:,5]-s[:,4]):.2f}); last WG done at {s[:,5].max():.2f}")
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["qkv"]
cfgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [50]
for nme in names:
    for cfg in cfgs:
        for exp in [int(e) for e in os.environ.get('EXPS', '0,4,5,6').split(',')]:
            run(nme, cfg, exp)
