"""A ViT-B/32 block's four weight gradients at B = 32 (1 664 padded rows): one launch each (csrc/gemm_pc.hip) against ONE launch of
256 x 128 tiles over the joint list (csrc/gemm_wgrad.hip).  Interleaved rounds in one process; operands cycled over NBUF sets so
that every launch finds its operands as cold as the step leaves them."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda", 0)
M = int(os.environ.get("ROWS", 1664))
W, MLP = int(os.environ.get("WIDTH", 768)), int(os.environ.get("MLP", 3072))
shapes = [(M, W, MLP, False), (M, MLP, W, True), (M, W, W, False), (M, 3 * W, W, True)]
NBUF = int(os.environ.get("NBUF", 6))
sets = []
for b in range(NBUF):
    prods = []
    for (m, n, k, bias) in shapes:
        dy = torch.randn(m, n, device=dev).to(torch.bfloat16); x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        prods.append((dy, x, torch.empty(n, k, device=dev), torch.empty(n, device=dev) if bias else None, m, n, k))
    sets.append(prods)
ws = torch.empty(lib.swat_colsum_bf16_workspace_bytes(3 * MLP), dtype=torch.uint8, device=dev)

def call(group):
    n = len(group)
    VP = ctypes.c_void_p * n; IA = ctypes.c_int * n
    _lib.check(lib.swat_linear_wgrad_bias_group(n, VP(*[p[0].data_ptr() for p in group]), VP(*[p[1].data_ptr() for p in group]),
                                                VP(*[p[2].data_ptr() for p in group]), VP(*[(p[3].data_ptr() if p[3] is not None else None) for p in group]),
                                                IA(*[p[4] for p in group]), IA(*[p[5] for p in group]), IA(*[p[6] for p in group]),
                                                IA(*[p[5] for p in group]), IA(*[p[6] for p in group]), _lib.ptr(ws), _lib.stream()), "group")

def timeit(mode, pairs, n=24):
    lib.swat_gemm_set_wgrad_big(mode)
    def run(i):
        s = sets[i % NBUF]
        if pairs:
            call(s[:2]); call(s[2:])
        else:
            call(s)
    run(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): run(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

fl = sum(2.0 * m * n * k for (m, n, k, _) in shapes)
for rnd in range(4):
    a = timeit(0, True)
    lib.swat_gemm_set_exp(4096); c = timeit(2, False); lib.swat_gemm_set_exp(0)
    b = timeit(2, False)
    print(f"round {rnd}: four launches {a:7.1f} us ({fl / a / 1e6:6.0f} TF)   one launch, every wave loads + computes {c:7.1f} us ({fl / c / 1e6:6.0f} TF)   "
          f"loader / consumer waves {b:7.1f} us ({fl / b / 1e6:6.0f} TF)")
for bits, name in ((4, "no epilogue"), (1 | 4, "no epilogue, no DMA"), (2 | 4, "no epilogue, no MFMA"), (256 | 4, "no epilogue, no reads"),
                   (2 | 256 | 4, "no epilogue, no MFMA, no reads"), (1 | 2 | 4, "no epilogue, no DMA, no MFMA"), (1 | 2 | 4 | 256, "launch + ring fill + barriers only")):
    lib.swat_gemm_set_exp(bits)
    print(f"ablation {bits:4d} ({name:28s}): {timeit(2, False):7.1f} us")
import numpy as np
buf = torch.zeros(8 * 512, dtype=torch.int64, device=dev)
lib.swat_gemm_set_trace(buf.data_ptr()); lib.swat_gemm_set_exp(16); lib.swat_gemm_set_wgrad_big(2)
for i in range(4): call(sets[i % NBUF])
torch.cuda.synchronize(); buf.zero_()
call(sets[4 % NBUF]); torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 0] != 0]
s = (t[:, :6] - t[:, 0].min()) / 100.0
med = lambda a: float(np.median(a))
print(f"stamps of {len(t)} workgroups (us): entry spread {s[:,0].max():.2f}; set-up + prologue issue {med(s[:,1]-s[:,0]):.2f}; first k-tile landed {med(s[:,2]-s[:,1]):.2f} "
      f"(max {np.max(s[:,2]-s[:,1]):.2f}); main loop {med(s[:,3]-s[:,2]):.2f} (min {np.min(s[:,3]-s[:,2]):.2f} max {np.max(s[:,3]-s[:,2]):.2f}); epilogue issue {med(s[:,4]-s[:,3]):.2f}; "
      f"store ack {med(s[:,5]-s[:,4]):.2f} (max {np.max(s[:,5]-s[:,4]):.2f}); last workgroup done at {s[:,5].max():.2f}")
lib.swat_gemm_set_trace(None)
lib.swat_gemm_set_exp(0)
lib.swat_gemm_set_wgrad_big(1)
