"""A transformer block's four weight-gradient products (ViT-B/32, B = 32: 1600 rows): n launches vs grouped launches
(tuning aid for the deferred, grouped weight-gradient pass)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda", 0)
M = 1600
shapes = [(M, 768, 3072, False), (M, 3072, 768, True), (M, 768, 768, False), (M, 2304, 768, True)]
NB = int(os.environ.get("BLOCKS", 1))      # how many blocks' worth of products per measurement
prods = []
for b in range(NB):
    for i, (m, n, k, bias) in enumerate(shapes):
        dy = torch.randn(m, n, device=dev).to(torch.bfloat16); x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        dw = torch.empty(n, k, device=dev); db = torch.empty(n, device=dev) if bias else None
        prods.append((dy, x, dw, db, m, n, k))
ws = torch.empty(lib.swat_colsum_bf16_workspace_bytes(3072), dtype=torch.uint8, device=dev)

def call(group):
    n = len(group)
    VP = ctypes.c_void_p * n; IA = ctypes.c_int * n
    rc = lib.swat_linear_wgrad_bias_group(n, VP(*[p[0].data_ptr() for p in group]), VP(*[p[1].data_ptr() for p in group]),
                                          VP(*[p[2].data_ptr() for p in group]), VP(*[(p[3].data_ptr() if p[3] is not None else None) for p in group]),
                                          IA(*[p[4] for p in group]), IA(*[p[5] for p in group]), IA(*[p[6] for p in group]),
                                          IA(*[p[5] for p in group]), IA(*[p[6] for p in group]), _lib.ptr(ws), _lib.stream())
    _lib.check(rc, "group")

def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for gs in [g for g in (1, 4, 16, 48) if g <= len(prods)]:
    def run():
        for i in range(0, len(prods), gs): call(prods[i:i + gs])
    print(f"{len(prods)} products in groups of {gs}: {timeit(run):8.1f} us  ({timeit(run) / NB:6.1f} us per block)")
