"""Dev tool: where the waves of the shadow scan spend their cycles (SWAT_DENSE_PROF=1 build: s_memtime sums per wave)."""
import os, sys
os.environ["SWAT_DENSE_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch
from swat_amd import _lib
from swat_amd.retrieval import dense
dev = torch.device("cuda:0"); L = _lib.lib()
n, d, k = 1_250_000, 512, 500
g = torch.Generator(device=dev).manual_seed(7)
db = torch.randn(n, d, device=dev, generator=g); db /= db.norm(dim=-1, keepdim=True)
SH = dense.DenseShadow(db)
for nq in (16, 100, 200):
    q = torch.randn(nq, d, device=dev, generator=g); q /= q.norm(dim=-1, keepdim=True)
    for _ in range(3): dense.topk_dense(db, q, k, shadow=SH)
    c = dense._cache
    c["cnt"].zero_(); dense.candidates(c, db, shadow=SH)
    out = np.zeros((256, 8, 5), np.uint64)
    L.swat_dense_debug_profile.argtypes = [ctypes.c_void_p]
    assert L.swat_dense_debug_profile(out.ctypes.data_as(ctypes.c_void_p)) == 0
    o = out.astype(np.float64)
    tot = o[..., 4].mean()
    names = ("counted wait", "barrier", "mfma phase", "epilogue")
    print(f"{nq} queries: kernel {tot:,.0f} cycles per wave (min {o[..., 4].min():,.0f}, max {o[..., 4].max():,.0f}); " +
          "; ".join(f"{nm} {o[..., i].mean() / tot:.3f}" for i, nm in enumerate(names)) +
          f"; unaccounted {1 - o[..., :4].sum(-1).mean() / tot:.3f}")
    per_wave = o.mean(0)
    print("   per wave of the workgroup (wait / barrier / mfma / epilogue, % of kernel): " +
          "  ".join("/".join(f"{per_wave[w, i] / per_wave[w, 4] * 100:.0f}" for i in range(4)) for w in range(8)))
