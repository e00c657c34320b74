"""Dev tool: the COLD dense top-k call (a query tensor the library has not seen: thresholds from a sample scan, query packing,
|q| max) — what a retrieval job pays per class-query batch.  Run under rocprofv3 --stats for the kernel list."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd.retrieval import dense
dev = torch.device("cuda:0")
n, d, nq, k = 1_250_000, 512, 200, 500
g = torch.Generator(device=dev).manual_seed(7)
db = torch.randn(n, d, device=dev, generator=g); db /= db.norm(dim=-1, keepdim=True)
sh = dense.DenseShadow(db) if os.environ.get("SWAT_DENSE_SHADOW", "1") != "0" else None
qs = []
for i in range(24):
    q = torch.randn(nq, d, device=dev, generator=g); q /= q.norm(dim=-1, keepdim=True); qs.append(q)
for q in qs[:4]:
    dense.topk_dense(db, q, k, shadow=sh)
torch.cuda.synchronize(); t0 = time.perf_counter()
for q in qs[4:]:
    dense.topk_dense(db, q, k, shadow=sh)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 20
print(f"cold call {t * 1e3:.3f} ms = {nq / t:,.0f} queries/s  ({dense.last_path})")
