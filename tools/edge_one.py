"""Dev tool: a sequence of dense top-k calls (d n nq k, ...) in one process, each checked against the exact scan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd.retrieval import dense
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
a = [int(x) for x in sys.argv[1:]]
for i in range(0, len(a), 4):
    d, n, nq, k = a[i:i + 4]
    print(d, n, nq, k, flush=True)
    db = torch.randn(n, d, device=dev, generator=g); db /= db.norm(dim=-1, keepdim=True)
    q = torch.randn(nq, d, device=dev, generator=g); q /= q.norm(dim=-1, keepdim=True)
    sh = dense.DenseShadow(db) if os.environ.get("SWAT_DENSE_SHADOW", "1") != "0" else None
    i1, v1, c1 = dense.topk_dense(db, q, k, row_base=7, shadow=sh)
    torch.cuda.synchronize()
    print("  fast ok", dense.last_path, flush=True)
    i2, v2, c2 = dense.topk_dense_exact(db, q, k, row_base=7)
    torch.cuda.synchronize()
    print("  exact ok", torch.equal(i1, i2) and torch.equal(v1, v2) and torch.equal(c1, c2), flush=True)
