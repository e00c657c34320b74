"""Dev check: the dense top-k with more queries than one pass of the scan takes (209 / 450 / 1000: two, three and five passes of 13, 8 or 2
query tiles) against the exact scan, at d = 512 and 768."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from swat_amd.retrieval import dense
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(11)
bad = 0
for d in (512, 768):
    n = 300_000
    db = torch.randn(n, d, device=dev, generator=g); db /= db.norm(dim=-1, keepdim=True)
    sh = dense.DenseShadow(db)
    for nq in (209, 450, 1000):
        q = torch.randn(nq, d, device=dev, generator=g); q /= q.norm(dim=-1, keepdim=True)
        for k in (10, 500):
            i1, v1, c1 = dense.topk_dense(db, q, k, shadow=sh)
            path = dense.last_path
            i2, v2, c2 = dense.topk_dense_exact(db, q, k)
            ok = torch.equal(i1, i2) and torch.equal(v1, v2) and torch.equal(c1, c2)
            bad += (not ok)
            print(d, nq, k, path, "ok" if ok else "MISMATCH")
print("mismatches:", bad)
