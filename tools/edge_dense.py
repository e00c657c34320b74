"""Dev tool: edge shapes of the dense top-k through the shadow scan (tiny / ragged shards, one query, k near n) against the exact scan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd.retrieval import dense
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
bad = 0
for d in (256, 512, 768, 1024):
    for n in (1, 17, 255, 256, 257, 300, 4097, 9001):
        for nq in (1, 33, 200):
            for k in (1, 10, 500):
                print(d, n, nq, k, flush=True)
                db = torch.randn(n, d, device=dev, generator=g); db /= db.norm(dim=-1, keepdim=True)
                q = torch.randn(nq, d, device=dev, generator=g); q /= q.norm(dim=-1, keepdim=True)
                sh = dense.DenseShadow(db)
                i1, v1, c1 = dense.topk_dense(db, q, k, row_base=7, shadow=sh)
                path = dense.last_path
                i2, v2, c2 = dense.topk_dense_exact(db, q, k, row_base=7)
                ok = torch.equal(i1, i2) and torch.equal(v1, v2) and torch.equal(c1, c2)
                if not ok:
                    bad += 1
                    print("MISMATCH", d, n, nq, k, path)
print("mismatches:", bad)
