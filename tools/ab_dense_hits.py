"""Dev tool: what the candidate hits cost the shadow scan — the same launch with the real thresholds, with thresholds scaled so that
about a quarter / a sixteenth as many scores pass, and with thresholds nothing reaches (scan + regroup, HIP events, one process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import _lib
from swat_amd.retrieval import dense
dev = torch.device("cuda:0")
n, d, nq, k = 1_250_000, 512, int(os.environ.get("NQ", 200)), 500
g = torch.Generator(device=dev).manual_seed(7)
db = torch.randn(n, d, device=dev, generator=g); db /= db.norm(dim=-1, keepdim=True)
q = torch.randn(nq, d, device=dev, generator=g); q /= q.norm(dim=-1, keepdim=True)
SH = dense.DenseShadow(db)
dense.topk_dense(db, q, k, shadow=SH); torch.cuda.synchronize()
c = dense._cache
thr0 = c["thr"].clone()
def ev(): return torch.cuda.Event(enable_timing=True)
for name, thr in (("real thresholds", thr0), ("+0.01", thr0 + 0.01), ("+0.02", thr0 + 0.02), ("+0.04", thr0 + 0.04), ("nothing passes", thr0 * 0 + 10)):
    c["thr"].copy_(thr)
    ts = []
    for rep in range(8):
        c["cnt"].zero_()
        a, b = ev(), ev(); a.record()
        for _ in range(5): dense.candidates(c, db, shadow=SH)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 5)
    ts.sort()
    print(f"{name:>16}: scan + regroup median {ts[len(ts) // 2]:.3f} ms (min {ts[0]:.3f}); candidates per query {c['cnt'].float().mean().item() / 5:.0f}")
c["thr"].copy_(thr0)
