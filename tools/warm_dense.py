"""Dev tool: per-call times of the first dense top-k calls on a fresh shard (how long until the rate is steady)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd.retrieval import dense
dev = torch.device("cuda:0")
n, d, nq, k = 1_250_000, 512, 200, 500
g = torch.Generator(device=dev).manual_seed(7)
db = torch.randn(n, d, device=dev, generator=g); db /= db.norm(dim=-1, keepdim=True)
sh = dense.DenseShadow(db)
qs = []
for i in range(60):
    q = torch.randn(nq, d, device=dev, generator=g); q /= q.norm(dim=-1, keepdim=True); qs.append(q)
torch.cuda.synchronize()
ts = []
for q in qs:
    t0 = time.perf_counter()
    dense.topk_dense(db, q, k, shadow=sh)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("cold calls, each synchronised (ms):", " ".join(f"{t:.2f}" for t in ts))
c = dense._cache
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    c["cnt"].zero_(); dense.candidates(c, db, shadow=sh); ev[i + 1].record()
torch.cuda.synchronize()
print("scan + regroup (ms):", " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.3f}" for i in range(40)))
time.sleep(2.0)
ev[0].record()
for i in range(40):
    c["cnt"].zero_(); dense.candidates(c, db, shadow=sh); ev[i + 1].record()
torch.cuda.synchronize()
print("after 2 s idle     :", " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.3f}" for i in range(40)))
