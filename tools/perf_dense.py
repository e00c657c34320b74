"""Dev tool: the dense (N_query x N_db) top-k of BASELINE config 4 on one shard, stage by stage (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import _lib
from swat_amd.retrieval import dense, ops
dev = torch.device("cuda:0"); L = _lib.lib()
n, d, nq, k = int(os.environ.get("N", 1_250_000)), 512, int(os.environ.get("NQ", 200)), 500
g = torch.Generator(device=dev).manual_seed(7)
db = torch.randn(n, d, device=dev, generator=g); db /= db.norm(dim=-1, keepdim=True)
q = torch.randn(nq, d, device=dev, generator=g); q /= q.norm(dim=-1, keepdim=True)
def ev(): return torch.cuda.Event(enable_timing=True)
SH = dense.DenseShadow(db) if os.environ.get("SWAT_DENSE_SHADOW", "1") != "0" else None
dense.topk_dense(db, q, k, shadow=SH); torch.cuda.synchronize()
print("route:", dense.last_path, "| SWAT_DENSE_WAVES =", os.environ.get("SWAT_DENSE_WAVES", "12"))
c = dense._cache
for rep in range(3):
    a, b = ev(), ev()
    a.record()
    for _ in range(5): dense.topk_dense(db, q, k, shadow=SH)
    b.record(); torch.cuda.synchronize()
    tot = a.elapsed_time(b) / 5
    c["cnt"].zero_(); c["n2"].zero_()
    a, b = ev(), ev(); a.record()
    for _ in range(5):
        dense.candidates(c, db, shadow=SH)
    b.record(); torch.cuda.synchronize()
    tc = a.elapsed_time(b) / 5
    print(f"pass {tot:.3f} ms = {nq / tot * 1e3:,.0f} queries/s; candidate kernel {tc:.3f} ms = {n * d * 4 / tc / 1e6:.0f} GB/s "
          f"({n * d * 4 / tc / 1e6 / 8000:.3f} of 8 TB/s), whole pass {n * d * 4 / tot / 1e6 / 8000:.3f} of 8 TB/s; "
          f"candidates per query {c['cnt'].float().mean().item() / 5:.0f} (min {c['cnt'].min().item() // 5}), max |d| {c['n2'].sqrt().item():.4f}")
a, b = ev(), ev(); a.record()
dense.topk_dense_exact(db, q, k)
b.record(); torch.cuda.synchronize()
print(f"exact f32-matrix-core scan: {a.elapsed_time(b):.3f} ms")
# ---- stage breakdown of one pass (events between the stages of dense.topk_dense) ----
K2, thr, keys, cnt, n2 = c["K2"], c["thr"], c["keys"], c["cnt"], c["n2"]
idx2, val2, cnt2, keys2 = c["idx2"], c["val2"], c["cnt2"], c["keys2"]
marks = [ev() for _ in range(7)]
torch.cuda.synchronize()
marks[0].record()
cnt.zero_(); n2.zero_()
dense.candidates(c, db, shadow=SH)
marks[1].record()
_lib.check(L.swat_candidates_select(_lib.ptr(keys), dense.CAP, _lib.ptr(cnt), nq, K2, _lib.ptr(idx2), _lib.ptr(val2), _lib.ptr(cnt2), _lib.ptr(c["fails"]), _lib.stream()), "csel")
marks[2].record()
ops.sim_gathered(db, idx2.view(-1), q, c["row_seg"], keys_out=keys2, want_sims=False)
marks[3].record()
idx = torch.empty((nq, k), dtype=torch.int64, device=dev); val = torch.empty((nq, k), dtype=torch.float32, device=dev); out_cnt = torch.empty(nq, dtype=torch.int64, device=dev)
_lib.check(L.swat_candidates_topk_proof(_lib.ptr(keys2), K2, _lib.ptr(cnt), nq, k, _lib.ptr(idx), _lib.ptr(val), _lib.ptr(out_cnt), _lib.ptr(cnt), dense.CAP, K2, _lib.ptr(val2), _lib.ptr(thr), k, float(dense.EPS_BF16), _lib.ptr(n2), _lib.ptr(c["qn2"]), _lib.ptr(c["fails"]), 0, _lib.stream()), "ctk2")
marks[4].record()
marks[5].record()
flag = c["fails"].item()
marks[6].record(); torch.cuda.synchronize()
names = ["candidates (bf16 MFMA scan)", "candidates_select (approx, top K2, unordered)", "sim_gathered (exact re-score -> keys)", "candidates_topk (exact, top k) + proof", "-", "item()"]
print(f"K2 = {K2}; " + "; ".join(f"{nme} {marks[i].elapsed_time(marks[i + 1]):.3f}" for i, nme in enumerate(names)) + f"  (ms); failing queries: {flag}")
