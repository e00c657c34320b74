"""Dev tool: A/B of the shadow scan's tuning switches in ONE process, launches alternating, HIP events around the candidate pass
(scan + regroup) and around the whole warm top-k.  AB=epi (default): swat_dense_set_epilogue 2 / 3; AB=spread: swat_dense_set_spread 0 / 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import _lib
from swat_amd.retrieval import dense
dev = torch.device("cuda:0"); L = _lib.lib()
n, d, nq, k = 1_250_000, 512, int(os.environ.get("NQ", 200)), 500
g = torch.Generator(device=dev).manual_seed(7)
db = torch.randn(n, d, device=dev, generator=g); db /= db.norm(dim=-1, keepdim=True)
q = torch.randn(nq, d, device=dev, generator=g); q /= q.norm(dim=-1, keepdim=True)
SH = dense.DenseShadow(db)
AB = os.environ.get("AB", "epi")
FORMS = (2, 3) if AB == "epi" else (0, 1)
SET = L.swat_dense_set_epilogue if AB == "epi" else L.swat_dense_set_spread
ref = None
for form in FORMS:
    SET(form)
    out = dense.topk_dense(db, q, k, shadow=SH)
    if ref is None: ref = out
    else: print("form", form, "same lists:", all(torch.equal(a, b) for a, b in zip(ref, out)), dense.last_path)
torch.cuda.synchronize()
c = dense._cache
def ev(): return torch.cuda.Event(enable_timing=True)
res = {f: [] for f in FORMS}; tot = {f: [] for f in FORMS}
for rep in range(12):
    for form in FORMS:
        SET(form)
        c["cnt"].zero_()
        a, b = ev(), ev(); a.record()
        for _ in range(5): dense.candidates(c, db, shadow=SH)
        b.record(); torch.cuda.synchronize()
        res[form].append(a.elapsed_time(b) / 5)
        a, b = ev(), ev(); a.record()
        for _ in range(5): dense.topk_dense(db, q, k, shadow=SH)
        b.record(); torch.cuda.synchronize()
        tot[form].append(a.elapsed_time(b) / 5)
for form in FORMS:
    r, t = sorted(res[form]), sorted(tot[form])
    print(f"{AB} {form}: scan+regroup median {r[len(r)//2]:.3f} ms (min {r[0]:.3f}); warm top-k median {t[len(t)//2]:.3f} ms (min {t[0]:.3f})")
