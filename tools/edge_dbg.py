import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import _lib
from swat_amd.retrieval import dense, ops
dev = torch.device("cuda:0"); L = _lib.lib()
g = torch.Generator(device=dev).manual_seed(5)
def mk(d, n, nq):
    db = torch.randn(n, d, device=dev, generator=g); db /= db.norm(dim=-1, keepdim=True)
    q = torch.randn(nq, d, device=dev, generator=g); q /= q.norm(dim=-1, keepdim=True)
    return db, q
db, q = mk(256, 17, 1)
dense.topk_dense(db, q, 500, row_base=7); torch.cuda.synchronize(); print("first ok", flush=True)
db, q = mk(256, 17, 33); k = 1; nq = 33
def S(msg):
    torch.cuda.synchronize(); print(msg, flush=True)
c = dense._prepared(db, q, k, 16384); S("prepared K2=%d" % c["K2"])
K2, thr, keys, cnt, n2 = c["K2"], c["thr"], c["keys"], c["cnt"], c["n2"]
cnt.zero_(); n2.zero_()
dense.candidates(c, db, 7, None); S("candidates; cnt[:4]=%s" % cnt[:4].tolist())
idx2, val2, cnt2, keys2 = c["idx2"], c["val2"], c["cnt2"], c["keys2"]
_lib.check(L.swat_candidates_select(_lib.ptr(keys), dense.CAP, _lib.ptr(cnt), nq, K2, _lib.ptr(idx2), _lib.ptr(val2), _lib.ptr(cnt2), _lib.ptr(c["fails"]), _lib.stream()), "sel"); S("select; idx2 range %d..%d" % (idx2.min().item(), idx2.max().item()))
ops.sim_gathered(db, idx2.view(-1), q, c["row_seg"], keys_out=keys2, row_base=7, want_sims=False); S("gathered")
idx = torch.empty((nq, k), dtype=torch.int64, device=dev); val = torch.empty((nq, k), dtype=torch.float32, device=dev); out_cnt = torch.empty(nq, dtype=torch.int64, device=dev)
_lib.check(L.swat_candidates_topk_proof(_lib.ptr(keys2), K2, _lib.ptr(cnt), nq, k, _lib.ptr(idx), _lib.ptr(val), _lib.ptr(out_cnt), _lib.ptr(cnt), dense.CAP, K2, _lib.ptr(val2), _lib.ptr(thr), 1, float(dense.EPS_BF16), _lib.ptr(n2), _lib.ptr(c["qn2"]), _lib.ptr(c["fails"]), _lib.stream()), "tp"); S("topk_proof fails=%d" % c["fails"].item())
r = dense.topk_dense_exact(db, q, k, 7); S("exact")
