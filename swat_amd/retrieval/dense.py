"""Dense (N_query x N_db) cosine top-k — BASELINE config 4 — and near-duplicate marks.

ref: retrieval/sample_retrieval.py:724-825 (per-class GEMV + sort, generalised to a query batch), :237-275 (dedup).

topk_dense: ONE pass over the database at the HBM rate selects candidates with bf16 MFMA scores (csrc/retrieval_bf16.hip),
the best K2 ~ 2k candidates per query are re-scored with the canonical k-ordered fp32 fma chain (bit-identical to the
lane-per-row scan and the oracle) and the exact (value desc, row asc) top-k is taken from them.  The result is proved
complete on the device from the bf16 error bound |approx - exact| <= (2^-8 + 2^-12) |q| |d| — every row that was not
re-scored has exact < the k-th kept value — else the call falls back to the exact f32-matrix-core scan
(topk_dense_exact).  One host synchronisation per call (the proof flag); thresholds and bf16 queries are cached per
(database, queries) pair.
"""
import torch

from .. import _lib
from . import ops

CAP = 8192


def sim_matrix(queries, db):
    """Full S = Q DB^T (fp32, canonical order) -> [n_q, n_db]."""
    nq, d = queries.shape
    n = db.shape[0]
    S = torch.empty((nq, n), dtype=torch.float32, device=db.device)
    rc = _lib.lib().swat_sim_topk_candidates(_lib.ptr(queries), nq, _lib.ptr(db), n, d, None, 0, None, 0, None, _lib.ptr(S), n,
                                             _lib.stream())
    _lib.check(rc, "swat_sim_topk_candidates")
    return S


_sample = {}


def _thresholds(queries, db, k, sample, boost, approx=False, want_qnorm=False):
    """Per-query pruning threshold from a strided row sample: the value at the rank that should leave ~boost*k rows.
    (16 384 sample rows put the rank near 30 for k = 500 of 1.25 M: the kept count scatters by ~20 % around boost*k,
    well inside [k, CAP]; the caller re-brackets when it does not.)"""
    n = db.shape[0]
    nq = queries.shape[0]
    if n <= CAP // 2:
        t = torch.full((nq,), float("-inf"), device=db.device)
        return (t, queries.norm(dim=1).max().reshape(1)) if want_qnorm else t
    ns = min(n, sample)
    stride = n // ns
    # the strided row sample depends on the shard only: kept per database object / version (33 MB gathered per call otherwise)
    import weakref
    skey = (db.data_ptr(), tuple(db.shape), db._version, ns)
    if _sample.get("key") != skey or _sample["ref"]() is not db:
        _sample.clear()
        _sample.update(key=skey, ref=weakref.ref(db), rows=db[::stride][:ns].contiguous())
    smp = _sample["rows"]
    d = db.shape[1]
    bkey = ("buf", nq, ns, d)
    if bkey not in _sample:   # per-shape buffers (a cold call is ~0.6 ms: allocations and extra launches count)
        _sample[bkey] = (torch.empty((nq, ns), dtype=torch.float32, device=db.device), torch.empty((nq, d), dtype=torch.bfloat16, device=db.device),
                         torch.empty(nq, dtype=torch.float32, device=db.device), torch.zeros(nq + 1, dtype=torch.float32, device=db.device),
                         torch.empty(1, dtype=torch.float32, device=db.device))
    S, qbf, thr, nws, qn2 = _sample[bkey]
    if approx and d % 64 == 0 and ns % 4 == 0:
        # the candidate pass compares bf16 MFMA scores with these thresholds, so bf16 sample scores serve as well as exact ones and
        # cost a sixth: one bf16 GEMM [n_q, d] x [n_s, d]^T (the step's own product kernels) instead of the f32-matrix-core scan
        from .. import ops as aops
        if "rows_bf16" not in _sample:
            _sample["rows_bf16"] = aops.cast_bf16(smp)
        aops.gemm(aops.NT, aops.cast_bf16(queries, out=qbf), _sample["rows_bf16"], nq, ns, d, d, d, S, ns)
    else:
        S = sim_matrix(queries, smp)
    rank = min(ns, 2048, max(int(boost * k * ns / n) + 8, 16))
    # the rank-th largest sample score of every query: one radix-select launch (csrc/retrieval.hip rank_select_kernel; the two
    # sorting launches it replaces took 62 us)
    # (rider: max |q| for the completeness proof, from the same launch)
    _lib.check(_lib.lib().swat_rank_select(_lib.ptr(S), nq, ns, rank, _lib.ptr(thr), _lib.ptr(queries) if want_qnorm else None, d,
                                           _lib.ptr(nws) if want_qnorm else None, _lib.ptr(qn2) if want_qnorm else None, _lib.stream()),
               "swat_rank_select")
    return (thr, qn2) if want_qnorm else thr


def topk_dense_exact(db, queries, k, row_base=0, sample=16384):
    """-> (idx [n_q, k] int64 global rows (-1 padded), val [n_q, k] fp32, cnt [n_q]) in (value desc, row asc) order."""
    assert db.dtype == torch.float32 and queries.dtype == torch.float32 and k <= CAP
    n, d = db.shape
    nq = queries.shape[0]
    dev = db.device
    keys = torch.empty((nq, CAP), dtype=torch.int64, device=dev)
    cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    need = min(k, n)
    boost = 3.0
    for attempt in range(6):
        # a PRIVATE copy: _thresholds returns the per-shape buffer that topk_dense's cache entry for (db, queries, k) keeps — this
        # path runs as that call's fallback and must not overwrite the tuned thresholds the cache key stands for
        thr = _thresholds(queries, db, k, sample, boost).clone()
        cnt.zero_()
        rc = _lib.lib().swat_sim_topk_candidates(_lib.ptr(queries), nq, _lib.ptr(db), n, d, _lib.ptr(thr), int(row_base),
                                                 _lib.ptr(keys), CAP, _lib.ptr(cnt), None, 0, _lib.stream())
        _lib.check(rc, "swat_sim_topk_candidates")
        lo, hi = int(cnt.min().item()), int(cnt.max().item())
        if lo >= need and hi <= CAP:
            break
        boost = boost * 4.0 if lo < need else boost / 2.0  # too few candidates -> lower the threshold, too many -> raise it
    else:
        raise RuntimeError("topk_dense: could not bracket the candidate count; similarity distribution too degenerate")
    idx = torch.empty((nq, k), dtype=torch.int64, device=dev)
    val = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_cnt = torch.empty(nq, dtype=torch.int64, device=dev)
    rc = _lib.lib().swat_candidates_topk(_lib.ptr(keys), CAP, _lib.ptr(cnt), nq, k, _lib.ptr(idx), _lib.ptr(val), _lib.ptr(out_cnt),
                                         _lib.stream())
    _lib.check(rc, "swat_candidates_topk")
    return idx, val, out_cnt


EPS_BF16 = 2.0 ** -8 + 2.0 ** -12  # two bf16 roundings per product ((1 + 2^-9)^2 - 1) + fp32 accumulation over <= 1024 terms
EPS_MAX_D = 1024                    # largest d the bound above covers
_cache = {}


_shape_cache = {}


def _shape_buffers(nq, d, k, dev):
    """Everything of a call that depends only on (n_q, d, k, device): candidate lists, re-score index / value tensors, the staging
    workspace (zeroed once: the scan's regroup pass leaves its counters zero), the packed-query buffer.  A NEW query tensor — the
    case a retrieval job has — re-uses all of it: the cold call then launches the threshold sample product, two rank selections,
    the query packing and one |q| reduction, and nothing else (round 3 rebuilt these 14 tensors per query tensor: ~0.05 ms of
    allocator work and fill launches per cold call)."""
    key = (nq, d, k, str(dev))
    b = _shape_cache.get(key)
    if b is None:
        K2 = min(TOPK2_MAX, max(64, (max(2 * k, k + 256) + 63) // 64 * 64))
        nbytes, wbytes = int(_lib.lib().swat_dense_packed_bytes(nq, d)), int(_lib.lib().swat_dense_workspace_bytes(nq))
        packed = bool(nbytes and wbytes)  # d % 128 == 0, n_q <= 4096: the packed image the deep kernel reads linearly (csrc/retrieval_bf16.hip)
        b = dict(K2=K2, packed=packed,
                 qb=torch.empty(nbytes, dtype=torch.uint8, device=dev) if packed else None,
                 ws=torch.zeros(wbytes, dtype=torch.uint8, device=dev) if packed else None,   # the slot counters start (and are left) zero
                 row_seg=torch.arange(nq, device=dev, dtype=torch.int32).repeat_interleave(K2).contiguous(),
                 idx2=torch.empty((nq, K2), dtype=torch.int64, device=dev), val2=torch.empty((nq, K2), dtype=torch.float32, device=dev),
                 cnt2=torch.empty(nq, dtype=torch.int64, device=dev), keys2=torch.empty((nq, K2), dtype=torch.int64, device=dev),
                 fails=torch.zeros(1, dtype=torch.int32, device=dev),   # zeroed by the select launch, added to by the last one
                 keys=torch.empty((nq, CAP), dtype=torch.int64, device=dev), cnt=torch.zeros(nq, dtype=torch.int32, device=dev),
                 n2=torch.zeros(1, dtype=torch.float32, device=dev), cnt_clean=True)
        if len(_shape_cache) >= 4:
            _shape_cache.clear()
        _shape_cache[key] = b
    return b


def _prepared(db, queries, k, sample):
    """Per (database, queries, k): pruning thresholds (from a row sample), the packed bf16 queries, |q| max — on top of the
    shape's buffers (_shape_buffers).  A hit needs the SAME tensor objects (weak references: a new tensor that happens to reuse a
    freed tensor's address is a miss) at the same version, so in-place updates invalidate it too."""
    import weakref
    key = (db.shape, db._version, queries.shape, queries._version, k)
    hit = (_cache.get("key") == key and _cache["db_ref"]() is db and _cache["q_ref"]() is queries
           and _cache["ptrs"] == (db.data_ptr(), queries.data_ptr()))
    if not hit:
        nq, d = queries.shape
        b = _shape_buffers(nq, d, k, db.device)
        K2 = b["K2"]
        # (the packing first: it needs nothing from the thresholds, and the GPU — idle at the start of a cold call — runs it while the
        # host is still marshalling the threshold sample's product)
        if b["packed"]:
            qb = b["qb"]
            _lib.check(_lib.lib().swat_dense_pack_queries(_lib.ptr(queries), nq, d, _lib.ptr(qb), _lib.stream()), "swat_dense_pack_queries")
        else:
            qb = queries.to(torch.bfloat16).contiguous()
        thr, qn2 = _thresholds(queries, db, k, sample, 2.5 * max(1.0, K2 / (2.0 * k)), approx=True, want_qnorm=True)
        clean = b["cnt_clean"]
        _cache.clear()
        _cache.update(b)
        _cache.update(key=key, db_ref=weakref.ref(db), q_ref=weakref.ref(queries), ptrs=(db.data_ptr(), queries.data_ptr()),
                      thr=thr, qb=qb, qn2=qn2,   # max |q| stays on the device (a rider of the threshold launch): no host sync
                      cnt_clean=clean, shape=b)
    return _cache


class DenseShadow:
    """bf16 image of a database shard for the candidate scan, built ONCE per index (csrc/retrieval_bf16.hip
    dense_shadow_kernel): rows rounded to bf16 in the scan's fragment order + max |row|^2 of the fp32 rows.  topk_dense(...,
    shadow=...) then streams 2 d bytes per row instead of 4 d and returns the same lists bit for bit (same roundings); the exact
    re-scoring and the completeness proof keep reading the fp32 rows.  Costs half the shard's bytes again in HBM.  Tied to the
    database tensor OBJECT and its version: an in-place update of the rows makes `matches` false and the fp32 scan answers."""

    def __init__(self, db):
        import weakref
        assert db.dtype == torch.float32 and db.is_cuda and db.is_contiguous()
        n, d = db.shape
        nbytes = int(_lib.lib().swat_dense_shadow_bytes(n, d))
        if nbytes == 0:
            raise _lib.SwatHipError(f"DenseShadow: width {d} is not a multiple of 128 (the deep candidate kernel's shapes)")
        self.rows = torch.empty(nbytes, dtype=torch.uint8, device=db.device)
        self.n2 = torch.zeros(1, dtype=torch.float32, device=db.device)
        _lib.check(_lib.lib().swat_dense_build_shadow(_lib.ptr(db), n, d, _lib.ptr(self.rows), _lib.ptr(self.n2), _lib.stream()),
                   "swat_dense_build_shadow")
        self._ref, self._key = weakref.ref(db), (db.data_ptr(), tuple(db.shape), db._version)

    def matches(self, db):
        return self._ref() is db and self._key == (db.data_ptr(), tuple(db.shape), db._version)


def candidates(c, db, row_base=0, shadow=None, set_n2=True):
    """The candidate pass on prepared buffers c (= _prepared(...)): appends to c["keys"] / c["cnt"], max |row|^2 to c["n2"]
    (with a shadow: a copy of its max |row|^2, unless set_n2 is False — topk_dense hands shadow.n2 to the proof directly)."""
    L = _lib.lib()
    nq = c["cnt"].numel()
    c["cnt_clean"] = False   # the counters now hold this pass's counts (topk_dense's last launch clears them again)
    if "shape" in c:
        c["shape"]["cnt_clean"] = False
    if shadow is not None and c["ws"] is not None:
        _lib.check(L.swat_sim_candidates_shadow(_lib.ptr(c["qb"]), nq, _lib.ptr(shadow.rows), db.shape[0], db.shape[1], _lib.ptr(c["thr"]),
                                                int(row_base), _lib.ptr(c["keys"]), CAP, _lib.ptr(c["cnt"]), _lib.ptr(c["ws"]),
                                                c["ws"].numel(), _lib.stream()), "swat_sim_candidates_shadow")
        if set_n2:
            c["n2"].copy_(shadow.n2)
        return
    args = (_lib.ptr(c["qb"]), nq, _lib.ptr(db), db.shape[0], db.shape[1], _lib.ptr(c["thr"]), int(row_base), _lib.ptr(c["keys"]), CAP,
            _lib.ptr(c["cnt"]), _lib.ptr(c["n2"]))
    if c["ws"] is not None:
        _lib.check(L.swat_sim_candidates_packed(*args, _lib.ptr(c["ws"]), c["ws"].numel(), _lib.stream()), "swat_sim_candidates_packed")
    else:
        _lib.check(L.swat_sim_candidates_bf16(*args, _lib.stream()), "swat_sim_candidates_bf16")


TOPK2_MAX = 2048
BIG = 1 << 62
last_path = None  # which route the last topk_dense call took (tests, bench.py)


def topk_dense(db, queries, k, row_base=0, sample=16384, shadow=None):
    """-> (idx [n_q, k] int64 global rows (-1 padded), val [n_q, k] fp32, cnt [n_q]) in (value desc, row asc) order;
    values and indices are those of the exact scan (oracle order), bit for bit.  shadow: a DenseShadow of db (optional)."""
    assert db.dtype == torch.float32 and queries.dtype == torch.float32 and k <= CAP
    n, d = db.shape
    nq = queries.shape[0]
    need = min(k, n)
    global last_path
    if 2 * k > TOPK2_MAX or d % 64 != 0 or d > EPS_MAX_D:
        # d > 1024: the fp32-accumulation share of EPS_BF16 (2^-12) no longer bounds the candidate scores' error, so the
        # device-side completeness proof would not be sound — the exact f32-matrix-core scan answers instead
        last_path = "exact"
        return topk_dense_exact(db, queries, k, row_base, sample)
    c = _prepared(db, queries, k, sample)
    if c.get("proof_failed"):
        # the completeness proof already failed for exactly these tensors: a warm repeat goes straight to the exact scan instead of
        # paying the candidate pass, a host synchronisation and the exact scan every time
        last_path = "exact-fallback"
        return topk_dense_exact(db, queries, k, row_base, sample)
    K2, thr, keys, cnt, n2 = c["K2"], c["thr"], c["keys"], c["cnt"], c["n2"]
    L = _lib.lib()
    if shadow is not None and not shadow.matches(db):
        shadow = None  # built from other rows (or the rows changed in place): the fp32 scan answers
    use_shadow = shadow is not None and c["ws"] is not None
    # small launches count at 0.5 ms per pass: the candidate counters are cleared by the last launch of the previous call (cleared
    # here only after a call that did not get that far), the staging slots by the regroup launch, and with a shadow max |row|^2 is
    # the build's own scalar — no fill / copy / memset launches on the way
    if not c.get("cnt_clean", False):
        cnt.zero_()
    c["cnt_clean"] = False
    if "shape" in c:
        c["shape"]["cnt_clean"] = False
    if use_shadow:
        n2 = shadow.n2
    else:
        n2.zero_()
    candidates(c, db, row_base, shadow, set_n2=False)
    idx2, val2, cnt2, keys2 = c["idx2"], c["val2"], c["cnt2"], c["keys2"]
    # the K2 best approximate rows of every query, unordered (radix select: the exact re-score below orders them again), the K2-th
    # best score in slot K2 - 1 for the proof
    _lib.check(L.swat_candidates_select(_lib.ptr(keys), CAP, _lib.ptr(cnt), nq, K2, _lib.ptr(idx2), _lib.ptr(val2), _lib.ptr(cnt2),
                                        _lib.ptr(c["fails"]), _lib.stream()), "swat_candidates_select")
    # exact value of the K2 best approximate rows of every query, as (value desc, global row asc) sort keys
    ops.sim_gathered(db, idx2.view(-1), queries, c["row_seg"], keys_out=keys2, row_base=row_base, want_sims=False)
    dev = db.device
    idx = torch.empty((nq, k), dtype=torch.int64, device=dev)
    val = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_cnt = torch.empty(nq, dtype=torch.int64, device=dev)
    # the exact top-k of the re-scored rows, and in the same launch the completeness proof: a row that was not re-scored has
    # approx <= bound_q (the K2-th kept approximate score, or the threshold when fewer than K2 candidates exist), hence
    # exact <= bound_q + delta < the k-th exact value
    _lib.check(L.swat_candidates_topk_proof(_lib.ptr(keys2), K2, _lib.ptr(cnt), nq, k, _lib.ptr(idx), _lib.ptr(val), _lib.ptr(out_cnt),
                                            _lib.ptr(cnt), CAP, K2, _lib.ptr(val2), _lib.ptr(thr), need, float(EPS_BF16), _lib.ptr(n2),
                                            _lib.ptr(c["qn2"]), _lib.ptr(c["fails"]), 1, _lib.stream()), "swat_candidates_topk_proof")
    c["cnt_clean"] = True   # (the launch above clears every cnt[q] once it has read it)
    if "shape" in c:
        c["shape"]["cnt_clean"] = True
    if c["fails"].item() != 0:  # the one host synchronisation of the call
        c["proof_failed"] = True
        last_path = "exact-fallback"
        return topk_dense_exact(db, queries, k, row_base, sample)
    last_path = "bf16-shadow-candidates+exact-rescore" if use_shadow else "bf16-candidates+exact-rescore"
    return idx, val, out_cnt


def near_dup_marks(feats, thr=0.9):
    """Positional marks of remove_near_duplicates2 (ref :237-275): marks[j] = 1 iff some i < j has <f_i, f_j> > thr."""
    n, d = feats.shape
    marks = torch.empty(n, dtype=torch.uint8, device=feats.device)
    rc = _lib.lib().swat_near_dup_marks(_lib.ptr(feats), n, d, float(thr), _lib.ptr(marks), _lib.stream())
    _lib.check(rc, "swat_near_dup_marks")
    return marks


def remove_near_duplicates2(index, thr=0.9):
    """Whole-index form with the reference's quirk (:262-267): positional marks are matched against int(basename).
    -> (duplicate_files: {str(cls): set(paths)}, dup_fraction per class, average)."""
    from collections import defaultdict
    import numpy as np
    dups, fracs = defaultdict(set), []
    off = index.offsets.tolist()
    paths, ids = index.paths_array(), index.file_ids()
    for s, c in enumerate(index.class_ids.tolist()):
        b, e = off[s], off[s + 1]
        if e == b:
            continue
        marks = near_dup_marks(index.image_features[b:e].contiguous(), thr).cpu().numpy()
        to_remove = marks.nonzero()[0]
        hit = np.isin(ids[b:e], to_remove)
        if hit.any():
            dups[str(c)].update(paths[b:e][hit].tolist())
        fracs.append(len(to_remove) / (e - b))
    return dups, fracs, sum(fracs) / max(1, len(fracs))
