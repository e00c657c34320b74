"""Dense (N_query x N_db) cosine top-k on the fp32 matrix cores — BASELINE config 4 — and near-duplicate marks.

ref: retrieval/sample_retrieval.py:724-825 (per-class GEMV + sort, generalised to a query batch), :237-275 (dedup).
Exact: similarities are bit-identical to the lane-per-row scan and the oracle (k-ordered fp32 fma chain); the
candidate threshold only prunes, the caller-visible top-k is checked to be complete (count >= k) or recomputed.
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


def _thresholds(queries, db, k, sample, boost):
    """Per-query pruning threshold from a strided row sample: the value at the rank that should leave ~boost*k rows.
    (16 384 sample rows put the rank near 30 for k = 500 of 1.25 M: the kept count scatters by ~20 % around boost*k,
    well inside [k, CAP]; the caller re-brackets when it does not.)"""
    n = db.shape[0]
    nq = queries.shape[0]
    if n <= CAP // 2:
        return torch.full((nq,), float("-inf"), device=db.device)
    ns = min(n, sample)
    stride = n // ns
    smp = db[::stride][:ns].contiguous()
    S = sim_matrix(queries, smp)
    rank = min(ns, 2048, max(int(boost * k * ns / n) + 8, 16))
    off = torch.arange(0, (nq + 1) * ns, ns, device=db.device, dtype=torch.int64)
    _, val, _ = ops.segment_topk(S.view(-1), off, rank, thr=float("-inf"))
    return val[:, rank - 1].contiguous()


def topk_dense(db, queries, k, row_base=0, sample=16384):
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
        thr = _thresholds(queries, db, k, sample, boost)
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
