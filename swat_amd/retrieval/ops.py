"""Path B device ops (thin wrappers over the C ABI; see include/swat_hip.h)."""
import torch

from .. import _lib


def similarity_segmented(feats, queries, row_seg=None, out=None):
    """sims[i] = <feats[i], queries[row_seg[i]]> — ONE fp32 fma chain over k ascending.

    ref: retrieval/sample_retrieval.py:397-416 (t2t_similarity), :335-353 (cal_t2i_similarity).
    feats [N,d] fp32 cuda, queries [S,d] fp32 cuda, row_seg [N] int32 cuda or None."""
    assert feats.dtype == torch.float32 and queries.dtype == torch.float32
    n, d = feats.shape
    if row_seg is not None:
        assert row_seg.dtype == torch.int32 and row_seg.numel() == n
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=feats.device)
    rc = _lib.lib().swat_sim_segmented(_lib.ptr(feats), n, d, _lib.ptr(queries), _lib.ptr(row_seg), _lib.ptr(out),
                                       _lib.stream())
    _lib.check(rc, "swat_sim_segmented")
    return out


def sim_gathered(feats, row_index, queries, row_seg, out=None, keys_out=None, row_base=0, want_sims=True):
    """s_r = <feats[row_index[r] - row_base], queries[row_seg[r]]> — the same k-ordered fp32 fma chain as
    similarity_segmented on a row list (row_index[r] < 0 -> -inf): the exact re-scoring of the dense scan's candidates.
    keys_out (int64 [n]): (value desc, row asc) sort keys for swat_candidates_topk."""
    assert feats.dtype == torch.float32 and queries.dtype == torch.float32 and row_index.dtype == torch.int64
    assert row_seg.dtype == torch.int32 and row_seg.numel() == row_index.numel()
    n = row_index.numel()
    if out is None and want_sims:
        out = torch.empty(n, dtype=torch.float32, device=feats.device)
    rc = _lib.lib().swat_sim_gathered(_lib.ptr(feats), _lib.ptr(row_index), n, feats.shape[1], _lib.ptr(queries), _lib.ptr(row_seg),
                                      _lib.ptr(out), _lib.ptr(keys_out), int(row_base), _lib.stream())
    _lib.check(rc, "swat_sim_gathered")
    return out


def segment_topk(sims, seg_offsets, k, thr=0.0, t2i=None, t2i_thr=0.25, excluded=None):
    """Filtered segmented top-k in (sim desc, row asc) order.

    ref: sorted(..., reverse=True) :754/:807 + add_to_split :439-482 / add_t2t_ranked_t2i_tshd_to_split :492-540.
    Returns (idx [S,k] int64 (-1 padded), val [S,k] fp32, cnt [S] int64)."""
    assert sims.dtype == torch.float32 and seg_offsets.dtype == torch.int64
    n_seg = seg_offsets.numel() - 1
    dev = sims.device
    idx = torch.empty((n_seg, k), dtype=torch.int64, device=dev)
    val = torch.empty((n_seg, k), dtype=torch.float32, device=dev)
    cnt = torch.empty(n_seg, dtype=torch.int64, device=dev)
    if t2i is not None:
        assert t2i.dtype == torch.float32 and t2i.numel() == sims.numel()
    if excluded is not None:
        assert excluded.dtype == torch.uint8 and excluded.numel() == sims.numel()
    rc = _lib.lib().swat_segment_topk(_lib.ptr(sims), _lib.ptr(t2i), _lib.ptr(excluded), _lib.ptr(seg_offsets), n_seg,
                                      float(thr), float(t2i_thr), int(k), _lib.ptr(idx), _lib.ptr(val), _lib.ptr(cnt),
                                      _lib.stream())
    _lib.check(rc, "swat_segment_topk")
    return idx, val, cnt


def i2i_similarity_p2p(fewshot_embedding, img_embeddings, mode):
    """ref: retrieval/sample_retrieval.py:369-394 — similarity of every row of img_embeddings [n, d] to the few-shot
    embeddings (a list of d-vectors or an [m, d] array), reduced over the few-shot axis by 'min' | 'max' | 'mean';
    returns a Python list of n floats like the reference.  The [m, n] product is the canonical k-ordered fp32 chain
    (swat_sim_topk_candidates), the reduction runs over the few-shot rows in order — bit-identical to
    oracle/retrieval_oracle.c:orc_similarity_p2p."""
    import numpy as np
    from . import dense
    if mode not in ("min", "max", "mean"):
        raise ValueError('Invalid mode type.')
    if not isinstance(img_embeddings, torch.Tensor) or not img_embeddings.is_cuda:
        raise _lib.SwatHipError("i2i_similarity_p2p needs the image embeddings on the GPU (no CPU fallback)")
    fs = fewshot_embedding if isinstance(fewshot_embedding, torch.Tensor) else torch.from_numpy(np.stack(fewshot_embedding))
    fs = fs.to(img_embeddings.device, torch.float32).contiguous()
    S = dense.sim_matrix(fs, img_embeddings.float().contiguous())  # [m, n]
    r = S[0].clone()
    for j in range(1, S.shape[0]):
        r = torch.minimum(r, S[j]) if mode == "min" else torch.maximum(r, S[j]) if mode == "max" else r + S[j]
    if mode == "mean":
        r = r / float(S.shape[0])
    return r.cpu().tolist()
