"""Row-sharded retrieval across the GPUs of one node: every rank holds a contiguous slice of the
class-grouped index and runs the local filtered top-k; one gather of [n_seg, k] (value, global row) lists and a
k-way merge by (value desc, global row asc) finish the query.  No other exchange on the data path.

ref: the single-GPU sampler loop retrieval/sample_retrieval.py:724-825 is the semantics being sharded.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world, rank):
    """Contiguous row range [lo, hi) of `rank` (first n_rows % world ranks get one extra row)."""
    q, r = divmod(n_rows, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def local_segments(offsets, lo, hi):
    """Clip global CSR class offsets to the shard [lo, hi) and rebase them to shard-local rows."""
    return (offsets.clamp(lo, hi) - lo).contiguous()


def merge_topk(idx_list, val_list, k):
    """idx/val per shard: [n_seg, k] (global rows, -1 padded / -inf), shards in rank order -> the k best of every
    query by (value desc, global row asc) — the reference's stable sorted(reverse=True) order, since rows of a lower
    rank are lower global rows.  On the GPU: swat_merge_topk_lists (the library's own bitonic top-k over the packed
    (value, ~row) keys) for up to 8 192 gathered entries per query.  CPU tensors (the gloo tests of the merge bookkeeping) and longer
    rows (e.g. 8 ranks x k = 2 048) use a stable descending sort — the same order by construction."""
    idx = torch.cat(idx_list, dim=1).contiguous()
    val = torch.cat(val_list, dim=1).contiguous()
    if idx.is_cuda and idx.shape[1] <= 8192:   # (the library's merge sorts up to 8 192 keys per query; longer rows: the stable sort below)
        from .. import _lib
        L = _lib.lib()
        nq, m = idx.shape
        kk = min(k, m)
        ws = torch.empty(L.swat_merge_topk_workspace_bytes(nq, m), dtype=torch.uint8, device=idx.device)
        out_idx = torch.empty((nq, kk), dtype=torch.int64, device=idx.device)
        out_val = torch.empty((nq, kk), dtype=torch.float32, device=idx.device)
        cnt = torch.empty(nq, dtype=torch.int64, device=idx.device)
        _lib.check(L.swat_merge_topk_lists(_lib.ptr(idx), _lib.ptr(val), nq, m, kk, _lib.ptr(ws), _lib.ptr(out_idx), _lib.ptr(out_val),
                                           _lib.ptr(cnt), _lib.stream()), "swat_merge_topk_lists")
        return out_idx, out_val, cnt
    val = torch.where(idx >= 0, val, torch.full_like(val, float("-inf")))
    order = torch.sort(val, dim=1, descending=True, stable=True).indices[:, :k]
    out_idx = torch.gather(idx, 1, order)
    out_val = torch.gather(val, 1, order)
    cnt = (out_idx >= 0).sum(dim=1)
    return out_idx, out_val, cnt


def gather_and_merge(idx, val, k, group=None):
    """All ranks call; every rank returns the merged global lists (all_gather keeps it symmetric and tiny:
    12 B x n_seg x k per rank)."""
    world = dist.get_world_size(group)
    idxs = [torch.empty_like(idx) for _ in range(world)]
    vals = [torch.empty_like(val) for _ in range(world)]
    dist.all_gather(idxs, idx.contiguous(), group=group)
    dist.all_gather(vals, val.contiguous(), group=group)
    return merge_topk(idxs, vals, k)


class ShardedIndex:
    """One rank's row range of a saved SegmentedIndex (index.SegmentedIndex.save) plus the gather + merge that finishes
    a query.  Every rank calls sample() with the same queries; every rank gets the global lists (grouped global rows)."""

    def __init__(self, path, device="cuda", group=None):
        import json
        import os
        from .index import SegmentedIndex
        self.group = group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        n = json.load(open(os.path.join(path, "meta.json")))["n_rows"]
        self.lo, self.hi = shard_bounds(n, world, rank)
        self.local = SegmentedIndex.load(path, device=device, rows=(self.lo, self.hi))

    def sample(self, queries, k, **kw):
        idx, val, cnt, _, _ = self.local.sample(queries, k, **kw)
        gidx = torch.where(idx >= 0, idx + self.lo, idx)
        return gather_and_merge(gidx, val, k, group=self.group)


class DenseIndex:
    """One rank's shard of a dense (N_query x N_db) cosine index — BASELINE config 4: `rows` [n, d] fp32 on the GPU with global
    row numbers row_base .. row_base + n - 1.  Owns the shard's bf16 scan image (dense.DenseShadow, built once; pass
    shadow=False to scan the fp32 rows instead and save half the shard's bytes again in HBM) and finishes a query batch: local
    exact top-k, then, inside a process group, the all_gather + k-way merge of the per-rank lists (SURVEY 8(e) retrieval row).
    ref: the per-class GEMV + sorted() of retrieval/sample_retrieval.py:724-825, generalised to a query batch."""

    def __init__(self, rows, row_base=0, group=None, shadow=True):
        from . import dense
        self.rows, self.row_base, self.group = rows, int(row_base), group
        self.shadow = None
        if shadow and rows.shape[1] % 128 == 0:
            self.shadow = dense.DenseShadow(rows)

    def topk(self, queries, k):
        """-> (idx [n_q, k] global rows (-1 padded), val [n_q, k], cnt [n_q]) in (value desc, row asc) order, bit-exact."""
        from . import dense
        idx, val, cnt = dense.topk_dense(self.rows, queries, k, row_base=self.row_base, shadow=self.shadow)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            return gather_and_merge(idx, val, k, group=self.group)
        return idx, val, cnt
