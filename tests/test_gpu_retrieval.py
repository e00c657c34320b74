"""GPU: path B HIP kernels (through the C ABI) against the C oracle and the reference's golden vectors."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import retrieval_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _load(golden_dir, kind):
    g = np.load(os.path.join(golden_dir, f"retrieval_{kind}.npz"), allow_pickle=False)
    spec = json.loads(str(g["spec"]))
    return g, spec, synth.retrieval_case(**spec)


@pytest.mark.parametrize("kind", ["grid", "float"])
def test_group_by_matches_reference(golden_dir, kind):
    from swat_amd.retrieval import group_by_label
    g, spec, case = _load(golden_dir, kind)
    order, cls, off = group_by_label(torch.from_numpy(case["labels"]).to(_dev()))
    assert np.array_equal(order.cpu().numpy(), g["group_order"])
    assert np.array_equal(cls.cpu().numpy(), g["group_classes"])
    assert np.array_equal(np.diff(off.cpu().numpy()), g["group_sizes"])


@pytest.mark.parametrize("kind", ["grid", "float"])
@pytest.mark.parametrize("name,kw", [("t2t", dict(rank_by="t2t", use_t2i_thr=False)),
                                     ("t2t_t2i", dict(rank_by="t2t", use_t2i_thr=True)),
                                     ("t2i", dict(rank_by="t2i"))])
def test_samplers_match_reference_golden(golden_dir, kind, name, kw):
    """Bit-exact kept rows / counts / <prefix>.txt bytes vs what the reference's own samplers produced."""
    from swat_amd.retrieval import SegmentedIndex
    g, spec, case = _load(golden_dir, kind)
    dev = _dev()
    index = SegmentedIndex(torch.from_numpy(case["img"]), torch.from_numpy(case["cap"]),
                           torch.from_numpy(case["labels"]), case["paths"], device=dev)
    excl = torch.from_numpy((case["excl_dup"] | case["excl_filt"]).astype(np.uint8)).to(dev)[index.order].contiguous()
    q = torch.from_numpy(case["queries"]).to(dev)
    idx, val, cnt, sims, t2i = index.sample(q, spec["k"], excluded=excl, **kw)
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    order = index.order.cpu().numpy()
    rows = np.concatenate([order[idx[s, :cnt[s]]] for s in range(index.n_seg)])
    assert np.array_equal(rows, g[f"{name}_rows"])
    assert {str(int(c)): int(n) for c, n in zip(index.class_ids.tolist(), cnt)} == json.loads(str(g[f"{name}_counts"]))
    assert orc.sample_file_list_text(index.filepath, index.class_ids.tolist(), idx, cnt) == str(g[f"{name}_txt"])
    assert (idx[np.arange(idx.shape[1])[None, :] >= cnt[:, None]] == -1).all()
    # similarities: bit-exact vs the oracle's canonical fma chain (and vs the reference on the exact-arithmetic grid)
    src = case["img"] if kw["rank_by"] == "t2i" else case["cap"]
    o_order, o_cls, o_off = orc.group_by_label(case["labels"])
    _, _, o_sims, _ = orc.sample_index(case["cap"][o_order], case["img"][o_order], case["queries"], o_off, spec["k"],
                                       return_sims=True, **kw)
    assert np.array_equal(sims.cpu().numpy(), o_sims)
    if kind == "grid":
        ref = g["sims_t2i" if kw["rank_by"] == "t2i" else "sims_t2t"][order]
        assert np.array_equal(sims.cpu().numpy(), ref)


@pytest.mark.parametrize("n,d,nseg,k", [(1, 512, 1, 5), (63, 512, 1, 64), (64, 64, 2, 1), (65, 768, 3, 500),
                                         (1000, 1024, 7, 2048), (20000, 512, 20, 500), (70000, 512, 3, 500)])
def test_sim_and_topk_vs_oracle_seeded(n, d, nseg, k):
    """Seeded random + ragged segments (incl. empty ones), ties, -0.0, masks: bit-exact vs the C oracle."""
    from swat_amd.retrieval import ops
    dev = _dev()
    rs = np.random.RandomState(n * 7 + d)
    feats = synth.unit_rows(rs, n, d)
    img = synth.unit_rows(rs, n, d)
    for _ in range(n // 10):  # exact ties
        a, b = rs.randint(n), rs.randint(n)
        feats[a] = feats[b]
    queries = synth.unit_rows(rs, nseg, d)
    cuts = np.sort(rs.randint(0, n + 1, size=nseg - 1)) if nseg > 1 else np.zeros(0, np.int64)
    off = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    if nseg > 2:
        off[2] = off[1]  # an empty segment
    row_seg = np.repeat(np.arange(nseg, dtype=np.int32), np.diff(off))
    excl = (rs.rand(n) < 0.1).astype(np.uint8)
    o_idx, o_cnt, o_sims, o_t2i = orc.sample_index(feats, img, queries, off, k, use_t2i_thr=True, t2i_thr=0.0,
                                                   excluded=excl, return_sims=True)
    f = torch.from_numpy(feats).to(dev)
    i = torch.from_numpy(img).to(dev)
    q = torch.from_numpy(queries).to(dev)
    rsg = torch.from_numpy(row_seg).to(dev)
    sims = ops.similarity_segmented(f, q, rsg)
    t2i = ops.similarity_segmented(i, q, rsg)
    assert np.array_equal(sims.cpu().numpy(), o_sims)
    assert np.array_equal(t2i.cpu().numpy(), o_t2i)
    idx, val, cnt = ops.segment_topk(sims, torch.from_numpy(off).to(dev), k, thr=0.0, t2i=t2i, t2i_thr=0.0,
                                     excluded=torch.from_numpy(excl).to(dev))
    assert np.array_equal(cnt.cpu().numpy(), o_cnt)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    got_val = val.cpu().numpy()
    for s in range(nseg):
        assert np.array_equal(got_val[s, :o_cnt[s]], o_sims[o_idx[s, :o_cnt[s]]])


def test_topk_negative_zero_and_all_filtered():
    from swat_amd.retrieval import ops
    dev = _dev()
    s = torch.tensor([0.0, -0.0, 0.0, 1.0, -0.0, -1.0, -2.0], device=dev)
    off = torch.tensor([0, 5, 7, 7], device=dev)
    idx, val, cnt = ops.segment_topk(s, off, 5, thr=0.0)
    assert idx.cpu().tolist() == [[3, 0, 1, 2, 4], [-1] * 5, [-1] * 5]
    assert cnt.cpu().tolist() == [5, 0, 0]
    idx, val, cnt = ops.segment_topk(s, off, 2, thr=-10.0)
    assert idx.cpu().tolist() == [[3, 0], [5, 6], [-1, -1]]


def test_full_size_properties():
    """BASELINE-size shard (1.25 M x 512, the per-GPU share of 10 M rows): size-independent properties."""
    from swat_amd.retrieval import ops
    dev = _dev()
    n, d, nseg, k = 1_250_000, 512, 125, 500
    g = torch.Generator(device=dev).manual_seed(7)
    feats = torch.randn(n, d, device=dev, generator=g)
    feats /= feats.norm(dim=-1, keepdim=True)
    q = torch.randn(nseg, d, device=dev, generator=g)
    q /= q.norm(dim=-1, keepdim=True)
    off = torch.arange(0, n + 1, n // nseg, device=dev, dtype=torch.int64)
    row_seg = torch.repeat_interleave(torch.arange(nseg, device=dev, dtype=torch.int32), n // nseg)
    sims = ops.similarity_segmented(feats, q, row_seg)
    # linearity / agreement with an independent fp64 product on a slice
    ref = (feats[:4096].double() * q[row_seg[:4096].long()].double()).sum(-1)
    assert (sims[:4096].double() - ref).abs().max().item() < 1e-6
    idx, val, cnt = ops.segment_topk(sims, off, k, thr=-2.0)
    assert (cnt == k).all()
    # sortedness (value desc, row asc on ties), membership in the right segment, idempotence
    assert (val[:, 1:] <= val[:, :-1]).all()
    ties = val[:, 1:] == val[:, :-1]
    assert (idx[:, 1:][ties] > idx[:, :-1][ties]).all()
    assert ((idx >= off[:-1, None]) & (idx < off[1:, None])).all()
    assert torch.equal(sims[idx], val)
    # the k-th value really is the k-th largest of its segment
    seg_sims = sims.view(nseg, -1)
    assert torch.equal(torch.topk(seg_sims, k, dim=1).values, val)
    idx2, val2, _ = ops.segment_topk(sims, off, k, thr=-2.0)
    assert torch.equal(idx, idx2) and torch.equal(val, val2)
    # oracle on the first two segments (20 k rows: seconds on the CPU)
    sub = feats[: 2 * (n // nseg)].cpu().numpy()
    o_idx, o_cnt = orc.sample_index(sub, None, q[:2].cpu().numpy(), off[:3].cpu().numpy(), k, thr=-2.0)
    assert np.array_equal(idx[:2].cpu().numpy(), o_idx)


@pytest.mark.parametrize("n,nq,k", [(3000, 5, 17), (70000, 33, 500), (200001, 200, 500), (50000, 65, 100), (30001, 225, 64),
                                    (20000, 449, 32)])
def test_dense_mfma_topk_bit_exact_vs_oracle(n, nq, k):
    """(N_query x N_db) scan on the fp32 matrix cores: same bits as the lane-per-row kernel and the C oracle."""
    from swat_amd.retrieval import dense, ops
    dev = _dev()
    rs = np.random.RandomState(n + nq)
    db = synth.unit_rows(rs, n, 512)
    for _ in range(n // 20):
        a, b = rs.randint(n), rs.randint(n)
        db[a] = db[b]
    q = synth.unit_rows(rs, nq, 512)
    d_db, d_q = torch.from_numpy(db).to(dev), torch.from_numpy(q).to(dev)
    idx, val, cnt = dense.topk_dense(d_db, d_q, k, row_base=0)
    chk = min(nq, 4)  # the oracle sorts all n rows per query: keep the CPU leg to seconds
    o_idx, o_val = orc.topk_multi(db, q[:chk], k)
    assert np.array_equal(idx[:chk].cpu().numpy(), o_idx)
    assert np.array_equal(val[:chk].cpu().numpy(), o_val)
    assert (cnt == min(k, n)).all()
    # every query: MFMA similarities == lane-per-row scan kernel, bit for bit; lists sorted with the tie rule
    S = dense.sim_matrix(d_q[:8].contiguous(), d_db[:4096].contiguous())
    for i in range(min(8, nq)):
        s = ops.similarity_segmented(d_db[:4096].contiguous(), d_q[i:i + 1].contiguous())
        assert torch.equal(S[i], s)
    assert (val[:, 1:] <= val[:, :-1]).all()
    ties = val[:, 1:] == val[:, :-1]
    assert (idx[:, 1:][ties] > idx[:, :-1][ties]).all()


@pytest.mark.parametrize("n,nq,k,scale,d", [(150001, 200, 500, 1.0, 512), (40000, 37, 100, 1.0, 512), (60000, 64, 256, 3.5, 512),
                                             (5000, 10, 20, 1.0, 512), (70003, 150, 300, 1.0, 768), (30000, 230, 64, 2.0, 128), (50000, 230, 64, 1.0, 512),
                                             (30000, 40, 64, 1.0, 192)])
def test_dense_bf16_candidate_pass_equals_the_exact_scan(n, nq, k, scale, d):
    """topk_dense = bf16 MFMA candidate pass + exact re-scoring + on-device completeness proof: the same (rows, values) as
    the exact f32-matrix-core scan for EVERY query, duplicates (exact ties) included, with a row_base and with rows that
    are not unit-norm (the error bound scales with |q| |d|); and the fast route is the one that ran.  d % 128 == 0 takes
    the packed-query deep kernel (one to three passes of 13 / 8 / 2 query tiles), d = 192 the row-major one."""
    from swat_amd.retrieval import dense
    dev = _dev()
    rs = np.random.RandomState(n * 7 + nq)
    db = synth.unit_rows(rs, n, d) * np.float32(scale)
    for _ in range(n // 10):
        a, b = rs.randint(n), rs.randint(n)
        db[a] = db[b]
    centres = synth.unit_rows(rs, nq, d)
    db[rs.randint(n, size=4 * nq)] += np.repeat(centres, 4, axis=0) * np.float32(0.4)  # a few strong hits per query
    d_db, d_q = torch.from_numpy(db).to(dev), torch.from_numpy(centres).to(dev)
    idx, val, cnt = dense.topk_dense(d_db, d_q, k, row_base=1000)
    assert dense.last_path == "bf16-candidates+exact-rescore"
    e_idx, e_val, e_cnt = dense.topk_dense_exact(d_db, d_q, k, row_base=1000)
    assert torch.equal(idx, e_idx) and torch.equal(val, e_val) and torch.equal(cnt, e_cnt)
    # second call: cached thresholds / bf16 queries, same answer; an in-place change of the queries invalidates the cache
    idx2, val2, _ = dense.topk_dense(d_db, d_q, k, row_base=1000)
    assert torch.equal(idx2, idx) and torch.equal(val2, val)
    # the completeness proof as a launch of its own (swat_dense_topk_proof) agrees with the one folded into the last sort
    # (swat_candidates_topk_proof): zero failures on these buffers, every query failing once the error bound is made absurd
    from swat_amd import _lib
    from swat_amd.retrieval import ops as rops
    L, c = _lib.lib(), dense._prepared(d_db, d_q, k, 16384)
    c["cnt"].zero_(); c["n2"].zero_()   # (topk_dense leaves the counters cleared: rebuild the intermediate state stage by stage)
    dense.candidates(c, d_db, 1000)
    _lib.check(L.swat_candidates_select(_lib.ptr(c["keys"]), dense.CAP, _lib.ptr(c["cnt"]), nq, c["K2"], _lib.ptr(c["idx2"]), _lib.ptr(c["val2"]),
                                        _lib.ptr(c["cnt2"]), None, _lib.stream()), "swat_candidates_select")
    rops.sim_gathered(d_db, c["idx2"].view(-1), d_q, c["row_seg"], keys_out=c["keys2"], row_base=1000, want_sims=False)
    fails = torch.full((1,), 7, dtype=torch.int32, device=dev)
    for eps, want in ((dense.EPS_BF16, 0), (1e3, nq)):
        fails.zero_()
        i2, v2, c2 = torch.empty_like(idx), torch.empty_like(val), torch.empty_like(cnt)
        _lib.check(L.swat_candidates_topk_proof(_lib.ptr(c["keys2"]), c["K2"], _lib.ptr(c["cnt"]), nq, k, _lib.ptr(i2), _lib.ptr(v2), _lib.ptr(c2),
                                                _lib.ptr(c["cnt"]), dense.CAP, c["K2"], _lib.ptr(c["val2"]), _lib.ptr(c["thr"]), min(k, n), float(eps),
                                                _lib.ptr(c["n2"]), _lib.ptr(c["qn2"]), _lib.ptr(fails), 0, _lib.stream()), "swat_candidates_topk_proof")
        assert fails.item() == want and torch.equal(i2, idx) and torch.equal(v2, val) and torch.equal(c2, cnt)
        fails.fill_(7)
        _lib.check(L.swat_dense_topk_proof(_lib.ptr(c["cnt"]), dense.CAP, c["K2"], _lib.ptr(c["val2"]), _lib.ptr(c["thr"]), _lib.ptr(v2), k,
                                           _lib.ptr(c2), min(k, n), float(eps), _lib.ptr(c["n2"]), _lib.ptr(c["qn2"]), nq, _lib.ptr(fails),
                                           _lib.stream()), "swat_dense_topk_proof")
        assert fails.item() == want
    c["cnt"].zero_()
    if d % 128 == 0:
        # the bf16 shadow of the shard (half the bytes per row): the same candidate lists bit for bit, hence the same answer
        sh = dense.DenseShadow(d_db)
        c = dense._prepared(d_db, d_q, k, 16384)
        c["cnt"].zero_(); dense.candidates(c, d_db, 1000)
        n_f32 = c["cnt"].clone(); keys_f32 = torch.sort(c["keys"][0, :int(n_f32[0])]).values.clone(); n2_f32 = c["n2"].clone()
        c["cnt"].zero_(); c["n2"].zero_(); dense.candidates(c, d_db, 1000, shadow=sh)
        assert torch.equal(c["cnt"], n_f32) and torch.equal(torch.sort(c["keys"][0, :int(n_f32[0])]).values, keys_f32)
        assert abs(c["n2"].item() - n2_f32.item()) <= 1e-5 * n2_f32.item()
        idx4, val4, cnt4 = dense.topk_dense(d_db, d_q, k, row_base=1000, shadow=sh)
        assert dense.last_path == "bf16-shadow-candidates+exact-rescore"
        assert torch.equal(idx4, e_idx) and torch.equal(val4, e_val) and torch.equal(cnt4, e_cnt)
        from swat_amd.retrieval import shard
        di = shard.DenseIndex(d_db, row_base=1000)   # the index object owns the shadow (no process group: local lists)
        idx5, val5, cnt5 = di.topk(d_q, k)
        assert dense.last_path == "bf16-shadow-candidates+exact-rescore"
        assert torch.equal(idx5, e_idx) and torch.equal(val5, e_val) and torch.equal(cnt5, e_cnt)
        d_db[0].mul_(1.0)  # an in-place write to the rows: the shadow no longer vouches for them, the fp32 scan answers
        assert not sh.matches(d_db)
        dense.topk_dense(d_db, d_q, k, row_base=1000, shadow=sh)
        assert dense.last_path == "bf16-candidates+exact-rescore"
    d_q.mul_(-1.0)
    idx3, val3, _ = dense.topk_dense(d_db, d_q, k, row_base=1000)
    e3_idx, e3_val, _ = dense.topk_dense_exact(d_db, d_q, k, row_base=1000)
    assert torch.equal(idx3, e3_idx) and torch.equal(val3, e3_val)


def test_dense_edge_shapes_and_the_staging_overflow_path():
    """Tiny / ragged shards, one query, k near n, through the shadow scan, in ONE process and over poisoned memory: with n <= 4096
    every (query, row) is a candidate, 17 rows x 33 queries overflow a wave's staging slot (the overflow bit of cnt[0] is set while
    other workgroups are still filing query 0's records — once taken as part of their position, which left its list unwritten and
    sent the re-score to garbage rows: a GPU memory fault), and the call must fall back to the exact scan, not crash."""
    from swat_amd.retrieval import dense
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(5)
    for d, n, nq, k in ((256, 17, 1, 500), (256, 17, 33, 1), (512, 1, 1, 1), (512, 255, 33, 10), (768, 257, 200, 500), (1024, 300, 1, 10),
                        (256, 4097, 200, 500), (512, 9001, 33, 10)):
        poison = torch.full((64 << 20,), -1, dtype=torch.int8, device=dev)   # the next allocations reuse these bytes
        del poison
        db = torch.randn(n, d, device=dev, generator=g)
        db /= db.norm(dim=-1, keepdim=True)
        q = torch.randn(nq, d, device=dev, generator=g)
        q /= q.norm(dim=-1, keepdim=True)
        i1, v1, c1 = dense.topk_dense(db, q, k, row_base=7, shadow=dense.DenseShadow(db))
        path = dense.last_path
        i2, v2, c2 = dense.topk_dense_exact(db, q, k, row_base=7)
        assert torch.equal(i1, i2) and torch.equal(v1, v2) and torch.equal(c1, c2), (d, n, nq, k, path)
        if (n, nq) == (17, 33):
            assert path == "exact-fallback"
            # a warm repeat on the same tensors: the cache entry remembers that the proof failed (no candidate pass + host sync + exact scan
            # per call), still holds the thresholds its key stands for, and the answer is the same
            thr_before = dense._cache["thr"].clone()
            i3, v3, c3 = dense.topk_dense(db, q, k, row_base=7)
            assert dense.last_path == "exact-fallback" and dense._cache.get("proof_failed")
            assert torch.equal(i3, i2) and torch.equal(v3, v2) and torch.equal(c3, c2) and torch.equal(dense._cache["thr"], thr_before)


def test_dense_topk_beyond_the_error_bound_width_takes_the_exact_scan():
    """d = 2048 > 1024: the bf16 error bound of the completeness proof does not cover the fp32 accumulation any more (ADVICE
    r02), so topk_dense answers with the exact scan — and equals the oracle."""
    from swat_amd.retrieval import dense
    rs = np.random.RandomState(3)
    db, q = synth.unit_rows(rs, 6000, 2048), synth.unit_rows(rs, 5, 2048)
    idx, val, cnt = dense.topk_dense(torch.from_numpy(db).to(_dev()), torch.from_numpy(q).to(_dev()), 32)
    assert dense.last_path == "exact"
    o_idx, o_val = orc.topk_multi(db, q, 32)
    assert np.array_equal(idx.cpu().numpy(), o_idx) and np.array_equal(val.cpu().numpy().view(np.uint32), o_val.view(np.uint32))


def test_dense_topk_full_shard_vs_oracle():
    """BASELINE config 4 at its per-GPU size: 200 queries x 1.25 M x 512 rows, k = 500.  Every query against the exact
    f32-matrix-core scan, and 4 of them against the C oracle over the whole shard (orc.topk_multi: ~1.3 G fma per query)."""
    from swat_amd.retrieval import dense
    dev = _dev()
    n, d, nq, k = 1_250_000, 512, 200, 500
    g = torch.Generator(device=dev).manual_seed(7)
    q = torch.randn(nq, d, device=dev, generator=g)
    q /= q.norm(dim=-1, keepdim=True)
    db = torch.randn(n, d, device=dev, generator=g)
    db /= db.norm(dim=-1, keepdim=True)
    idx, val, cnt = dense.topk_dense(db, q, k)
    assert dense.last_path == "bf16-candidates+exact-rescore"
    e_idx, e_val, e_cnt = dense.topk_dense_exact(db, q, k)
    assert torch.equal(idx, e_idx) and torch.equal(val, e_val) and torch.equal(cnt, e_cnt)
    # the route bench.py reports (the shard's bf16 shadow scanned for candidates, exact fp32 re-score, on-device proof), on the
    # whole shard: the first (cold: new query tensor) call and a repeated one
    sh = dense.DenseShadow(db)
    for _ in range(2):
        s_idx, s_val, s_cnt = dense.topk_dense(db, q, k, shadow=sh)
        assert dense.last_path == "bf16-shadow-candidates+exact-rescore"
        assert torch.equal(s_idx, e_idx) and torch.equal(s_val, e_val) and torch.equal(s_cnt, e_cnt)
    pick = [0, 57, 123, 199]
    o_idx, o_val = orc.topk_multi(db.cpu().numpy(), q[pick].cpu().numpy(), k)
    for got_i, got_v in ((idx, val), (s_idx, s_val)):   # both routes against the C oracle directly
        assert np.array_equal(got_i[pick].cpu().numpy(), o_idx)
        assert np.array_equal(got_v[pick].cpu().numpy().view(np.uint32), o_val.view(np.uint32))


def test_dense_cache_is_keyed_on_tensor_identity_not_on_addresses():
    """A new query tensor that reuses a freed tensor's address (same shape, version 0) must not hit the cached thresholds /
    packed queries of the old one."""
    from swat_amd.retrieval import dense
    dev = _dev()
    rs = np.random.RandomState(2)
    db = torch.from_numpy(synth.unit_rows(rs, 30000, 256)).to(dev)
    outs = []
    for i in range(3):
        q = torch.from_numpy(synth.unit_rows(np.random.RandomState(10 + i), 24, 256)).to(dev)
        ptr = q.data_ptr()
        idx, val, _ = dense.topk_dense(db, q, 40)
        e_idx, e_val, _ = dense.topk_dense_exact(db, q, 40)
        assert torch.equal(idx, e_idx) and torch.equal(val, e_val), i
        outs.append(ptr)
        del q, idx, val, e_idx, e_val
    # (the caching allocator normally hands the same block back: that is the case the test is about, but not required)


def test_packed_query_image_layout():
    """swat_dense_pack_queries: per (32-deep chunk, 16-query tile) a 1 KiB block whose 16-B unit L holds query 16T + L/4,
    k-slot s = (L % 4) ^ (3 * ((L / 32) % 2)) = k 32c + 4s .. +3 and 32c + 16 + 4s .. +3 (the order in which the scan reads a
    database row), rounded to bf16 (RNE); rows >= n_q and the 13 slack tiles are zero.  The swizzle makes the scan's fragment read
    (lane r16 + 16 g -> unit 4 r16 + (g ^ slot swizzle)) conflict-free in each of the four lane groups ds_read_b128 is served in."""
    from swat_amd import _lib
    dev = _dev()
    nq, d = 37, 256
    q = torch.randn(nq, d, device=dev)
    nbytes = int(_lib.lib().swat_dense_packed_bytes(nq, d))
    tiles = (nq + 15) // 16 + 13
    assert nbytes == (d // 32) * tiles * 1024 and _lib.lib().swat_dense_packed_bytes(nq, 192) == 0
    out = torch.full((nbytes // 2,), 0x7fff, dtype=torch.int16, device=dev)
    _lib.check(_lib.lib().swat_dense_pack_queries(_lib.ptr(q), nq, d, _lib.ptr(out), _lib.stream()), "pack")
    got = out.view(torch.bfloat16).view(d // 32, tiles, 64, 8).float().cpu()
    qb = torch.zeros(tiles * 16, d)
    qb[:nq] = q.to(torch.bfloat16).float().cpu()
    L = torch.arange(64)
    slot = (L % 4) ^ (3 * ((L // 32) % 2))
    for group in ([0, 1, 2, 3, 12, 13, 14, 15] + list(range(20, 28)), list(range(4, 12)) + [16, 17, 18, 19, 28, 29, 30, 31]):
        for base in (0, 32):   # MI355X_MICROARCH.md: the groups of ds_read_b128; a 16-B unit U covers banks 4 (U % 16) .. + 3
            units = [4 * (ln % 16) + ((ln // 16) ^ (3 * (((ln % 16) >> 3) & 1))) for ln in (x + base for x in group)]
            assert len({u % 16 for u in units}) == 16
    for c in range(d // 32):
        for T in (0, 1, 2, 3, tiles - 1):
            want = torch.stack([torch.cat([qb[16 * T + int(L[i]) // 4, 32 * c + 4 * int(slot[i]):32 * c + 4 * int(slot[i]) + 4],
                                           qb[16 * T + int(L[i]) // 4, 32 * c + 16 + 4 * int(slot[i]):32 * c + 20 + 4 * int(slot[i])]])
                                for i in range(64)])
            assert torch.equal(got[c, T], want), (c, T)


def test_sim_gathered_is_the_scan_on_a_row_list():
    from swat_amd.retrieval import ops
    dev = _dev()
    rs = np.random.RandomState(4)
    db = torch.from_numpy(synth.unit_rows(rs, 5000, 512)).to(dev)
    q = torch.from_numpy(synth.unit_rows(rs, 3, 512)).to(dev)
    rows = torch.from_numpy(rs.randint(0, 5000, size=3 * 128)).to(dev)
    rows[5] = -1
    seg = torch.arange(3, device=dev, dtype=torch.int32).repeat_interleave(128)
    got = ops.sim_gathered(db, rows, q, seg)
    for s in range(3):
        full = ops.similarity_segmented(db, q[s:s + 1].contiguous())
        r = rows[128 * s:128 * (s + 1)]
        want = torch.where(r >= 0, full[r.clamp_min(0)], torch.tensor(float("-inf"), device=dev))
        assert torch.equal(got[128 * s:128 * (s + 1)], want)


@pytest.mark.parametrize("kind", ["grid", "float"])
@pytest.mark.parametrize("mode", ["min", "max", "mean"])
def test_p2p_similarity_matches_oracle_and_reference_golden(golden_dir, kind, mode):
    """i2i_similarity_p2p (ref :369-394): bit-exact against the C oracle; against the imported reference's output exact on
    the grid case (min / max) and within 1e-6 elsewhere (its BLAS order is unspecified)."""
    from swat_amd.retrieval import ops
    g, spec, case = _load(golden_dir, kind)
    c0 = min(case["class_ids"])
    rows = np.nonzero(case["labels"] == c0)[0]
    img = torch.from_numpy(case["img"][rows]).to(_dev())
    got = np.array(ops.i2i_similarity_p2p([case["img"][r] for r in rows[:16]], img, mode), np.float32)
    assert np.array_equal(got.view(np.uint32), orc.similarity_p2p(case["img"][rows], case["img"][rows[:16]], mode).view(np.uint32))
    if kind == "grid" and mode != "mean":
        assert np.array_equal(got, g[f"p2p_{mode}"])
    else:
        assert np.abs(got - g[f"p2p_{mode}"]).max() <= 1e-6
    with pytest.raises(ValueError):
        ops.i2i_similarity_p2p([case["img"][0]], img, "median")


@pytest.mark.parametrize("kind", ["grid", "float"])
def test_near_dup_matches_reference_golden(golden_dir, kind):
    from swat_amd.retrieval import SegmentedIndex, dense
    g, spec, case = _load(golden_dir, kind)
    dev = _dev()
    index = SegmentedIndex(torch.from_numpy(case["img"]), torch.from_numpy(case["cap"]), torch.from_numpy(case["labels"]),
                           case["paths"], device=dev)
    dups, fracs, avg = dense.remove_near_duplicates2(index)
    path_to_row = {p: i for i, p in enumerate(case["paths"])}
    rows = np.array(sorted(path_to_row[p] for c in dups for p in dups[c]), np.int64)
    assert np.array_equal(rows, g["dedup_rows"])
    assert np.allclose(fracs, g["dedup_frac"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("kind", ["grid", "float"])
@pytest.mark.parametrize("method,name", [("T2T-rank", "t2t"), ("T2T-rank-T2I-tshd", "t2t_t2i"), ("T2I-rank", "t2i")])
def test_sampling_driver_writes_reference_bytes(golden_dir, tmp_path, kind, method, name):
    """sampling() with the reference's plug-in signature: `<prefix>.txt` bytes, per-class counts and the returned
    mined_split equal what the reference's own samplers + save_sample_file_list produced."""
    import types
    from collections import defaultdict
    from swat_amd.retrieval import samplers
    g, spec, case = _load(golden_dir, kind)
    mined = {"image_features": torch.from_numpy(case["img"]), "caption_features": torch.from_numpy(case["cap"]),
             "labels": torch.from_numpy(case["labels"]), "filepath": case["paths"]}
    prompt_tensors = {str(c): {"mean": torch.from_numpy(case["queries"][i])} for i, c in enumerate(case["class_ids"])}
    dup, filt = defaultdict(set), defaultdict(set)
    for p, e in zip(case["paths"], case["excl_dup"]):
        if e:
            dup[p.split("/")[-2]].add(p)
    for p, e in zip(case["paths"], case["excl_filt"]):
        if e:
            filt[p.split("/")[-2]].add(p)

    class Lg:
        def info(self, *a):
            pass

    args = types.SimpleNamespace(output_folder=str(tmp_path), prefix="golden", sampling_method=method, num_samples=spec["k"])
    index = samplers.transform_extracted_fea(mined)
    split, counts = samplers.SAMPLERS[method](args, Lg(), prompt_tensors, spec["k"], 0.0, index, dup, filt)
    files = [p for fl in split["file_list"] for p in fl]
    labels = torch.cat(split["label_list"])
    samplers.save_sample_file_list(args, files, labels)
    assert open(tmp_path / "golden.txt").read() == str(g[f"{name}_txt"])
    assert {c: n for c, n in counts.items()} == json.loads(str(g[f"{name}_counts"]))
    path_to_row = {p: i for i, p in enumerate(case["paths"])}
    rows = np.array([path_to_row[p] for p in files], np.int64)
    assert torch.equal(torch.cat(split["feature_list"]), torch.from_numpy(case["img"][rows]))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_saved_index_shards_merge_to_the_single_gpu_answer(tmp_path, world):
    """SURVEY 8(e) on one GPU: save the grouped index, load every rank's row range from disk, run the local filtered
    top-k per shard on the HIP kernels and merge the [n_seg, k] lists in rank order — bit-identical (rows and values,
    ties included) to the unsharded index."""
    from swat_amd.retrieval import shard
    from swat_amd.retrieval.index import SegmentedIndex
    dev = _dev()
    rs = np.random.RandomState(77)
    n, d, ncls, k = 30000, 512, 23, 500
    cap, img = synth.unit_rows(rs, n, d), synth.unit_rows(rs, n, d)
    for _ in range(2000):  # exact ties, many of them across shard boundaries
        a, b = rs.randint(n), rs.randint(n)
        cap[a] = cap[b]
    labels = torch.from_numpy(rs.randint(0, ncls, size=n))
    full = SegmentedIndex(torch.from_numpy(img), torch.from_numpy(cap), labels, filepath=[str(i) for i in range(n)], device=dev)
    q = torch.from_numpy(synth.unit_rows(rs, full.n_seg, d)).to(dev)
    kw = dict(rank_by="t2t", use_t2i_thr=True, thr=0.0, t2i_thr=0.0)
    w_idx, w_val, w_cnt, _, _ = full.sample(q, k, **kw)
    full.save(str(tmp_path / "ix"))
    idxs, vals = [], []
    for rank in range(world):
        lo, hi = shard.shard_bounds(n, world, rank)
        loc = SegmentedIndex.load(str(tmp_path / "ix"), device=dev, rows=(lo, hi))
        assert np.array_equal(loc.filepath, full.filepath[lo:hi])  # numpy path arrays (mmap-able filepath.npy)
        i, v, _, _, _ = loc.sample(q, k, **kw)
        idxs.append(torch.where(i >= 0, i + lo, i))
        vals.append(v)
    m_idx, m_val, m_cnt = shard.merge_topk(idxs, vals, k)
    assert torch.equal(m_cnt, w_cnt)
    assert torch.equal(m_idx, w_idx)
    valid = w_idx >= 0
    assert torch.equal(m_val[valid], w_val[valid])


@pytest.mark.parametrize("ns,rank", [(16384, 41), (16384, 1), (16384, 2048), (5000, 17), (1024, 1024), (40000, 33), (777, 5)])
def test_rank_select_is_the_rank_th_largest_of_every_row(ns, rank):
    """swat_rank_select (the dense top-k's pruning thresholds: the score at a small rank of each query's sample scores) against a full
    sort: exact for every shape — rows in registers (<= 16 384 scores) through the maxima / survivors path, ranks beyond 1 024 and
    longer rows through the plain bisection, ties across the rank, negative scores and zeros of both signs; the rider's max |q| is an
    upper bound within 2^-19 of torch's."""
    from swat_amd import _lib
    L = _lib.lib()
    dev = _dev()
    g = torch.Generator().manual_seed(ns + rank)
    nq, d = 37, 512
    S = (torch.randn(nq, ns, generator=g) * 0.07)
    S[1] = torch.round(S[1] * 50) / 50          # many exact ties (also across the rank)
    S[2, ::3] = 0.0
    S[2, 1::3] = -0.0
    S[3] = -S[3].abs()                          # all negative
    S[4] = S[4, 0]                              # one value only
    S = S.to(dev).contiguous()
    q = torch.randn(nq, d, generator=g).to(dev)
    out = torch.empty(nq, device=dev)
    ws = torch.zeros(nq + 1, device=dev)
    qn = torch.empty(1, device=dev)
    for rider in (True, False, True):           # (twice with the rider: its arrival counter resets itself)
        out.fill_(7.0)
        _lib.check(L.swat_rank_select(_lib.ptr(S), nq, ns, rank, _lib.ptr(out), _lib.ptr(q) if rider else None, d,
                                      _lib.ptr(ws) if rider else None, _lib.ptr(qn) if rider else None, _lib.stream()), "swat_rank_select")
        want = torch.sort(S, dim=1, descending=True).values[:, rank - 1]
        assert torch.equal(out == 0, want == 0) and torch.equal(torch.where(out == 0, torch.zeros_like(out), out),
                                                                torch.where(want == 0, torch.zeros_like(want), want))
        if rider:
            ref = q.norm(dim=1).max().item()
            assert ref <= qn.item() <= ref * (1 + 2.0 ** -19)
    assert ws[0].item() == 0.0


def test_shard_merge_of_more_than_8192_gathered_entries_per_query():
    """8 ranks x k = 2 048 gathered entries per query are beyond the library merge's 8 192 keys: merge_topk takes its stable
    descending sort there — the same lists as the library merge gives on the first 8 192 and as a host stable sort gives on all."""
    from swat_amd.retrieval import shard
    dev = _dev()
    rs = np.random.RandomState(9)
    nq, world, k = 5, 8, 2048
    idxs, vals = [], []
    for r in range(world):
        v = np.sort(rs.randint(0, 400, size=(nq, k)).astype(np.float32) / 400.0, axis=1)[:, ::-1].copy()   # many exact ties
        i = np.sort(rs.choice(100000, size=(nq, k), replace=False), axis=1) + r * 100000
        i[:, -17:] = -1
        v[:, -17:] = -np.inf
        idxs.append(torch.from_numpy(i).to(dev))
        vals.append(torch.from_numpy(v).to(dev))
    m_idx, m_val, m_cnt = shard.merge_topk(idxs, vals, k)
    c_idx, c_val, c_cnt = shard.merge_topk([i.cpu() for i in idxs], [v.cpu() for v in vals], k)
    assert torch.equal(m_idx.cpu(), c_idx) and torch.equal(m_val.cpu(), c_val) and torch.equal(m_cnt.cpu(), c_cnt)
    l_idx, l_val, l_cnt = shard.merge_topk(idxs[:4], vals[:4], k)       # 8 192 entries: the library merge
    h_idx, h_val, h_cnt = shard.merge_topk([i.cpu() for i in idxs[:4]], [v.cpu() for v in vals[:4]], k)
    assert torch.equal(l_idx.cpu(), h_idx) and torch.equal(l_cnt.cpu(), h_cnt)


def _write_mined(root, case):
    os.makedirs(f"{root}/synthetic", exist_ok=True)
    torch.save({"image_features": torch.from_numpy(case["img"]), "caption_features": torch.from_numpy(case["cap"]),
                "labels": torch.from_numpy(case["labels"]), "filepath": case["paths"]},
               f"{root}/synthetic/synthetic_vitb32_openclip_laion400m_mined.pth")


@pytest.mark.parametrize("kind", ["float", "grid"])
@pytest.mark.parametrize("method,tag", [("T2T-rank-T2I-tshd", "t2t_t2i"), ("T2T-rank", "t2t"), ("T2I-rank", "t2i")])
@pytest.mark.parametrize("dedup", [False, True])
def test_sample_retrieval_command_line_writes_the_reference_files(golden_dir, tmp_path, monkeypatch, kind, method, tag, dedup):
    """`python -m swat_amd.retrieval.sample_retrieval --sampling_method ... [--image_dedup]` (ref __main__ :1673-1748) on a
    mined.pth FILE: <prefix>.txt and <prefix>_num_imgs_sampled.json are byte-for-byte what the reference's own sampling()
    wrote for the same file (tests/golden/retrieval_cli.npz), and the split file is copied next to the dataset lists."""
    from swat_amd.retrieval import sample_retrieval as cli
    g = np.load(os.path.join(golden_dir, "retrieval_cli.npz"), allow_pickle=False)
    spec = synth.RETRIEVAL_CASES[kind]
    case = synth.retrieval_case(**spec)
    root, data = tmp_path / "retrieved", tmp_path / "data"
    _write_mined(root, case)
    os.makedirs(data / "synthetic" / "prompts")
    json.dump({str(c): {"name": f"class {c}"} for c in case["class_ids"]}, open(data / "synthetic" / "synthetic_metrics-LAION400M.json", "w"))
    # cached prompt tensors (what cal_prompt_tensors would have computed with the text tower): the CLI loads them
    pt = {str(c): {"mean": torch.from_numpy(case["queries"][i])} for i, c in enumerate(case["class_ids"])}
    base = data / "synthetic" / "prompts" / "synthetic_vitb32_openclip_laion400m_alternates"
    torch.save(pt, f"{base}_prompt_tensors.pth")
    torch.save({}, f"{base}_text_prompts.pth")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr("swat_amd.engine.get_engine", lambda **kw: (torch.nn.Linear(1, 1), None, None))  # no tower needed here
    argv = ["--prefix", "golden", "--dataset", "synthetic", "--root", str(root), "--sampling_method", method, "--num_samples",
            str(spec["k"]), "--data_dir", str(data), "--log_mode", "file"] + (["--image_dedup"] if dedup else [])
    path, ct = cli.main(argv)
    key = f"{kind}_{tag}_{'dedup' if dedup else 'plain'}"
    assert ct == int(g[key + "_ct"])
    assert open(path).read() == str(g[key + "_txt"])
    out = os.path.dirname(path)
    assert open(f"{out}/golden_num_imgs_sampled.json").read() == str(g[key + "_counts"])
    assert open(data / "synthetic" / "golden.txt").read() == str(g[key + "_txt"])       # copied to data/<dataset>/
    assert os.path.basename(out) == "synthetic_vitb32_openclip_laion400m_golden"


@pytest.mark.parametrize("kind", ["float", "grid"])
def test_zeroshot_clip_img_filter_matches_reference(golden_dir, kind):
    """ref :278-329 incl. its matching rule (positions of mispredicted rows against the files' numeric names)."""
    from swat_amd.models import MyLinear
    from swat_amd.retrieval import samplers
    from swat_amd.retrieval.sample_retrieval import zeroshot_clip_img_filter
    g = np.load(os.path.join(golden_dir, "retrieval_cli.npz"), allow_pickle=False)
    case = synth.retrieval_case(**synth.RETRIEVAL_CASES[kind])
    index = samplers.transform_extracted_fea({"image_features": torch.from_numpy(case["img"]), "caption_features": torch.from_numpy(case["cap"]),
                                              "labels": torch.from_numpy(case["labels"]), "filepath": case["paths"]})
    C = max(case["class_ids"]) + 1
    W = torch.zeros(C, case["queries"].shape[1])
    for i, c in enumerate(case["class_ids"]):
        W[c] = torch.from_numpy(case["queries"][i])
    got = zeroshot_clip_img_filter(None, None, "/unused", pre_extracted_feats=index, head=MyLinear(weights=W).to("cuda"))
    want = json.loads(str(g[f"{kind}_zsfilter"]))
    # grid: exact arithmetic; float: the canonical k-ordered fp32 chain against the reference's BLAS order — the two class
    # logits of a row would have to agree to ~1e-7 for the argmax to differ, which no row of the fixture does
    assert {c: sorted(v) for c, v in got.items()} == want


def test_excluded_mask_is_vectorised_and_equals_the_row_loop():
    from collections import defaultdict
    from swat_amd.retrieval import samplers
    case = synth.retrieval_case(**synth.RETRIEVAL_CASES["float"])
    index = samplers.transform_extracted_fea({"image_features": torch.from_numpy(case["img"]), "caption_features": torch.from_numpy(case["cap"]),
                                              "labels": torch.from_numpy(case["labels"]), "filepath": case["paths"]})
    dup, filt = defaultdict(set), defaultdict(set)
    for p, e, f in zip(case["paths"], case["excl_dup"], case["excl_filt"]):
        if e:
            dup[p.split("/")[-2]].add(p)
        if f:
            filt[p.split("/")[-2]].add(p)
    mask = samplers._excluded_mask(index, dup, filt).cpu().numpy()
    want = (case["excl_dup"] | case["excl_filt"])[index.order.cpu().numpy()]
    assert np.array_equal(mask.astype(bool), want)
    assert index.paths_array().dtype.kind == "U" and index.file_ids().dtype == np.int64
    assert samplers._excluded_mask(index, {}, {}) is None


def test_extract_mined_features_matches_oracle_and_round_trips_mined_pth(tmp_path):
    """B1 / B2: MinedDataset (folder = label, file name = image id, caption map) -> two passes (captions, images) on the HIP
    towers -> the reference's mined.pth dict in loader order; against the fp32 oracle encoding the same files."""
    import pickle
    import types
    from PIL import Image
    from oracle import vit_oracle as vo
    from swat_amd.clip_model import CLIP
    from swat_amd.dataloader import MinedDataset
    from swat_amd.retrieval import extract_mined_feature as ex, samplers
    from swat_amd.transforms import build_transform
    rs = np.random.RandomState(3)
    root = tmp_path / "retrieved" / "toy" / "toy_retrieved_LAION400M-all_synonyms-random"
    cap_map = {}
    words = "red bird on a branch small yellow finch blue jay near water photo of".split()
    for c in (0, 3, 7):
        os.makedirs(root / str(c))
        cap_map[str(c)] = {}
        for i in range(5):
            img_id = str(100 * c + 7 * i + 1)
            Image.fromarray(rs.randint(0, 255, (48 + 8 * i, 64 + 4 * c, 3), dtype=np.uint8)).save(root / str(c) / f"{img_id}.jpg", quality=95)
            cap_map[str(c)][img_id] = " ".join(rs.choice(words, size=3 + i % 4))
    cap_path = tmp_path / "toy.map"
    pickle.dump(cap_map, open(cap_path, "wb"))
    vocab = {w: 2 + i for i, w in enumerate(words)}

    def tokenizer(texts, ctx=16):  # SOT = 98, EOT = 99 (highest id), zero padding
        out = torch.zeros(len(texts), ctx, dtype=torch.long)
        for r, t in enumerate(texts):
            ids = [98] + [vocab[w] for w in t.split()] + [99]
            out[r, :len(ids)] = torch.tensor(ids)
        return out

    vis = (64, 16, 128, 2, 2, 64, "quick_gelu")
    torch.manual_seed(5)
    ref = vo.OracleCLIP(vision_cfg=vis, text_width=64, text_heads=1, text_layers=2, ctx=16, vocab=100)
    model = CLIP(vision_cfg=vis, text_cfg=(16, 100, 64, 1, 2))
    model.load_state_dict(ref.state_dict())
    model.to("cuda")
    preprocess = build_transform(64, "val")
    args = types.SimpleNamespace(root=str(tmp_path / "retrieved"), bsz=4, num_workers=0)
    from swat_amd import jpeg
    n_dev0 = jpeg.stats["device"]
    got = ex.pre_extract_mined_fea(args, "toy", str(root), str(cap_path), "tinycfg", model, preprocess, tokenizer)
    # the default input pipeline: the workers read files, the 15 baseline JPEGs are decoded and transformed on host threads + the GPU —
    # and the dict is the one the reference-shaped pipeline (Pillow decode + host transform in the workers) produces, bit for bit
    assert jpeg.stats["device"] - n_dev0 == 15
    args.host_transform = True
    host = ex.pre_extract_mined_fea(args, "toy", str(root), str(cap_path), "tinycfg", model, preprocess, tokenizer)
    assert jpeg.stats["device"] - n_dev0 == 15
    assert host["filepath"] == got["filepath"] and all(torch.equal(host[k], got[k]) for k in ("caption_features", "labels", "image_features"))
    args.host_transform = False
    fn = tmp_path / "retrieved" / "toy" / "toy_tinycfg_mined.pth"
    disk = torch.load(fn, weights_only=False)
    assert set(disk) == {"caption_features", "labels", "filepath", "image_features"}
    ds = MinedDataset(dataset_root=str(root), transform=preprocess, caption_map=cap_map)
    assert disk["filepath"] == [str(f) for f in ds.fnames] and len(disk["filepath"]) == 15      # loader order
    assert disk["labels"].tolist() == [int(f.parent.name) for f in ds.fnames]
    with torch.no_grad():
        items = [ds[i] for i in range(len(ds))]
        wi = ref.encode_image(torch.stack([it[0] for it in items]))
        wt = ref.encode_text(tokenizer([it[3] for it in items]))
    wi, wt = wi / wi.norm(dim=-1, keepdim=True), wt / wt.norm(dim=-1, keepdim=True)
    assert (disk["image_features"] - wi).abs().max() <= 0.01 and (disk["caption_features"] - wt).abs().max() <= 0.01
    assert torch.allclose(disk["image_features"].norm(dim=-1), torch.ones(15), atol=1e-5)
    assert torch.equal(got["image_features"], disk["image_features"])
    # a rank-strided extraction merges to the same dict
    parts = []
    for r in range(2):
        d = MinedDataset(dataset_root=str(root), transform=preprocess, caption_map=cap_map, rank=r, world=2)
        ld = torch.utils.data.DataLoader(d, batch_size=4, shuffle=False)
        parts.append(ex.extract_mined_feats_batch(model, ld, tokenizer))
    merged = ex.merge_rank_dicts(parts)
    assert merged["filepath"] == disk["filepath"] and torch.equal(merged["labels"], disk["labels"])
    assert (merged["image_features"] - disk["image_features"]).abs().max() <= 1e-6  # batch composition differs, rows do not
    # and the samplers read the file's dict
    index = samplers.transform_extracted_fea(disk)
    assert index.n_rows == 15 and index.class_ids.tolist() == [0, 3, 7]
