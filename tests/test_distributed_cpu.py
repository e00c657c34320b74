"""CPU, world_size 2, gloo: the multi-GPU logic that does not need a kernel — sharded retrieval merge and the
per-stage gradient-slice reduction (the N > 1 path of bench.py / FusedTrainStep)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import retrieval_oracle as orc, synth
        from swat_amd.retrieval import shard
        from swat_amd.ddp import SliceReducer

        # ---- sharded retrieval: local top-k per shard (oracle stands in for the kernel) + gather + merge ----
        rs = np.random.RandomState(5)
        n, d, k = 4001, 64, 37
        feats = synth.unit_rows(rs, n, d)
        for _ in range(200):  # ties across and inside shards
            a, b = rs.randint(n), rs.randint(n)
            feats[a] = feats[b]
        queries = synth.unit_rows(rs, 3, d)
        offsets = np.array([0, 1500, 1500, n], np.int64)  # one empty class
        lo, hi = shard.shard_bounds(n, world, rank)
        loc_off = shard.local_segments(torch.from_numpy(offsets), lo, hi).numpy()
        idx, cnt, sims, _ = orc.sample_index(feats[lo:hi], None, queries, loc_off, k, thr=-2.0, return_sims=True)
        val = np.full(idx.shape, -np.inf, np.float32)
        for s in range(3):
            val[s, :cnt[s]] = sims[idx[s, :cnt[s]]]
        gidx = np.where(idx >= 0, idx + lo, -1)
        m_idx, m_val, m_cnt = shard.gather_and_merge(torch.from_numpy(gidx), torch.from_numpy(val), k)
        w_idx, w_cnt = orc.sample_index(feats, None, queries, offsets, k, thr=-2.0)
        assert np.array_equal(m_idx.numpy(), w_idx), "sharded merge != single-shard top-k"
        assert np.array_equal(m_cnt.numpy(), w_cnt)

        # ---- gradient slices: every rank ends with the sum; untouched gaps stay untouched ----
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        red = SliceReducer()
        for lo_, hi_ in [(0, 64), (128, 512), (960, 1000)]:
            red.reduce(flat, lo_, hi_)
        red.wait()
        want = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        tot = sum(r + 1 for r in range(world))
        for lo_, hi_ in [(0, 64), (128, 512), (960, 1000)]:
            want[lo_:hi_] = torch.arange(lo_, hi_, dtype=torch.float32) * tot
        assert torch.equal(flat, want)

        # ---- how a gradient range crosses the links (ddp.GradWire): all_reduce | reduce_scatter + all_gather, fp32 | bf16 ----
        from swat_amd.ddp import GradWire
        g0 = torch.Generator().manual_seed(100 + rank)
        base = torch.randn(4096, generator=g0) * 1e-3
        both = [torch.randn(4096, generator=torch.Generator().manual_seed(100 + r)) * 1e-3 for r in range(world)]
        exact = sum(both)
        for wire in ("fp32", "bf16"):
            for coll in ("all_reduce", "rs_ag"):
                gw = GradWire(4096, torch.device("cpu"), wire=wire, collective=coll)
                buf = base.clone()
                gw.reduce(buf, 64, 4032)                      # a 64-aligned range: divisible by the world size -> rs_ag really runs
                assert torch.equal(buf[:64], base[:64]) and torch.equal(buf[4032:], base[4032:])
                got, want_r = buf[64:4032], exact[64:4032]
                if wire == "fp32":
                    assert torch.equal(got, want_r), (wire, coll)
                else:
                    # bf16 wire: every addend rounded to 8 significant bits (unit roundoff 2^-8) and every partial sum
                    # rounded again: |err| <= (1 + log2(world)) * 2^-8 * sum_r |g_r| per element — the bound the
                    # data-parallel step's bf16 option is held to (world 2: 2^-7)
                    bound = 2.0 ** -7 * sum(b.abs() for b in both)[64:4032] + 1e-12
                    assert bool(((got - want_r).abs() <= bound).all()), (wire, coll)
                    assert not torch.equal(got, want_r)       # it did go through bf16
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_shard_merge_and_grad_slices():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_shard_bounds_cover_everything():
    from swat_amd.retrieval.shard import shard_bounds
    for n in (0, 1, 7, 8, 10_000_001):
        for w in (1, 2, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
