"""CPU, world_size 2, gloo: the host logic of the command line's data-parallel mode (swat_amd/dp.py) — RNG re-sync,
global-batch sharding with CutMix partners, the rank-strided sampler, rank-0 broadcasts and the per-epoch loss average."""
import os
import random
import socket
import sys
import types

import pytest

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


def _paste(images, perm, box):
    """ref utils/training.py:1466 — inputs[:, :, x1:x2, y1:y2] = inputs[perm, :, x1:x2, y1:y2] through a gathered copy."""
    x1, y1, x2, y2 = box
    out = images.clone()
    out[:, :, x1:x2, y1:y2] = images[perm][:, :, x1:x2, y1:y2]
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        from swat_amd import dp
        from swat_amd.training import cutmix_draw
        args = types.SimpleNamespace(dist_backend="gloo", cutmix_beta=1.0, mix_prob=0.5)
        dp.init_from_env(args)
        assert (args.rank, args.world) == (rank, world) and dp.is_main(args) == (rank == 0)
        # ranks start from different RNG states (rank 0 alone ran RNG-consuming code); sync_rng puts them in step again
        torch.manual_seed(100 + rank); np.random.seed(200 + rank); random.seed(300 + rank)
        dp.sync_rng(args)
        draws = (torch.rand(3).tolist(), np.random.rand(3).tolist(), random.random())
        assert dp.bcast_object(args, draws) == draws, "RNG streams differ after sync_rng"
        # global batch: identical draws on every rank, rows split, partners carried along -> the union equals the
        # single-process CutMix batch
        B, b = 8, 4
        g = torch.Generator().manual_seed(5)
        images, labels = torch.randn(B, 3, 16, 16, generator=g), torch.randint(0, 10, (B,), generator=g)
        seen_mix = False
        for it in range(6):
            lam, perm, box = cutmix_draw(args, images.size())
            loc, ya, yb, perm_l, nb = dp.shard_batch(images, labels, perm, rank, world)
            assert nb == b and torch.equal(ya, labels[rank * b:(rank + 1) * b])
            if perm is None:
                assert perm_l is None and loc.shape[0] == b and torch.equal(yb, ya)
                mine = loc
            else:
                seen_mix = True
                assert loc.shape[0] == 2 * b and torch.equal(yb, labels[perm][rank * b:(rank + 1) * b])
                mine = _paste(loc, torch.cat([perm_l, torch.arange(b, 2 * b)]), box)[:b]  # what the fused paste computes
            whole = _paste(images, perm, box) if perm is not None else images
            assert torch.equal(mine, whole[rank * b:(rank + 1) * b])
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            assert torch.equal(torch.cat(gathered), whole)
        assert seen_mix
        # --dp_weak: the shuffled epoch is a partition over ranks, drawn from RNG streams that stay in step
        dp.sync_rng(args)
        smp = dp.RankStridedSampler(21, rank, world)
        idx = list(iter(smp))
        assert len(idx) == len(smp) == 10
        allidx = [None] * world
        dist.all_gather_object(allidx, idx)
        assert len(set(allidx[0]) | set(allidx[1])) == 20 and not set(allidx[0]) & set(allidx[1])
        nxt = torch.rand(1).item()
        assert dp.bcast_object(args, nxt) == nxt  # the sampler consumed the same amount of the stream on every rank
        # one scalar all-reduce per epoch for loss.csv
        assert abs(dp.mean_over_ranks(args, 1.0 + rank) - 1.5) < 1e-12
        # SURVEY 8(e) row 3, mined-feature extraction: rank-strided shares of 7 files -> rank 0 holds the single-process dict
        # (feature matrices gathered as padded tensors, paths as objects)
        from swat_amd.retrieval.extract_mined_feature import gather_rank_parts
        n = 7
        feats = torch.arange(n * 4, dtype=torch.float32).view(n, 4)
        mine = list(range(rank, n, world))
        part = {"caption_features": feats[mine], "image_features": -feats[mine], "labels": torch.tensor(mine),
                "filepath": [f"f{i}.jpg" for i in mine]}
        merged = gather_rank_parts(part, n, rank, world)
        if rank == 0:
            assert merged["filepath"] == [f"f{i}.jpg" for i in range(n)] and torch.equal(merged["labels"], torch.arange(n))
            assert torch.equal(merged["caption_features"], feats) and torch.equal(merged["image_features"], -feats)
        else:
            assert merged is None
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_world2_gloo_data_parallel_host_logic():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    assert all(r[1] == "ok" for r in res), res


def test_extraction_does_not_split_the_batch_over_ranks(monkeypatch):
    """`--bsz 1024` with 3 ranks is fine for feature extraction (nothing is split); training still refuses."""
    from swat_amd import dp
    import pytest
    monkeypatch.setenv("WORLD_SIZE", "3")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(dist, "init_process_group", lambda *a, **k: None)
    with pytest.raises(ValueError):
        dp.init_from_env(types.SimpleNamespace(dist_backend="gloo", bsz=1024))
    a = dp.init_from_env(types.SimpleNamespace(dist_backend="gloo", bsz=1024), split_batch=False)
    assert a.world == 3


def test_gradient_wire_follows_the_data_parallel_mode(monkeypatch):
    """--dp_wire auto: the mode that splits --bsz over the ranks reproduces the one-GPU trajectory and keeps fp32 gradients on the links;
    --dp_weak (throughput) sends bf16; the flag and an exported SWAT_DP_WIRE override it (swat_amd/dp.py, ddp.GradWire)."""
    from swat_amd import dp
    from swat_amd.parser import parse_args
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(dist, "init_process_group", lambda *a, **k: None)
    for kw, want in ((dict(), "fp32"), (dict(dp_weak=True), "bf16"), (dict(dp_weak=True, dp_wire="fp32"), "fp32"), (dict(dp_wire="bf16"), "bf16")):
        monkeypatch.delenv("SWAT_DP_WIRE", raising=False)
        dp.init_from_env(types.SimpleNamespace(dist_backend="gloo", bsz=32, dp_wire=kw.get("dp_wire", "auto"), dp_weak=kw.get("dp_weak", False)))
        assert os.environ["SWAT_DP_WIRE"] == want, (kw, os.environ["SWAT_DP_WIRE"])
    monkeypatch.setenv("SWAT_DP_WIRE", "fp32")
    dp.init_from_env(types.SimpleNamespace(dist_backend="gloo", bsz=32, dp_wire="auto", dp_weak=True))
    assert os.environ["SWAT_DP_WIRE"] == "fp32"   # an exported value wins over `auto` ...
    monkeypatch.setenv("SWAT_DP_WIRE", "bf16")
    with pytest.warns(UserWarning, match="overrides the exported"):
        dp.init_from_env(types.SimpleNamespace(dist_backend="gloo", bsz=32, dp_wire="fp32", dp_weak=True))
    assert os.environ["SWAT_DP_WIRE"] == "fp32"   # ... but not over an explicit --dp_wire
    monkeypatch.delenv("SWAT_DP_WIRE", raising=False)
    import sys
    monkeypatch.setattr(sys, "argv", ["main.py"])
    assert parse_args().dp_wire == "auto"


def test_single_process_is_untouched():
    from swat_amd import dp
    args = types.SimpleNamespace()
    os.environ.pop("WORLD_SIZE", None)
    dp.init_from_env(args)
    assert (args.rank, args.world) == (0, 1) and dp.is_main(args)
    assert dp.bcast_object(args, {"a": 1}) == {"a": 1} and dp.mean_over_ranks(args, 2.5) == 2.5
    dp.sync_rng(args)
    x = torch.randn(4, 3, 8, 8)
    y = torch.arange(4)
    loc, ya, yb, pl, b = dp.shard_batch(x, y, None, 0, 1)
    assert b == 4 and torch.equal(loc, x) and pl is None


def test_rank_batch_sampler_yields_the_single_process_batches_rank_by_rank():
    """dp.RankBatchSampler (global-batch data parallelism without W-fold loading): with the ranks' RNG states equal, the rows each rank
    loads are its slice of the batches a single process gets from DataLoader(shuffle=True, drop_last=True), its partner rows are the
    single process's CutMix partners of those rows, and the queued (lambda, permutation, box) draws are the single process's draws —
    over two epochs, CutMix on (mix_prob 0.5: iterations with and without a paste) and off."""
    import types
    import numpy as np
    from torch.utils.data import DataLoader, Dataset
    from swat_amd import dp, training

    class Ids(Dataset):
        def __len__(self):
            return 53

        def __getitem__(self, i):
            return i

    args = types.SimpleNamespace(cutmix_beta=1.0, mix_prob=0.5)
    n, bsz, world = 53, 12, 3

    def seed():
        torch.manual_seed(11)
        np.random.seed(11)

    for use_cutmix in (True, False):
        seed()
        single = []          # (batch indices, lam, perm, box) per iteration of the 1-process loop
        for epoch in range(2):
            for batch in DataLoader(Ids(), batch_size=bsz, shuffle=True, drop_last=True):
                lam, perm, box = training.cutmix_draw(args, (bsz, 3, 224, 224)) if use_cutmix else (1.0, None, None)
                single.append((batch.tolist(), lam, perm, box))
        assert len(single) == 2 * (n // bsz)
        for rank in range(world):
            seed()
            sampler = dp.RankBatchSampler(n, bsz, rank, world)
            sampler.configure((lambda size: training.cutmix_draw(args, size)) if use_cutmix else None)
            loader = DataLoader(Ids(), batch_sampler=sampler)
            assert len(loader) == n // bsz
            it = 0
            for epoch in range(2):
                for rows in loader:
                    lam, perm, box = sampler.pop_draw()
                    b_idx, s_lam, s_perm, s_box = single[it]
                    b = bsz // world
                    assert lam == s_lam and box == s_box and (perm is None) == (s_perm is None)
                    own = b_idx[rank * b:(rank + 1) * b]
                    if perm is None:
                        assert rows.tolist() == own
                    else:
                        assert torch.equal(perm, s_perm)
                        assert rows.tolist() == own + [b_idx[j] for j in s_perm[rank * b:(rank + 1) * b].tolist()]
                    it += 1
            assert it == len(single) and sampler.draws == []


def test_rank_zero_only_validation_between_epochs_keeps_the_ranks_in_step():
    """with --early_stop rank 0 alone validates after every epoch, and iterating ANY DataLoader advances torch's global
    generator (the iterator's base seed).  swat_amd/training.py therefore re-synchronises the generator states at the top of EVERY
    epoch (dp.sync_rng: ranks > 0 continue from rank 0's states).  Emulated in one process: rank 0 runs first and its states at each
    epoch's start are what the other ranks restore; without that restore their second-epoch permutations differ (asserted too)."""
    import types
    import numpy as np
    from torch.utils.data import DataLoader, Dataset
    from swat_amd import dp, training

    class Ids(Dataset):
        def __len__(self):
            return 41

        def __getitem__(self, i):
            return i

    args = types.SimpleNamespace(cutmix_beta=1.0, mix_prob=0.5)
    n, bsz, world, epochs = 41, 8, 2, 3

    def seed():
        torch.manual_seed(5)
        np.random.seed(5)

    def validate_like():
        for _ in DataLoader(Ids(), batch_size=16, shuffle=False):   # (draws the iterator's base seed from the global generator)
            pass

    seed()
    single = []
    for epoch in range(epochs):
        for batch in DataLoader(Ids(), batch_size=bsz, shuffle=True, drop_last=True):
            single.append((batch.tolist(),) + tuple(training.cutmix_draw(args, (bsz, 3, 224, 224))))
        validate_like()

    def run_rank(rank, states_in, resync):
        seed()
        sampler = dp.RankBatchSampler(n, bsz, rank, world)
        sampler.configure(lambda size: training.cutmix_draw(args, size))
        loader = DataLoader(Ids(), batch_sampler=sampler)
        got, states_out = [], []
        for epoch in range(epochs):
            if rank == 0:
                states_out.append((torch.get_rng_state(), np.random.get_state()))
            elif resync:   # dp.sync_rng at the top of the epoch
                torch.set_rng_state(states_in[epoch][0])
                np.random.set_state(states_in[epoch][1])
            for rows in loader:
                got.append((rows.tolist(),) + tuple(sampler.pop_draw()))
            if rank == 0:
                validate_like()   # rank 0 only (dp.is_main)
        return got, states_out

    r0, states = run_rank(0, None, True)
    r1, _ = run_rank(1, states, True)
    r1_stale, _ = run_rank(1, states, False)
    b = bsz // world
    for it, (b_idx, lam, perm, box) in enumerate(single):
        for rank, got in ((0, r0), (1, r1)):
            rows, g_lam, g_perm, g_box = got[it]
            assert g_lam == lam and g_box == box and (g_perm is None) == (perm is None)
            own = b_idx[rank * b:(rank + 1) * b]
            assert rows == (own if perm is None else own + [b_idx[j] for j in perm[rank * b:(rank + 1) * b].tolist()]), (it, rank)
    per_epoch = n // bsz
    assert [g[0] for g in r1_stale[:per_epoch]] == [g[0] for g in r1[:per_epoch]]       # the first epoch needs no re-sync ...
    assert [g[0] for g in r1_stale[per_epoch:]] != [g[0] for g in r1[per_epoch:]]       # ... the later ones do
