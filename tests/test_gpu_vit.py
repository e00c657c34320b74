"""GPU: the HIP vision tower / fused step (bf16 GEMMs, fp32 accumulate + fp32 residual stream) against the fp32 CPU oracle.

Stated tolerance (BASELINE north_star "logits/loss within a stated fp tolerance"): embeddings within 0.7% of their
max-abs, per-tensor gradient relative L2 error < 1.1%, loss within 2e-2 absolute — bf16 operand rounding (2^-9 relative
per element) through 12-24 blocks; an fp32 run of the same graph on the CPU is the reference.  The two bounds are 1.5 x
the worst case measured over TINY / TINY_GELU / ViT-B/32 / ViT-B/16 / ViT-L/14 and two seeds (tools/grad_error.py on
MI355X: embeddings 0.15-0.41 %, gradients 0.63-0.68 %, the worst tensor is a LayerNorm weight or conv1)."""
EMB_TOL, GRAD_TOL = 0.007, 0.011
import copy
import math
import types

import numpy as np
import pytest
import torch

from oracle import vit_oracle as vo

pytestmark = pytest.mark.gpu

TINY = (64, 16, 128, 2, 2, 64, "quick_gelu")  # image, patch, width, layers, heads, embed, act
TINY_GELU = (56, 14, 128, 2, 2, 64, "gelu")   # patch 14: 3*14*14 = 588 columns (not a multiple of 8) like ViT-L/14
B32 = (224, 32, 768, 12, 12, 512, "quick_gelu")
B16 = (224, 16, 768, 12, 12, 512, "gelu")      # BASELINE config 3: T = 197 (ragged against the 64-wide tiles)
L14 = (224, 14, 1024, 24, 16, 768, "gelu")     # BASELINE config 5: T = 257, 24 blocks, patch columns 588


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def make_pair(cfg, seed=0):
    from swat_amd.clip_model import VisionTransformer
    torch.manual_seed(seed)
    ref = vo.OracleVisionTransformer(*cfg)
    with torch.no_grad():  # make biases / LN params non-trivial
        for n, p in ref.named_parameters():
            if n.endswith("bias"):
                p.add_(0.02 * torch.randn_like(p))
            if "ln_" in n and n.endswith("weight"):
                p.add_(0.05 * torch.randn_like(p))
    mine = VisionTransformer(*cfg)
    mine.load_state_dict(ref.state_dict())
    return ref, mine.to(dev())


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.mark.parametrize("cfg,B", [(TINY, 5), (TINY_GELU, 3), (B32, 4), (B32, 32), (B16, 2), (L14, 2)])
def test_forward_backward_match_oracle(cfg, B):
    """(B32, 32) is the benchmark's shape: the batch fills two rounds of 16 two-image panels, so the default mode runs all twelve
    blocks of the forward as ONE persistent launch (csrc/vit_mega.hip) — held here to the same tolerances against the fp32 oracle
    as the op-by-op path (ref call site utils/training.py:1474)."""
    from swat_amd import _lib
    ref, mine = make_pair(cfg)
    launches0 = _lib.lib().swat_vit_mega_launches()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, cfg[0], cfg[0], generator=g)
    w = torch.randn(B, cfg[5], generator=g)
    f_ref = ref(x)
    (f_ref * w).sum().backward()
    f = mine.forward_raw(x.to(dev()), save=True)
    persistent = _lib.lib().swat_vit_mega_launches() - launches0
    assert persistent == (1 if (cfg, B) == (B32, 32) else 0), "which forward ran is part of what this case pins"
    assert (f.cpu() - f_ref.detach()).abs().max().item() <= EMB_TOL * f_ref.abs().max().item()
    bwd0 = _lib.lib().swat_vit_mega_backward_launches()
    mine.backward_raw(w.to(dev()).contiguous())
    torch.cuda.synchronize()
    assert _lib.lib().swat_vit_mega_poll() == 0
    # the data gradients of the twelve blocks of the benchmark's shape come from vit_bwd_mega_kernel (the slowest kernel of the step):
    # this is its one direct comparison with the oracle, so which backward ran is pinned as well
    assert _lib.lib().swat_vit_mega_backward_launches() - bwd0 == (1 if (cfg, B) == (B32, 32) else 0)
    arena = mine.arena()
    ref_params = dict(ref.named_parameters())
    worst = 0.0
    for n, p in mine.named_parameters():
        o = arena.offset_of(p)
        got = arena.grad[o:o + p.numel()].view(p.shape)
        r = rel(got, ref_params[n].grad)
        worst = max(worst, r)
        assert r < GRAD_TOL, f"{n}: relative gradient error {r}"
    # inference path (no saved activations) gives the same embedding
    f2 = mine.forward_raw(x.to(dev()), save=False)
    assert torch.equal(f, f2)


def test_autograd_node_state_dict_deepcopy():
    ref, mine = make_pair(TINY)
    x = torch.randn(3, 3, 64, 64)
    f_ref = ref(x)
    f_ref.square().sum().backward()
    f = mine(x.to(dev()))
    f.square().sum().backward()
    ref_params = dict(ref.named_parameters())
    for n, p in mine.named_parameters():
        assert p.grad is not None and rel(p.grad, ref_params[n].grad) < GRAD_TOL, n
    # open_clip key names round-trip and deepcopy keeps working (main.py:555, utils/training.py:1515)
    assert set(mine.state_dict().keys()) == set(ref.state_dict().keys())
    twin = copy.deepcopy(mine)
    with torch.no_grad():
        f_twin = twin(x.to(dev()))
    assert torch.equal(f_twin, f.detach())
    with torch.no_grad():  # mutating the copy must not touch the original (separate arenas)
        for p in twin.parameters():
            p.mul_(0.5)
        assert torch.equal(mine(x.to(dev())), f.detach())
        assert not torch.equal(twin(x.to(dev())), f.detach())


def test_cutmix_fused_into_patch_embed_matches_paste():
    ref, mine = make_pair(TINY)
    x = torch.randn(6, 3, 64, 64)
    perm = torch.tensor([2, 0, 5, 1, 3, 4])
    box = (5, 11, 40, 61)
    f_ref = ref(vo.cutmix_paste(x, perm, box)).detach()
    f = mine.forward_raw(x.to(dev()), perm=perm.to(dev()), box=box)
    assert (f.cpu() - f_ref).abs().max().item() <= EMB_TOL * f_ref.abs().max().item()


def _args(**kw):
    a = types.SimpleNamespace(method="cutmix", device="cuda", lr_classifier=1e-4, lr_backbone=1e-6, temperature=0.07,
                              optim="AdamW", wd=1e-2, epochs=2, cutmix_beta=1.0, mix_prob=0.5, loss_name="CE")
    a.__dict__.update(kw)
    return a


class _Lg:
    handlers = []

    def info(self, *a):
        pass


@pytest.mark.parametrize("cfg,B,C", [(TINY, 8, 10), (B32, 8, 200), (B32, 32, 200)])
def test_fused_steps_track_oracle_trajectory(cfg, B, C):
    """8 optimisation steps (CutMix on/off, 3 LR groups, warm-up schedule) vs the same steps on the fp32 oracle
    with torch.optim.AdamW: per-step loss within 2e-2, final weights within 1e-3 relative.  (B32, 32, 200) is bench.py's step:
    every forward of it is the persistent launch (asserted)."""
    from swat_amd import _lib, optimizers
    launches0 = _lib.lib().swat_vit_mega_launches()
    bwd0 = _lib.lib().swat_vit_mega_backward_launches()
    from swat_amd.clip_model import CLIP
    from swat_amd.models import MyLinear
    from swat_amd.step import FusedTrainStep
    torch.manual_seed(0)
    ref_clip = vo.OracleCLIP(vision_cfg=cfg, with_text=False)
    model = CLIP(vision_cfg=cfg, text_cfg=(8, 64, 64, 1, 1))
    model.visual.load_state_dict(ref_clip.visual.state_dict())
    model.to(dev())
    W0 = torch.randn(C, cfg[5])
    W0 /= W0.norm(dim=-1, keepdim=True)
    head = MyLinear(weights=W0.clone()).to(dev())
    args = _args()
    params, logit_scale = optimizers.set_params(args, model, head, _Lg())
    loader = [None] * 100
    opt, sched, total = optimizers.set_optimizer(args, params, loader)
    step = FusedTrainStep(model, head, logit_scale, opt, world_size=1)

    r_head = torch.nn.Linear(cfg[5], C, bias=False)
    with torch.no_grad():
        r_head.weight.copy_(W0)
    r_ls = torch.nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
    groups = [{"params": r_head.parameters(), "lr": 1e-4}, {"params": ref_clip.visual.parameters(), "lr": 1e-6},
              {"params": [r_ls], "lr": 1e-4}]
    r_opt = torch.optim.AdamW(groups, betas=(0.9, 0.999), weight_decay=1e-2)
    base = [1e-4, 1e-6, 1e-4]
    rs = np.random.RandomState(3)
    g = torch.Generator().manual_seed(4)
    for it in range(8):
        x = torch.randn(B, 3, cfg[0], cfg[0], generator=g)
        y = torch.randint(0, C, (B,), generator=g)
        if it % 2 == 0:
            perm = torch.from_numpy(rs.permutation(B))
            x1, y1 = rs.randint(0, cfg[0] // 2, size=2)
            box = (int(x1), int(y1), int(x1 + cfg[0] // 3), int(y1 + cfg[0] // 2))
            lam = 1 - (box[2] - box[0]) * (box[3] - box[1]) / (cfg[0] * cfg[0])
            xm, yb = vo.cutmix_paste(x, perm, box), y[perm]
        else:
            perm, box, lam, xm, yb = None, None, 1.0, x, y
        for gi, grp in enumerate(r_opt.param_groups):
            grp["lr"] = vo.lr_at(it, base[gi], 200)
        r_loss, _ = vo.head_loss(ref_clip.visual(xm), r_head.weight, r_ls, y, yb, lam)
        r_opt.zero_grad()
        r_loss.backward()
        r_opt.step()
        assert [round(v, 12) for v in sched.get_last_lr()] == [round(vo.lr_at(it, b, 200), 12) for b in base]
        loss = step(x.to(dev()), y.to(dev()), yb.to(dev()), lam, perm=None if perm is None else perm.to(dev()), box=box)
        sched.step()
        assert abs(loss.item() - r_loss.item()) < 2e-2, (it, loss.item(), r_loss.item())
    assert rel(head.linear.weight, r_head.weight) < 1e-3
    assert abs(logit_scale.item() - r_ls.item()) < 1e-3
    assert rel(model.visual.proj, ref_clip.visual.proj) < 1e-3
    total_loss = step.pop_loss_sum()
    assert total_loss > 0
    assert _lib.lib().swat_vit_mega_launches() - launches0 == (8 if (cfg, B) == (B32, 32) else 0)
    assert _lib.lib().swat_vit_mega_backward_launches() - bwd0 == (8 if (cfg, B) == (B32, 32) else 0)


def test_validate_matches_oracle_eval():
    from swat_amd.clip_model import CLIP
    from swat_amd.models import MyLinear
    from swat_amd.testing import validate, calculate_scores
    torch.manual_seed(1)
    cfg, C = TINY, 7
    ref_clip = vo.OracleCLIP(vision_cfg=cfg, with_text=False)
    model = CLIP(vision_cfg=cfg, text_cfg=(8, 64, 64, 1, 1))
    model.visual.load_state_dict(ref_clip.visual.state_dict())
    model.to(dev())
    W = torch.randn(C, cfg[5])
    head = MyLinear(weights=W.clone()).to(dev())
    ls = torch.tensor([4.60517])
    data = [(torch.randn(9, 3, 64, 64), torch.randint(0, C, (9,)), None, None) for _ in range(3)]
    acc, loss, cm = validate(_args(), data, model, _Lg(), None, ls.to(dev()), head, show_confusion_matrix=True, device="cuda")
    losses, correct, n = [], 0, 0
    cm_ref = np.zeros((C, C), np.int64)
    with torch.no_grad():
        for x, y, _, _ in data:
            f = ref_clip.visual(x)
            f = f / f.norm(dim=-1, keepdim=True)
            logit = f @ W.t()
            losses.append(torch.nn.functional.cross_entropy(logit * ls.exp(), y).item())
            pred = logit.argmax(1)
            correct += (pred == y).sum().item()
            n += len(y)
            np.add.at(cm_ref, (y.numpy(), pred.numpy()), 1)
    assert abs(loss - np.mean(losses)) < 0.05 * max(1.0, np.mean(losses))
    assert abs(acc - 100.0 * correct / n) <= 100.0 / n * 2  # at most two borderline predictions flip under bf16
    assert cm.sum() == n and cm.shape == (C, C)
    s = calculate_scores(cm)
    assert abs(s["acc"] * 100 - acc) < 1e-9


def test_staged_backward_with_process_group_equals_monolithic_step():
    """The N > 1 code paths run on ONE GPU inside a 1-rank RCCL process group must reproduce the monolithic
    single-stream step bit for bit (same kernels, same order per tensor): stage-by-stage backward + a slice all-reduce
    after every stage, and the trailing form (weight-gradient lane + per-stage all-reduce / AdamW on a third stream)."""
    import os
    import torch.distributed as dist
    from swat_amd import optimizers
    from swat_amd.clip_model import CLIP
    from swat_amd.models import MyLinear
    from swat_amd.step import FusedTrainStep
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev())
        created = True
    try:
        results = []
        for staged, trail in ((False, False), (True, False), (True, True), (False, True)):
            torch.manual_seed(3)
            model = CLIP(vision_cfg=TINY, text_cfg=(8, 64, 64, 1, 1)).to(dev())
            W = torch.randn(10, TINY[5])
            head = MyLinear(weights=W).to(dev())
            args = _args()
            params, ls = optimizers.set_params(args, model, head, _Lg())
            opt, sched, _ = optimizers.set_optimizer(args, params, [None] * 50)
            step = FusedTrainStep(model, head, ls, opt, world_size=1, staged=staged, trail=trail)
            assert step.trail == trail
            if trail:
                step.reduce = True  # all-reduce every slice on the trailing stream (1-rank group)
            g = torch.Generator().manual_seed(9)
            for it in range(3):
                x = torch.randn(8, 3, 64, 64, generator=g).to(dev())
                y = torch.randint(0, 10, (8,), generator=g).to(dev())
                step(x, y, y.flip(0), 0.6, perm=torch.arange(7, -1, -1, device=dev()), box=(3, 9, 40, 50))
                sched.step()
            results.append((opt.arena.flat.clone(), opt.arena.grad.clone(), step.loss.clone()))
        for other in results[1:]:
            for a, b in zip(results[0], other):
                assert torch.equal(a, b)
    finally:
        if created:
            dist.destroy_process_group()


def test_text_tower_matches_oracle():
    """encode_text (token gather, causal blocks, ln_final at EOT, text_projection) vs the fp32 oracle."""
    from swat_amd.clip_model import CLIP
    torch.manual_seed(2)
    ctx, vocab, tw, th, tl = 16, 300, 128, 2, 3
    ref = vo.OracleCLIP(vision_cfg=TINY, text_width=tw, text_heads=th, text_layers=tl, ctx=ctx, vocab=vocab)
    model = CLIP(vision_cfg=TINY, text_cfg=(ctx, vocab, tw, th, tl))
    model.load_state_dict(ref.state_dict())
    model.to(dev())
    tok = torch.randint(1, vocab - 1, (6, ctx))
    for i in range(6):  # EOT (highest id) at a ragged position, zero padding after it
        e = 3 + 2 * i
        tok[i, e] = vocab - 1
        tok[i, e + 1:] = 0
    with torch.no_grad():
        want = ref.encode_text(tok)
    got = model.encode_text(tok.to(dev()))
    assert (got.cpu() - want).abs().max().item() <= 0.01 * want.abs().max().item()


def test_bf16_shadow_follows_load_state_dict_into_a_live_arena():
    """load_state_dict / p.copy_ into a model whose tower already lives in an arena (testing.load_model after _prepare,
    --skip_stage1, early-stop reload; ref testing.py:76-115, main.py:398-420) must refresh the bf16 shadow the GEMMs read:
    Parameters are attached with `p.data = view`, so torch bumps the Parameter's own version counter, not the arena's."""
    ref_a, mine = make_pair(TINY, seed=0)
    ref_b, fresh = make_pair(TINY, seed=7)
    x = torch.randn(3, 3, TINY[0], TINY[0], generator=torch.Generator().manual_seed(2)).to(dev())
    f_a = mine.forward_raw(x)                       # arena + shadow built from weights A
    mine.load_state_dict(ref_b.state_dict())        # weights B through the Parameters
    f_b = mine.forward_raw(x)
    f_fresh = fresh.forward_raw(x)
    assert (f_a - f_fresh).abs().max().item() > 1e-3, "the two weight sets must differ for the test to mean anything"
    assert torch.equal(f_b, f_fresh), "stale bf16 shadow: the GEMMs still read the weights from before load_state_dict"
    with torch.no_grad():                           # a single in-place write through one Parameter
        p = next(q for n, q in mine.named_parameters() if n.endswith("mlp.c_fc.weight"))
        p.mul_(1.5)
        q = next(q for n, q in fresh.named_parameters() if n.endswith("mlp.c_fc.weight"))
        q.mul_(1.5)
    a = mine.arena()
    a.refresh_shadow()
    o = a.offset_of(p)
    assert torch.equal(a.shadow[o:o + p.numel()].view(p.shape), p.detach().to(torch.bfloat16))
    assert torch.equal(mine.forward_raw(x), fresh.forward_raw(x))


def test_clip_deepcopy_drops_the_text_arena_cache():
    """copy.deepcopy(model) (ref main.py:555, utils/training.py:1534) must not drag the text tower's arena along; the copy
    rebuilds its own on first use and encodes the same."""
    from swat_amd.clip_model import CLIP
    torch.manual_seed(3)
    vis = (32, 16, 64, 1, 1, 32, "quick_gelu")  # head width 64 everywhere
    m = CLIP("tiny", vision_cfg=vis, text_cfg=(16, 100, 64, 1, 2)).to(dev())
    tok = torch.randint(1, 99, (4, 16), device=dev())
    tok[:, -1] = 99
    t0 = m.encode_text(tok)
    assert "_text_arena" in m.__dict__
    c = copy.deepcopy(m)
    assert "_text_arena" not in c.__dict__
    assert torch.equal(c.encode_text(tok), t0)
    with torch.no_grad():
        c.text_projection.mul_(2.0)
    assert torch.equal(m.encode_text(tok), t0), "the copy must own its parameters"
    assert torch.allclose(c.encode_text(tok), 2.0 * t0, rtol=1e-5, atol=1e-6)


def _hf_state(kind, name, tmp_path):
    """The golden case's HF-initialised weights (own process: transformers probes torchvision on import)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / f"{kind}_{name}.pt")
    code = f"import sys; sys.path.insert(0, {root!r}); from oracle import make_golden as mg; mg.export_hf_state({kind!r}, {name!r}, {path!r})"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return torch.load(path)


def _digest(t):
    f = t.detach().double().flatten().cpu()
    idx = torch.linspace(0, f.numel() - 1, 16).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item(), f.norm().item()], f[idx].numpy()])


@pytest.mark.parametrize("name", ["tiny_gelu", "b32_quickgelu"])
def test_vision_tower_matches_the_hf_golden_forward_and_backward(name, golden_dir, tmp_path):
    """SURVEY §8(c) G8 on the GPU: the HIP tower with the golden case's weights against what HF CLIP produced — embeddings
    and the gradients of six named parameters (Frobenius norm + 16 fixed samples each)."""
    from oracle import make_golden as mg
    from swat_amd.clip_model import VisionTransformer
    g = np.load(f"{golden_dir}/vit_hf.npz")
    (img, patch, width, layers, heads, embed), act = mg.VIT_CASES[name]
    model = VisionTransformer(img, patch, width, layers, heads, embed, act)
    model.load_state_dict(_hf_state("vit", name, tmp_path))
    model.to(dev())
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, img, img, generator=gen)
    w = torch.randn(2, embed, generator=gen)
    f = model.forward_raw(x.to(dev()), save=True)
    want = g[name + "_embeds"]
    assert np.abs(f.cpu().numpy() - want).max() <= EMB_TOL * np.abs(want).max()
    model.backward_raw(w.to(dev()).contiguous())
    arena = model.arena()
    params = dict(model.named_parameters())
    for n in mg.VIT_GRAD_NAMES:
        p = params[n]
        o = arena.offset_of(p)
        got = _digest(arena.grad[o:o + p.numel()].view(p.shape))
        ref = g[f"{name}_grad_{n}"]
        assert abs(got[2] - ref[2]) <= GRAD_TOL * ref[2], (n, got[2], ref[2])                   # Frobenius norm
        assert np.abs(got[3:] - ref[3:]).max() <= 0.03 * max(np.abs(ref[3:]).max(), ref[2] / math.sqrt(p.numel())), n  # 16 samples


@pytest.mark.parametrize("name", ["tiny_text", "b32_text"])
def test_text_tower_matches_the_hf_golden(name, golden_dir, tmp_path):
    """encode_text on the HIP kernels against HF CLIPTextModelWithProjection's text_embeds for the golden case (causal
    mask, EOT pooling by argmax, text_projection): the text tower's pin to an independent implementation."""
    from oracle import make_golden as mg
    from swat_amd.clip_model import CLIP
    g = np.load(f"{golden_dir}/text_hf.npz")
    ctx, vocab, width, heads, layers, embed, act = mg.TEXT_CASES[name]
    model = CLIP(vision_cfg=(32, 16, 64, 1, 1, embed, act), text_cfg=(ctx, vocab, width, heads, layers))
    missing = model.load_state_dict(_hf_state("text", name, tmp_path), strict=False)
    assert all(k.startswith("visual.") or k == "logit_scale" for k in missing.missing_keys), missing
    model.to(dev())
    tok = torch.from_numpy(g[name + "_tokens"])
    got = model.encode_text(tok.to(dev())).cpu().numpy()
    want = g[name + "_embeds"]
    assert np.abs(got - want).max() <= 0.01 * np.abs(want).max()


@pytest.mark.parametrize("B", [32, 7, 40])
def test_persistent_forward_equals_the_op_by_op_forward_and_ignores_history(B):
    """csrc/vit_mega.hip — all 12 blocks of the ViT-B/32 forward in ONE persistent launch (teams of 16 workgroups hand
    panels of 2 images from phase to phase) — against the op-by-op composite it replaces: same buffers, same rounding points
    (the two differ only in the k order of a few products), identical results from run to run, independent of what the
    previous launch left in the workspace / caches (hand-off protocol), the eval form (activations not saved: buffers re-used
    layer after layer, agent-scope acquire at every hand-off) and the backward that consumes the saved activations."""
    from swat_amd import _lib
    from swat_amd.clip_model import CLIP
    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(B)
    model = CLIP("ViT-B-32-quickgelu").to(dev)
    vis = model.visual
    xa, xb = torch.randn(B, 3, 224, 224, device=dev), torch.randn(B, 3, 224, 224, device=dev)
    df = torch.randn(B, vis.output_dim, device=dev)
    try:
        L.swat_vit_set_mega(0)
        f0 = vis.forward_raw(xb, save=True).clone()
        vis.backward_raw(df)
        g0 = vis.arena().grad.clone()
        e0 = vis.forward_raw(xb, save=False).clone()
        L.swat_vit_set_mega(2)                                     # every batch size (1 = only batches that fill whole rounds of panels)
        vis.forward_raw(xa, save=True)
        nb0 = L.swat_vit_mega_backward_launches()
        f1 = vis.forward_raw(xb, save=True).clone()               # history: a launch on other images
        vis.backward_raw(df)
        g1 = vis.arena().grad.clone()
        w1 = vis._ws[(B, True)].clone()
        f2 = vis.forward_raw(xb, save=True).clone()               # history: the same images
        vis.backward_raw(df)
        g2 = vis.arena().grad.clone()
        w2 = vis._ws[(B, True)].clone()
        nb = L.swat_vit_mega_backward_launches() - nb0
        assert nb == 2 if B == 32 else nb in (0, 2)                 # B = 32: the blocks' data gradients ran as the persistent backward
        vis.forward_raw(xa, save=False)
        e1 = vis.forward_raw(xb, save=False).clone()
        assert L.swat_vit_mega_status() == 0
    finally:
        L.swat_vit_set_mega(1)
    tol = 0.004 * f0.abs().max().item()
    assert (f1 - f0).abs().max().item() <= tol and (e1 - e0).abs().max().item() <= tol
    assert torch.equal(f1, f2) and torch.equal(e1, f1)             # run to run, saved or not: bit-identical
    # forward + backward twice on the same inputs: every byte of the workspace (saved activations, the per-block gradient buffers of
    # the persistent backward, scratch) and every gradient identical — no atomics, fixed summation orders, hand-offs that publish
    # complete data
    assert torch.equal(w1, w2) and torch.equal(g1, g2)
    L.swat_vit_set_mega(2)
    try:
        f3 = vis.forward_raw(xb, save=True)
        assert torch.equal(vis._ws[(B, True)], w2) and torch.equal(f3, f2)
    finally:
        L.swat_vit_set_mega(1)
    # gradients behind the persistent forward vs behind the op-by-op forward: per tensor, within the oracle tolerance (the two
    # forwards differ in the k order of a few products only; the oracle itself is compared at B = 32 in
    # test_forward_backward_match_oracle[cfg3-32])
    arena = vis.arena()
    for n, p in vis.named_parameters():
        o = arena.offset_of(p)
        r = ((g1[o:o + p.numel()] - g0[o:o + p.numel()]).norm() / (g0[o:o + p.numel()].norm() + 1e-12)).item()
        assert r < GRAD_TOL, f"{n}: {r}"


@pytest.mark.parametrize("B", [32, 7, 16, 40])   # 40: twenty panels on sixteen teams — a second, ragged round
def test_persistent_kernels_do_not_depend_on_placement_or_wave_roles(B):
    """Round 6 (csrc/vit_mega.hip): (i) where a team's sixteen workgroups sit is for speed only — a team spread over 1, 2, 4 or 8 XCDs
    (mg_place / swat_vit_set_mega_xsplit; default 4: an XCD then sees a quarter of every weight matrix, read by eight teams) gives the
    same bytes in the whole workspace and every gradient; (ii) the forward with fixed wave roles (the loader waves move operand bytes
    only and carry the next product's weight stages in registers: swat_vit_set_mega_roles(2), measured and not adopted as the default)
    gives the same bytes as the kernel in which every wave takes part in every phase."""
    from swat_amd import _lib
    from swat_amd.clip_model import CLIP
    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(B)
    model = CLIP("ViT-B-32-quickgelu").to(dev)
    vis = model.visual
    x = torch.randn(B, 3, 224, 224, device=dev)
    df = torch.randn(B, vis.output_dim, device=dev)
    outs = []
    try:
        L.swat_vit_set_mega(2)
        for roles, xs in ((1, 1), (1, 2), (1, 4), (1, 8), (2, 4), (2, 1)):
            assert L.swat_vit_set_mega_roles(roles) == 0 and L.swat_vit_set_mega_xsplit(xs) == 0
            n0 = L.swat_vit_mega_launches()
            vis._ws = {}
            f = vis.forward_raw(x, save=True).clone()
            vis.backward_raw(df)
            torch.cuda.synchronize()
            assert L.swat_vit_mega_launches() - n0 == 1
            assert L.swat_vit_mega_status() == 0
            outs.append((f, vis._ws[(B, True)].clone(), vis.arena().grad.clone()))
            e = vis.forward_raw(x, save=False)   # the eval form re-uses its buffers block after block (acquire at every hand-off)
            assert torch.equal(e, f)
    finally:
        L.swat_vit_set_mega(1)
        L.swat_vit_set_mega_roles(1)
        L.swat_vit_set_mega_xsplit(4)
    for f, w, g in outs[1:]:
        assert torch.equal(f, outs[0][0]) and torch.equal(w, outs[0][1]) and torch.equal(g, outs[0][2])


@pytest.mark.parametrize("cfg", [TINY_GELU, TINY])
def test_owed_join_leaves_the_trajectory_bit_identical(cfg, monkeypatch):
    """the join with the weight-gradient lane may be OWED to the next forward only when the step that owes it
    stepped stage 0 on the main stream (the fused ViT-B/32 step).  Every other shape — here patch 14, whose stem first re-packs the fp32
    conv1 MASTER weights (588 columns) — must see the lane's AdamW before its stem: SWAT_LAZY_JOIN=1 and =0 give the same bits."""
    from swat_amd import optimizers
    from swat_amd.clip_model import CLIP
    from swat_amd.models import MyLinear
    from swat_amd.step import FusedTrainStep

    def run(lazy):
        monkeypatch.setenv("SWAT_LAZY_JOIN", lazy)
        torch.manual_seed(0)
        model = CLIP(vision_cfg=cfg, text_cfg=(8, 64, 64, 1, 1)).to(dev())
        W = torch.randn(10, cfg[5])
        head = MyLinear(weights=W / W.norm(dim=-1, keepdim=True)).to(dev())
        args = _args(lr_backbone=1e-3)   # (large enough for every step to move conv1 / cls / pos / ln_pre in bf16)
        params, ls = optimizers.set_params(args, model, head, _Lg())
        opt, sched, _ = optimizers.set_optimizer(args, params, [None] * 100)
        step = FusedTrainStep(model, head, ls, opt, world_size=1)
        assert step.lazy_join == (lazy == "1")
        g = torch.Generator().manual_seed(5)
        losses = []
        for it in range(6):
            x = torch.randn(48, 3, cfg[0], cfg[0], generator=g).to(dev())
            y = torch.randint(0, 10, (48,), generator=g).to(dev())
            losses.append(step(x, y, y, 1.0).detach().clone())
            sched.step()
        step.join()
        torch.cuda.synchronize()
        return torch.stack([l.detach().reshape(()) for l in losses]).cpu(), step.arena.flat.detach().cpu().clone()

    l1, w1 = run("1")
    l0, w0 = run("0")
    assert torch.equal(l1, l0)
    assert torch.equal(w1, w0)


def _b32_step(seed=0, C=200):
    from swat_amd import optimizers
    from swat_amd.clip_model import CLIP
    from swat_amd.models import MyLinear
    from swat_amd.step import FusedTrainStep
    torch.manual_seed(seed)
    model = CLIP("ViT-B-32-quickgelu").to(dev())
    W = torch.randn(C, 512)
    W /= W.norm(dim=-1, keepdim=True)
    head = MyLinear(weights=W).to(dev())
    args = _args()
    params, ls = optimizers.set_params(args, model, head, _Lg())
    opt, sched, _ = optimizers.set_optimizer(args, params, [None] * 100)
    return model, head, opt, sched, FusedTrainStep(model, head, ls, opt, world_size=1)


def _one_rank_group():
    """A 1-rank RCCL process group on the test GPU: the N > 1 code of the step runs unchanged inside it (every collective is the
    identity).  -> True when this call created the group (the caller destroys it)."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if dist.is_initialized():
        return False
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev())
    return True


def _b32_dp_step(seed=0):
    """The benchmark's step as a data-parallel rank runs it (gradient buckets, all-reduce and AdamW on their own streams)."""
    from swat_amd.step import FusedTrainStep
    model, head, opt, sched, _ = _b32_step(seed)
    ls = next(g["params"][0] for g in opt.param_groups if g["params"][0].numel() == 1)
    step = FusedTrainStep(model, head, ls, opt, world_size=1, staged=True)
    step.reduce = True
    return model, head, opt, sched, step


def test_data_parallel_step_runs_the_persistent_backward_and_equals_the_one_gpu_step():
    """SURVEY 8(e) row 1 / round-4 verdict: at N > 1 the blocks' data gradients are the SAME one persistent launch as on one GPU; on
    the lane behind it each gradient bucket's weight gradients are one launch with an event, the bucket's all-reduce and AdamW trail
    it (swat_vit_backward_set_buckets).  Run inside a 1-rank RCCL group the data-parallel step must land on the one-GPU step's
    weights, moments, bf16 shadow and losses bit for bit — same kernels, same order per tensor, only streams and launch
    groupings differ (ref: utils/training.py:1474-1489, one process in the reference)."""
    import torch.distributed as dist
    from swat_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(13)
    data = [(torch.randn(32, 3, 224, 224, generator=g).to(dev()), torch.randint(0, 200, (32,), generator=g).to(dev())) for _ in range(3)]

    def run(make):
        model, head, opt, sched, step = make(seed=4)
        nb0 = L.swat_vit_mega_backward_launches()
        losses = []
        for it, (x, y) in enumerate(data):
            if it % 2 == 0:
                step(x, y, y.flip(0), 0.6, perm=torch.arange(31, -1, -1, device=dev()), box=(30, 40, 150, 190))
            else:
                step(x, y, y, 1.0)
            sched.step()
            losses.append(step.loss.clone())
        total = step.pop_loss_sum()
        assert L.swat_vit_mega_backward_launches() - nb0 == len(data) and L.swat_vit_mega_status() == 0
        a = opt.arena
        return step, (torch.stack(losses).cpu(), a.flat.clone(), a.exp_avg.clone(), a.exp_avg_sq.clone(), a.shadow.clone(), a.grad.clone()), total

    created = _one_rank_group()
    try:
        _, one, total1 = run(_b32_step)
        step, dp, total2 = run(_b32_dp_step)
        assert step.reduce and step.trail and step._dp_fused
        plan = step._bucket_plan()
        assert [len([s for s in b if 1 <= s <= 12]) for b in plan] == [4, 4, 2, 1, 1] and plan[0][0] == 13 and plan[-1] == [1, 0]
        assert L.swat_vit_backward_buckets_recorded() == len(plan)     # one weight-gradient launch + event per bucket
    finally:
        if created:
            dist.destroy_process_group()
    for name, x, y in zip(("loss", "weights", "exp_avg", "exp_avg_sq", "bf16 shadow", "gradient"), one, dp):
        assert torch.equal(x, y), f"{name}: {(x.float() - y.float()).abs().max().item()}"
    assert total1 == total2 and total1 > 0


def test_data_parallel_stage_by_stage_backward_at_the_benchmark_shape_tracks_the_one_gpu_step(monkeypatch):
    """The other data-parallel form the step may choose when the links are slow (step.py:_decide_dp_path; SWAT_DP_PERSISTENT=0 forces it):
    persistent forward, then the backward stage by stage with each stage's slice reduced beside the next stage's kernels.  Same
    arithmetic per tensor up to the order of a few sums: loss equal, gradients within 4e-3 of the one-GPU step's in the norm, and no
    persistent backward launch."""
    import torch.distributed as dist
    from swat_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(19)
    x, y = torch.randn(32, 3, 224, 224, generator=g).to(dev()), torch.randint(0, 200, (32,), generator=g).to(dev())

    def run(make):
        model, head, opt, sched, step = make(seed=6)
        nb0 = L.swat_vit_mega_backward_launches()
        step(x, y, y.flip(0), 0.6, perm=torch.arange(31, -1, -1, device=dev()), box=(30, 40, 150, 190))
        loss = step.loss.clone()
        step.pop_loss_sum()
        assert L.swat_vit_mega_status() == 0
        return step, loss.cpu(), opt.arena.grad.clone(), L.swat_vit_mega_backward_launches() - nb0

    created = _one_rank_group()
    try:
        _, l1, g1, n1 = run(_b32_step)
        monkeypatch.setenv("SWAT_DP_PERSISTENT", "0")
        step, l2, g2, n2 = run(_b32_dp_step)
        assert step.reduce and not step._dp_fused
    finally:
        if created:
            dist.destroy_process_group()
    assert n1 == 1 and n2 == 0
    assert abs(l1.item() - l2.item()) < 1e-5 * abs(l1.item())
    assert ((g2 - g1).norm() / g1.norm()).item() < 4e-3


def test_data_parallel_step_with_bf16_gradients_on_the_wire_over_rccl(monkeypatch):
    """The --dp_weak / bench default (SWAT_DP_WIRE=bf16, ddp.GradWire): each bucket is cast to bf16, all-reduced by RCCL as bf16 and widened
    back before its AdamW.  In a 1-rank RCCL group the collective is the identity, so the gradient the optimizer reads is the one-GPU
    step's gradient rounded to bf16 — element by element — and the first step's loss is the same number."""
    import torch.distributed as dist
    from swat_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(17)
    x, y = torch.randn(32, 3, 224, 224, generator=g).to(dev()), torch.randint(0, 200, (32,), generator=g).to(dev())
    monkeypatch.setenv("SWAT_DP_WIRE", "bf16")
    monkeypatch.setenv("SWAT_DP_CALIBRATE_FORCE", "1")   # ... and the step times a collective to choose its backward (step.py:_decide_dp_path)

    def run(make):
        model, head, opt, sched, step = make(seed=5)
        w0 = opt.arena.flat.clone()
        step(x, y, y.flip(0), 0.6, perm=torch.arange(31, -1, -1, device=dev()), box=(30, 40, 150, 190))
        loss = step.loss.clone()
        step.pop_loss_sum()
        assert L.swat_vit_mega_status() == 0
        return step, loss.cpu(), opt.arena.grad.clone(), opt.arena.flat.clone() - w0

    created = _one_rank_group()
    try:
        _, l1, g1, w1 = run(_b32_step)
        step, l2, g2, w2 = run(_b32_dp_step)
        assert step._wire().wire == "bf16" and step._dp_fused
        cal = step.dp_calibration   # a 1-rank collective costs next to nothing: the persistent form stays
        assert cal and cal["path"].startswith("persistent") and 0 <= cal["gradient_ms_on_the_links"] < cal["break_even_ms"]
    finally:
        if created:
            dist.destroy_process_group()
    assert torch.equal(l1, l2)
    assert torch.equal(g2, g1.to(torch.bfloat16).float()), (g2 - g1).abs().max().item()
    assert (w2 - w1).norm() <= 0.2 * w1.norm()   # (w = the first AdamW step's move: only signs of rounding-level gradients may flip)


@pytest.mark.parametrize("when", ["early", "last"])
def test_data_parallel_give_up_voids_the_step_and_every_rank_replays(when):
    """A hand-off give-up at N > 1 (round-4 verdict: it used to raise).  The step's flag exchange (swat_vit_mega_void_flag ->
    all-reduce -> swat_vit_mega_void_absorb, before the step's first optimizer launch) raises the guard; the host acts on it
    VOID_LAG steps later ('early': injected at step 2 of 7) or at the epoch's synchronisation ('last': injected at the last step),
    switches to the op-by-op path and replays — the trajectory equals a run that never used the persistent launches, no step is
    lost or doubled (second moments), the loss sum holds each step once."""
    import torch.distributed as dist
    from swat_amd import _lib
    L = _lib.lib()
    n = 7
    g = torch.Generator().manual_seed(6)
    data = [(torch.randn(32, 3, 224, 224, generator=g).to(dev()), torch.randint(0, 200, (32,), generator=g).to(dev())) for _ in range(n)]
    bad = 1 if when == "early" else n - 1

    def run(inject):
        model, head, opt, sched, step = _b32_dp_step(seed=2)
        for it, (x, y) in enumerate(data):
            if inject and it == bad:
                L.swat_vit_mega_backward_inject_failure(1, 9)
            step(x, y, y.flip(0), 0.7, perm=torch.arange(31, -1, -1, device=dev()), box=(10, 20, 120, 200))
            sched.step()
        total = step.pop_loss_sum()
        return total, opt.arena.flat.clone(), opt.arena.exp_avg_sq.clone(), opt.step_count

    created = _one_rank_group()
    try:
        L.swat_vit_set_mega(0)
        want = run(False)
        L.swat_vit_set_mega(1)
        nb0 = L.swat_vit_mega_backward_launches()
        with pytest.warns(UserWarning, match=f"data-parallel step {bad + 1}"):
            got = run(True)
        assert L.swat_vit_mega_backward_launches() - nb0 >= bad + 1
        assert L.swat_vit_mega_poll() == 0 and L.swat_vit_mega_poll_void() == 0 and L.swat_vit_mega_status() == 0
    finally:
        L.swat_vit_mega_backward_inject_failure(0, 0)
        L.swat_vit_set_mega(1)
        if created:
            dist.destroy_process_group()
    assert got[3] == want[3] == n
    assert abs(got[0] - want[0]) < 1e-2 * abs(want[0])
    assert rel(got[1], want[1]) < 1e-5 and rel(got[2], want[2]) < 5e-2


def test_adamw_inside_the_weight_gradient_launch_equals_the_separate_optimizer_launch():
    """swat_vit_arm_fused_adamw: behind the persistent backward the blocks' AdamW runs inside the backward call — piped (a block's
    AdamW on a second stream beside the next block's weight gradients) or in the epilogue of ONE weight-gradient launch over all 48
    products (csrc/gemm_xl.hip table form: AdamW on the tile in the accumulators, the gradient never goes to memory; biases and
    LayerNorm weights follow in one launch).  Same arithmetic as swat_adamw (ref utils/optimizers.py:131-146): weights, both moments,
    the bf16 shadow and the losses of a 4-step run are bit-identical to the run with the separate optimizer launch."""
    from swat_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    data = [(torch.randn(32, 3, 224, 224, generator=g).to(dev()), torch.randint(0, 200, (32,), generator=g).to(dev())) for _ in range(4)]

    def run(mode):
        model, head, opt, sched, step = _b32_step(seed=3)
        step.fuse_adamw = mode > 0
        L.swat_vit_set_block_adamw(max(mode, 1))
        nb0 = L.swat_vit_mega_backward_launches()
        losses = []
        for x, y in data:
            step(x, y, y.flip(0), 0.6, perm=torch.arange(31, -1, -1, device=dev()), box=(30, 40, 150, 190))
            sched.step()
            losses.append(step.loss.clone())
        torch.cuda.synchronize()
        assert L.swat_vit_mega_backward_launches() - nb0 == 4 and L.swat_vit_mega_status() == 0
        a = opt.arena
        return torch.stack(losses).cpu(), a.flat.clone(), a.exp_avg.clone(), a.exp_avg_sq.clone(), a.shadow.clone()

    try:
        sep = run(0)      # the caller's own swat_adamw launch over the blocks' arena range
        piped = run(1)    # block by block beside the next block's weight gradients (the default)
        epi = run(2)      # in the epilogue of one weight-gradient launch over all blocks
    finally:
        L.swat_vit_set_block_adamw(1)
    for form, got in (("piped", piped), ("epilogue", epi)):
        for name, x, y in zip(("loss", "weights", "exp_avg", "exp_avg_sq", "bf16 shadow"), sep, got):
            assert torch.equal(x, y), f"{form}: {name}: {(x.float() - y.float()).abs().max().item()}"
    assert not torch.equal(sep[1], _b32_step(seed=3)[2].arena.flat)   # (the weights did move)


@pytest.mark.parametrize("which", ["forward", "backward"])
def test_persistent_forward_give_up_is_detected_and_the_step_is_re_run_op_by_op(which):
    """Hardening of csrc/vit_mega.hip (round-3 advisor finding): a team member that never arrives (here injected: it skips one
    arrival, spin bound 2 ms) must not cost the job its weights.  The launch flags the give-up on the device and in a pinned host
    word; every AdamW launch and the running loss sum skip their writes while the flag stands; the next call of the step notices
    (no synchronisation needed), switches the process to the op-by-op forward and runs the voided steps again.  The trajectory
    must equal a run that never used the persistent forward.  "backward": the same for the persistent BACKWARD (vit_bwd_mega_kernel) —
    by then the step's loss is computed, but no optimizer launch of the step may have run yet (the head's, too, are queued behind
    the persistent backward): the give-up voids the step as a whole."""
    from swat_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    data = [(torch.randn(32, 3, 224, 224, generator=g).to(dev()), torch.randint(0, 200, (32,), generator=g).to(dev())) for _ in range(4)]

    def run(inject):
        model, head, opt, sched, step = _b32_step()
        losses = []
        for it, (x, y) in enumerate(data):
            if inject and it == 1:   # the 2nd step: member 1 of team 0 misses hand-off 9 (forward: block 2; backward: block 10)
                (L.swat_vit_mega_inject_failure if which == "forward" else L.swat_vit_mega_backward_inject_failure)(1, 9)
            step(x, y, y.flip(0), 0.7, perm=torch.arange(31, -1, -1, device=dev()), box=(10, 20, 120, 200))
            sched.step()
            losses.append(step.loss.clone())
        total = step.pop_loss_sum()
        return torch.stack(losses).cpu(), total, opt.arena.flat.clone(), opt.arena.exp_avg_sq.clone(), opt.step_count

    try:
        L.swat_vit_set_mega(0)
        want = run(False)
        L.swat_vit_set_mega(1)
        n0 = L.swat_vit_mega_launches()
        with pytest.warns(UserWarning, match="gave up on a team hand-off"):
            got = run(True)
        assert L.swat_vit_mega_launches() - n0 >= 2, "the first two steps ran the persistent forward"
        assert L.swat_vit_mega_poll() == 0 and L.swat_vit_mega_status() == 0
    finally:
        L.swat_vit_mega_inject_failure(0, 0)
        L.swat_vit_mega_backward_inject_failure(0, 0)
        L.swat_vit_set_mega(1)
    assert got[4] == want[4] == 4
    assert abs(got[1] - want[1]) < 1e-2 * abs(want[1])          # the loss sum holds each step once (no void step in it)
    # step 1 ran the persistent forward, steps 2-4 op by op after the recovery: same weights as the all-op-by-op run up to
    # the persistent forward's k order in step 1
    # (the second moments tell a lost or doubled step apart from rounding: one step is >= 25 % of them after four steps)
    assert rel(got[2], want[2]) < 1e-5 and rel(got[3], want[3]) < 5e-2
    assert abs(got[0][0].item() - want[0][0].item()) < 2e-2


def test_two_towers_on_two_streams_run_the_persistent_forward_without_sharing_hand_off_state():
    """WiSE-FT / zero-shot + fine-tuned towers, or two host threads: persistent forwards issued on two streams at once.  The
    library serialises persistent launches per device (two of them in flight could each hold part of the chip and wait for
    members the other keeps from becoming resident) — results must equal the one-stream results and no hand-off may give up."""
    from swat_amd import _lib
    from swat_amd.clip_model import CLIP
    L = _lib.lib()
    torch.manual_seed(1)
    va, vb = CLIP("ViT-B-32-quickgelu").to(dev()).visual, CLIP("ViT-B-32-quickgelu").to(dev()).visual
    xa, xb = torch.randn(32, 3, 224, 224, device=dev()), torch.randn(32, 3, 224, 224, device=dev())
    want_a, want_b = va.forward_raw(xa).clone(), vb.forward_raw(xb).clone()
    torch.cuda.synchronize()
    n0 = L.swat_vit_mega_launches()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for it in range(6):
        with torch.cuda.stream(sa):
            fa = va.forward_raw(xa)
        with torch.cuda.stream(sb):
            fb = vb.forward_raw(xb)
        outs.append((fa, fb))
    torch.cuda.synchronize()
    assert L.swat_vit_mega_launches() - n0 == 12 and L.swat_vit_mega_poll() == 0
    for fa, fb in outs:
        assert torch.equal(fa, want_a) and torch.equal(fb, want_b)

