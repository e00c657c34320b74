"""CPU: the product's host-side logic against golden vectors recorded from the imported reference
(tests/golden/host_logic.json, loop_tiny.npz — oracle/make_golden.py `host` / `loop`)."""
import json
import os
import sys
import types

import pytest

import numpy as np
import torch


def _gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "host_logic.json")))


class _Opt:
    def __init__(self, lrs):
        self.param_groups = [{"lr": lr, "initial_lr": lr} for lr in lrs]


def test_lr_schedule_matches_reference_trace(golden_dir):
    """LinearWarmupScheduler over CosineAnnealingLR (ref utils/optimizers.py:22-35,158-206), 3 LR groups, 300 its."""
    from swat_amd.optimizers import LinearWarmupScheduler
    for key, rec in _gold(golden_dir)["lr_trace"].items():
        lrb = float(key.split("_")[1])
        opt = _Opt([1e-4, lrb, 1e-4])
        sched = LinearWarmupScheduler(opt, rec["total_iter"], warmup_iter=50, warmup_lr=1e-5 if lrb > 5e-5 else 1e-6)
        for it, want in enumerate(rec["lrs"]):
            got = [g["lr"] for g in opt.param_groups]
            assert np.allclose(got, want, rtol=1e-9, atol=1e-18), (key, it, got, want)
            sched.step()


def test_rand_bbox_matches_reference(golden_dir):
    from swat_amd.training import rand_bbox
    for i, (lam, x1, y1, x2, y2) in enumerate(_gold(golden_dir)["rand_bbox"]):
        np.random.seed(1000 + i)
        assert [int(v) for v in rand_bbox((32, 3, 224, 224), lam)] == [x1, y1, x2, y2]


def test_mixup_criterion_matches_reference(golden_dir):
    from swat_amd.training import mixup_criterion
    rec = _gold(golden_dir)["mixup_criterion"]
    g = torch.Generator().manual_seed(rec["seed"])
    logits = torch.randn(32, 200, generator=g) * 3
    ya = torch.randint(0, 200, (32,), generator=g)
    yb = torch.randint(0, 200, (32,), generator=g)
    ce = torch.nn.CrossEntropyLoss()
    for lam, want in rec["vals"]:
        assert abs(float(mixup_criterion(ce, logits, ya, yb, lam)) - want) < 1e-6


def test_calculate_scores_matches_reference(golden_dir):
    from swat_amd.testing import calculate_scores
    rec = _gold(golden_dir)["validate"]
    s = calculate_scores(np.array(rec["cm"]))
    assert abs(s["acc"] - rec["scores"]["acc"]) < 1e-12
    assert np.isclose(s["avg_class_accuracy"], rec["scores"]["avg_class_accuracy"], equal_nan=True)
    C = rec["C"]
    assert np.allclose([s["per_class_recall"][i] for i in range(C)], rec["scores"]["recall"])
    assert np.allclose([s["per_class_precision"][i] for i in range(C)], rec["scores"]["precision"])
    assert np.allclose([s["per_class_f1score"][i] for i in range(C)], rec["scores"]["f1"])


def test_cutmix_host_rng_transcript_matches_reference_loop(golden_dir):
    """Same seeds -> the product's cutmix_draw consumes the global NumPy / torch generators in the reference's order
     and reproduces every (lam', target_b) the reference's train_cutmix fed to mixup_criterion."""
    from oracle.make_golden import loop_inputs
    from swat_amd.training import cutmix_draw, set_training_seed
    g = np.load(os.path.join(golden_dir, "loop_tiny.npz"), allow_pickle=False)
    transcript = json.loads(str(g["cutmix_transcript"]))
    train = loop_inputs(11, 4, 8, 10)
    args = types.SimpleNamespace(cutmix_beta=1.0, mix_prob=0.5, training_seed=1)
    set_training_seed(args)
    it = 0
    for epoch in range(2):
        for x, y, _, _ in train:
            lam, perm, box = cutmix_draw(args, x.size())
            yb = y[perm] if perm is not None else y
            assert abs(lam - transcript[it]["lam"]) < 1e-15, it
            assert yb.tolist() == transcript[it]["y_b"], it
            if box is not None:
                assert abs(lam - (1 - (box[2] - box[0]) * (box[3] - box[1]) / (64 * 64))) < 1e-15
            it += 1
    assert it == len(transcript)


def test_model_cfg_mapping():
    from swat_amd.engine import parse_model_cfg
    assert parse_model_cfg("vitb32_openclip_laion400m") == ("ViT-B-32-quickgelu", "laion400m_e32")
    assert parse_model_cfg("vitb16_openclip_laion400m") == ("ViT-B-16", "laion400m_e32")
    assert parse_model_cfg("vitb32_openclip_laion2b") == ("ViT-B-32", "laion2b_s34b_b79k")
    assert parse_model_cfg("vitl14_openclip_laion400m") == ("ViT-L-14", "laion400m_e32")


def test_state_dict_uses_open_clip_key_names():
    from oracle import vit_oracle as vo
    from swat_amd.clip_model import CLIP
    cfg = (64, 16, 128, 2, 2, 64, "quick_gelu")
    mine = CLIP(vision_cfg=cfg, text_cfg=(8, 64, 64, 1, 1))
    ref = vo.OracleCLIP(vision_cfg=cfg, text_width=64, text_heads=1, text_layers=1, ctx=8, vocab=64)
    a = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert a == b
    assert "visual.transformer.resblocks.1.attn.in_proj_weight" in a and "visual.proj" in a and "logit_scale" in a
    mine.load_state_dict(ref.state_dict())


def test_tower_refuses_cpu():
    import pytest
    from swat_amd import _lib
    from swat_amd.clip_model import CLIP
    m = CLIP(vision_cfg=(64, 16, 128, 2, 2, 64, "quick_gelu"), text_cfg=(8, 64, 64, 1, 1))
    with pytest.raises(_lib.SwatHipError):
        m.encode_image(torch.zeros(1, 3, 64, 64))


def test_segmented_index_save_load_round_trip(tmp_path):
    """The on-disk index (raw .npy arrays in grouped row order): whole and as a row-range shard."""
    from swat_amd.retrieval.index import SegmentedIndex
    g = torch.Generator().manual_seed(0)
    n = 1000
    lab = torch.randint(0, 17, (n,), generator=g)
    img, cap = torch.randn(n, 64, generator=g), torch.randn(n, 64, generator=g)
    ix = SegmentedIndex(img, cap, lab, filepath=[f"p{i}" for i in range(n)], device="cpu")
    ix.save(str(tmp_path / "ix"))
    full = SegmentedIndex.load(str(tmp_path / "ix"), device="cpu")
    assert torch.equal(full.caption_features, ix.caption_features) and torch.equal(full.image_features, ix.image_features)
    assert torch.equal(full.offsets, ix.offsets) and torch.equal(full.order, ix.order) and np.array_equal(full.filepath, ix.filepath)
    sh = SegmentedIndex.load(str(tmp_path / "ix"), device="cpu", rows=(300, 700))
    assert torch.equal(sh.image_features, ix.image_features[300:700]) and np.array_equal(sh.filepath, ix.filepath[300:700])
    assert int(sh.offsets[0]) == 0 and int(sh.offsets[-1]) == 400 and torch.equal(sh.row_seg, ix.row_seg[300:700])
    assert torch.equal(sh.offsets, ix.offsets.clamp(300, 700) - 300)


def test_ensemble_model_matches_the_reference_golden(golden_dir):
    """WiSE-FT weight-space interpolation (ref main.py:76-98) on the fp32 oracle models of the stages golden: every
    state-dict tensor of the interpolated model (digest) and its embeddings equal what the reference's function produced."""
    import numpy as np
    import torch
    from oracle import make_golden as mg
    from swat_amd.stages import ensemble_model
    g = np.load(f"{golden_dir}/stages_tiny.npz")
    (best, _), (zero, _) = mg.stage_models()
    x = mg.stage_loader(21, 22, 8)[0][0]
    for alpha in (0.5, 0.3):
        w = ensemble_model(best, zero, alpha)
        sd = w.state_dict()
        dig = np.stack([mg._digest(sd[k].float()) for k in sorted(sd) if sd[k].dtype.is_floating_point])
        assert np.allclose(dig, g[f"wsft_{alpha}_digest"], rtol=1e-6, atol=1e-7)
        with torch.no_grad():
            assert np.allclose(w.encode_image(x).numpy(), g[f"wsft_{alpha}_embeds"], atol=1e-5)


def test_prompt_makers_match_reference_golden(golden_dir):
    """prompt_maker / prompt_maker_aves (ref utils/prompt.py:83-148): same keys, order and prompt strings as the imported
    reference for every dataset whose template list is built in (tests/golden/prompts.json, oracle/make_golden.py prompts)."""
    import json
    import os
    import pytest
    from oracle.make_golden import prompt_metrics
    from swat_amd import prompt
    g = json.load(open(os.path.join(golden_dir, "prompts.json")))
    m = prompt_metrics()
    for key, want in g["prompt_maker"].items():
        ds, nt = key.split("|")
        got = prompt.prompt_maker(m, ds, nt)
        assert list(got.keys()) == list(want.keys()) and got == want, key
    for nt, want in g["prompt_maker_aves"].items():
        got = prompt.prompt_maker_aves(m, "semi-aves", nt)
        assert list(got.keys()) == list(want.keys()) and got == want, nt
    with pytest.raises(KeyError):
        prompt.prompt_maker(m, "imagenet", "name")  # the 80-template list comes from SWAT_TEMPLATES_JSON
    assert prompt.prompt_maker(m, "some-new-dataset", "name")["0"]["corpus"] == ["a photo of a barn owl."]


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """`bench.py --gpus 8` inside a 1-rank (or 2-rank) environment is a hard error, never a silent 1-GPU number
    (VERDICT r02 missing #1 / ADVICE medium); without WORLD_SIZE it launches the ranks itself."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_self_launch_starts_one_rank_per_gpu(monkeypatch):
    import importlib.util
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return subprocess.CompletedProcess(cmd, 0, stdout='noise\n{"n_gpus": 4}\n', stderr="")
    monkeypatch.setattr(subprocess, "run", fake_run)
    assert bench.self_launch(4, ["--gpus", "4", "--steps", "3"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]


def test_gpu_input_pipeline_is_the_default_and_the_reference_pipeline_an_opt_out():
    """Round-4 verdict, item 6: recipes run unchanged (ref scripts/run_dataset_seed_SWAT.sh:85-88 pass no loader flags) get the GPU
    decode + transform; --host_decode keeps Pillow decode in the workers, --host_transform the reference's whole pipeline
    (utils/dataloader.py:107-135); the round-3 / round-4 flags keep their meaning."""
    from swat_amd.dataloader import input_pipeline
    from swat_amd.parser import build_parser
    p = build_parser()
    assert input_pipeline(p.parse_args([])) == "gpu_decode"
    assert input_pipeline(p.parse_args(["--host_decode"])) == "gpu_transform"
    assert input_pipeline(p.parse_args(["--host_transform"])) == "host"
    assert input_pipeline(p.parse_args(["--gpu_transform"])) == "gpu_transform"
    assert input_pipeline(p.parse_args(["--gpu_decode"])) == "gpu_decode"
    assert input_pipeline(p.parse_args(["--gpu_transform", "--gpu_decode"])) == "gpu_decode"


def test_cutmix_permutation_drawn_into_a_buffer_is_the_reference_draw():
    """training.cutmix_draw draws the permutation with torch.randperm(n, out=<pinned buffer>) when a GPU is present: the same
    generator consumption and the same values as the reference's torch.randperm(n) (utils/training.py:1462)."""
    import torch
    for n in (1, 7, 32, 256):
        torch.manual_seed(123)
        a, after_a = torch.randperm(n), torch.rand(1)
        torch.manual_seed(123)
        b = torch.empty(n, dtype=torch.int64)
        torch.randperm(n, out=b)
        assert torch.equal(a, b) and torch.equal(after_a, torch.rand(1))


def test_register_carried_weight_stages_are_never_touched_while_in_flight(tmp_path):
    """csrc/vit_mega.hip, round 6: the loader waves of vit_fwd_mega2_kernel load weight stages into registers with inline-asm loads the
    compiler knows nothing about (a load it counts makes it drain the operand ring once per product).  What keeps that sound is a
    property of the GENERATED code: between such a load and the ds_write_b128 that consumes its register quad no instruction may
    read or write the quad (a copy or a spill would move bytes that have not landed).  tools/micro/check_inflight_regs.py walks the
    control-flow graph of the compiler's assembly; this compiles the file for gfx950 (no GPU needed) and runs it."""
    import shutil
    import subprocess
    import sys
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "vit_mega.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-Wno-unused-function",
                           "--cuda-device-only", "-S", "-o", str(out), os.path.join(root, "swat_amd", "csrc", "vit_mega.hip")],
                          stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "micro", "check_inflight_regs.py"), str(out), "mega2"],
                       stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "register-stage loads, 0 violations" in r.stdout and " 0 register-stage loads" not in r.stdout, r.stdout[-500:]
    shutil.rmtree(tmp_path, ignore_errors=True)
