"""GPU: the product's train_cutmix / train_ce / validate (HIP tower, fused step) against what the REFERENCE's own loops
produced on the same seeded inputs with the fp32 oracle tower (tests/golden/loop_tiny.npz, host_logic.json)."""
import io
import json
import os
import types

import numpy as np
import pytest
import torch

from oracle import vit_oracle as vo
from oracle.make_golden import TINY, loop_inputs

pytestmark = pytest.mark.gpu


class _Lg:
    handlers = []

    def info(self, *a):
        pass


def _args(**kw):
    a = types.SimpleNamespace(method="cutmix", device="cuda", freeze_visual=False, lr_classifier=1e-4, lr_backbone=1e-6,
                              temperature=0.07, optim="AdamW", wd=1e-2, epochs=2, early_stop=False, cutmix_beta=1.0,
                              mix_prob=0.5, loss_name="CE", pre_extracted=False, dataset="synthetic", output_dir="/tmp",
                              save_ckpt=False, save_freq=100, stop_epochs=-1, bsz=8, training_seed=1)
    a.__dict__.update(kw)
    return a


@pytest.mark.parametrize("method", ["cutmix", "finetune"])
def test_training_loop_reproduces_reference_loss_csv(golden_dir, method):
    from swat_amd import optimizers, training
    from swat_amd.clip_model import CLIP
    from swat_amd.models import MyLinear
    g = np.load(os.path.join(golden_dir, "loop_tiny.npz"), allow_pickle=False)
    B, C, nb, epochs = 8, 10, 4, 2
    torch.manual_seed(0)  # same construction order as oracle/make_golden.py golden_loop
    ref = vo.OracleCLIP(vision_cfg=TINY, text_width=64, text_heads=1, text_layers=1, ctx=8, vocab=64)
    W = torch.randn(C, TINY[5])
    W /= W.norm(dim=-1, keepdim=True)
    model = CLIP(vision_cfg=TINY, text_cfg=(8, 64, 64, 1, 1))
    model.load_state_dict(ref.state_dict())
    model.to("cuda")
    head = MyLinear(weights=W.clone()).to("cuda")
    args = _args(method=method, epochs=epochs)
    args.loss = torch.nn.CrossEntropyLoss()
    params, logit_scale = optimizers.set_params(args, model, head, _Lg())
    train, val, test = loop_inputs(11, nb, B, C), loop_inputs(12, 2, B, C), loop_inputs(13, 2, B, C)
    opt, sched, _ = optimizers.set_optimizer(args, params, train)
    args.optimizer, args.scheduler = opt, sched
    training.set_training_seed(args)
    csv = io.StringIO()
    if method == "cutmix":
        out = training.train_cutmix(args, _Lg(), csv, model, head, train, val, test)
    else:
        out = training.train_ce(args, _Lg(), csv, model, head, train, val, test, False)
    best_model, best_head, best_records, best_ls = out
    got = [[float(v) for v in line.split(",")] for line in csv.getvalue().strip().split("\n")]
    want = [[float(v) for v in line.split(",")] for line in str(g[f"{method}_csv"]).strip().split("\n")]
    assert len(got) == len(want) == epochs
    for a, b in zip(got, want):
        assert a[0] == b[0] and a[1] == b[1]                # epoch, iter
        assert abs(a[2] - b[2]) < 2e-2                      # train loss (stated bf16 tolerance)
        assert abs(a[3] - b[3]) < 5e-2                      # val loss (-1 before the last epoch)
        assert abs(a[4] - b[4]) <= 6.25 + 1e-9 and abs(a[5] - b[5]) <= 6.25 + 1e-9  # acc: at most one of 16 samples flips
    want_best = json.loads(str(g[f"{method}_best"]))
    assert best_records["best_epoch"] == want_best[1] and best_records["best_iter"] == want_best[2]
    assert best_records["best_confusion_matrix"].sum() == np.asarray(g[f"{method}_cm"]).sum()
    # trained head / logit_scale end up where the reference's did (AdamW, 3 LR groups, warm-up LR trace)
    ref_head = torch.from_numpy(g[f"{method}_head"])
    assert ((head.linear.weight.cpu() - ref_head).norm() / ref_head.norm()).item() < 1e-3
    assert abs(logit_scale.item() - float(g[f"{method}_ls"])) < 1e-3
    # the returned best model is a usable deep copy on its own arena
    with torch.no_grad():
        f = best_model.visual.forward_raw(train[0][0].to("cuda"))
    assert torch.isfinite(f).all()


def test_validate_on_pre_extracted_features_matches_reference(golden_dir):
    from swat_amd.models import MyLinear
    from swat_amd.testing import validate
    rec = json.load(open(os.path.join(golden_dir, "host_logic.json")))["validate"]
    g = torch.Generator().manual_seed(rec["seed"])
    C, E = rec["C"], rec["E"]
    W = torch.randn(C, E, generator=g)
    head = MyLinear(weights=W).to("cuda")
    feats = [torch.nn.functional.normalize(torch.randn(9, E, generator=g), dim=-1) for _ in range(4)]
    labels = [torch.randint(0, C, (9,), generator=g) for _ in range(4)]
    data = [(f, l, None, None) for f, l in zip(feats, labels)]
    acc, loss, cm = validate(_args(), data, torch.nn.Module(), _Lg(), None, torch.tensor([4.60517], device="cuda"), head,
                             show_confusion_matrix=True, device="cuda", pre_extracted=True)
    assert acc == rec["acc"]                       # fp32 head on exact inputs: identical predictions
    assert np.array_equal(cm, np.array(rec["cm"]))
    assert abs(loss - rec["loss"]) < 1e-4


def test_cli_end_to_end_on_synthetic_data(tmp_path, monkeypatch):
    """main.py: stage 1 (cutmix) -> WiSE-FT -> feature re-extraction -> tau-norm -> checkpoint -> stage-2 probing + FSFT,
    on a tiny synthetic dataset (ViT-B/32 weights random): the whole pipeline runs on the HIP path and prints the CSV line."""
    import sys
    monkeypatch.chdir(tmp_path)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import main as cli
    summary = cli.main(["--synthetic", "96", "--num_classes", "5", "--method", "cutmix", "--epochs", "1", "--bsz", "16",
                        "--num_workers", "0", "--log_mode", "file", "--shots", "4", "--folder", str(tmp_path / "out")])
    parts = summary.split(",")
    assert parts[0] == "synthetic" and parts[1] == "cutmix" and len(parts) == 11
    accs = [float(v) for v in parts[7:]]
    assert all(0.0 <= a <= 100.0 for a in accs)
    import glob
    csvs = glob.glob(str(tmp_path / "out" / "**" / "loss.csv"), recursive=True)
    assert len(csvs) == 1
    case_dir = os.path.dirname(csvs[0])
    files = os.listdir(case_dir)
    assert any(f.startswith("stage1_model_best") for f in files)
    out, case = type(tmp_path)(case_dir).parent, os.path.basename(case_dir)
    ck = torch.load(out / case / [f for f in files if f.startswith("stage1_model_best")][0], map_location="cpu", weights_only=False)
    assert {"clip", "head", "logit_scale", "test_acc", "best_tau", "wsft_test_acc", "wsft_backbone"} <= set(ck.keys())
    assert "visual.transformer.resblocks.11.mlp.c_proj.weight" in ck["clip"]


@pytest.mark.parametrize("cfg,width", [("vitl14_openclip_laion400m", 768), ("vitb16_openclip_laion400m", 512)])
def test_cli_other_model_cfgs_build_a_head_of_the_tower_width(tmp_path, monkeypatch, cfg, width):
    """BASELINE config 5 (ViT-L/14) and ViT-B/16 through main.py: `--synthetic` forces cls_init = random, whose head the
    reference hard-codes to 512 columns (utils/models.py:26,36) — here the head takes the tower's embedding width, so one
    stage-1 epoch (cutmix, WiSE-FT, tau-norm, checkpoint) runs on a 768-wide embedding too."""
    import glob
    import sys
    monkeypatch.chdir(tmp_path)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import main as cli
    summary = cli.main(["--synthetic", "16", "--num_classes", "3", "--method", "cutmix", "--epochs", "1", "--bsz", "8",
                        "--num_workers", "0", "--log_mode", "file", "--shots", "4", "--skip_stage2", "--model_cfg", cfg,
                        "--folder", str(tmp_path / "out")])
    parts = summary.split(",")
    assert parts[1] == "cutmix" and 0.0 <= float(parts[7]) <= 100.0
    ck = glob.glob(str(tmp_path / "out" / "**" / "stage1_model_best*"), recursive=True)
    assert len(ck) == 1
    st = torch.load(ck[0], map_location="cpu", weights_only=False)
    assert st["head"]["linear.weight"].shape == (3, width) and st["clip"]["visual.proj"].shape[1] == width


def test_cli_zeroshot_method_matches_the_oracle(tmp_path, monkeypatch, capsys):
    """BASELINE config 1 (`--method zeroshot`, ref main.py:219-227, utils/training.py:23-44): the CLI builds the model and the
    classifier head, validates on the test split through the HIP tower + fused head, prints the 8-field summary and exits.
    The accuracy and confusion matrix are compared with the fp32 oracle run on the same weights and the same loader."""
    import sys
    monkeypatch.chdir(tmp_path)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import main as cli
    from swat_amd import stages
    seen = {}
    real = stages.run_zeroshot

    def spy(args, test_loader, model, logger, loss, logit_scale, classifier_head):
        seen.update(loader=test_loader, model=model, head=classifier_head, logit_scale=logit_scale)
        seen["acc"] = real(args, test_loader, model, logger, loss, logit_scale, classifier_head)
        return seen["acc"]

    monkeypatch.setattr(stages, "run_zeroshot", spy)
    with pytest.raises(SystemExit) as ex:
        cli.main(["--synthetic", "48", "--num_classes", "5", "--method", "zeroshot", "--bsz", "16", "--num_workers", "0",
                  "--log_mode", "file", "--shots", "4", "--folder", str(tmp_path / "out")])
    assert ex.value.code == 0
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("synthetic,zeroshot,")][-1]
    parts = line.split(",")
    assert len(parts) == 8 and float(parts[7]) == round(seen["acc"], 1)
    # the same pass on the CPU oracle
    ref = vo.OracleCLIP("ViT-B-32-quickgelu", with_text=False)
    sd = {k: v.detach().cpu() for k, v in seen["model"].state_dict().items() if k.startswith("visual.") or k == "logit_scale"}
    ref.load_state_dict(sd)
    W = seen["head"].linear.weight.detach().cpu()
    ls = seen["logit_scale"].detach().cpu()
    right = total = 0
    flips = 0
    with torch.no_grad():
        for inputs, labels, _, _ in seen["loader"]:
            f = ref.encode_image(inputs.float())
            logits = (f / f.norm(dim=-1, keepdim=True)) @ W.t() * ls.exp()
            top2 = logits.topk(2, dim=1).values
            flips += int(((top2[:, 0] - top2[:, 1]) < 2e-2 * logits.abs().max()).sum())  # near-ties bf16 may flip
            right += int((logits.argmax(1) == labels).sum())
            total += labels.numel()
    want = 100.0 * right / total
    assert abs(seen["acc"] - want) <= 100.0 * flips / total + 1e-9, (seen["acc"], want, flips)


def _hip_clip_from(oracle_model):
    from swat_amd.clip_model import CLIP
    m = CLIP(vision_cfg=TINY, text_cfg=(8, 64, 64, 1, 1))
    m.load_state_dict(oracle_model.state_dict())
    return m.to("cuda")


def test_stage_fan_out_matches_the_reference_golden(golden_dir):
    """A10 / A11 / f.4 on the GPU: extract_test_feats, ensemble_model + run_wsft_alpha and run_tau_normalization of
    swat_amd (HIP tower + fused head behind validate) against the outputs of the REFERENCE's own functions
    (utils/features.py:122-148, main.py:27-195) recorded on the fp32 oracle by oracle/make_golden.py `stages`."""
    from oracle import make_golden as mg
    from swat_amd import stages
    from swat_amd.features import extract_test_feats
    from swat_amd.models import MyLinear
    g = np.load(f"{golden_dir}/stages_tiny.npz")
    (best_o, Wb), (zero_o, Wz) = mg.stage_models()
    best, zero = _hip_clip_from(best_o), _hip_clip_from(zero_o)
    # extract_test_feats: loader order, labels, L2-normalised embeddings
    feats = extract_test_feats(best, mg.stage_loader(21, 22, 8))
    assert np.array_equal(feats["labels"].numpy(), g["feats_labels"])
    want = g["feats_image_features"]
    assert feats["image_features"].shape == want.shape and not feats["image_features"].is_cuda
    assert np.abs(feats["image_features"].numpy() - want).max() <= 0.007 * np.abs(want).max()
    assert np.allclose(np.linalg.norm(feats["image_features"].numpy(), axis=1), 1.0, atol=1e-5)
    # WiSE-FT sweep (alpha = 0.5): interpolated tower + head, validate on val / test
    args = types.SimpleNamespace(device="cuda", dataset="synthetic", output_dir="/tmp", pre_extracted=False, freeze_visual=False,
                                 tau_norm=True, loss=torch.nn.CrossEntropyLoss())
    ls = torch.tensor(4.0, device="cuda")
    Wn = Wb / Wb.norm(dim=-1, keepdim=True)
    val, test = mg.stage_loader(31, 40, 8, teacher=(best_o, Wn)), mg.stage_loader(32, 48, 8, teacher=(best_o, Wn))
    head_b, head_z = MyLinear(weights=Wb.clone()).to("cuda"), MyLinear(weights=Wz.clone()).to("cuda")
    wm, wh, wacc = stages.run_wsft_alpha(args, best, head_b, val, test, zero, head_z, ls, _Lg())
    assert np.allclose(wh.linear.weight.detach().cpu().numpy(), g["wsft_alpha_head"], atol=1e-6)
    assert abs(wacc - float(g["wsft_alpha_test_acc"])) <= 100.0 / 48 + 1e-9     # bf16 tower: at most one near-tie flips
    # the interpolated tower itself: embeddings of the batch the reference's ensemble_model(0.5) tower was recorded on
    with torch.no_grad():
        e = wm.visual.forward_raw(mg.stage_loader(21, 22, 8)[0][0].to("cuda")).cpu().numpy()
    want_e = g["wsft_0.5_embeds"]
    assert e.shape == want_e.shape and np.abs(e - want_e).max() <= 0.007 * np.abs(want_e).max()
    # tau-normalisation on pre-extracted features of the same splits
    fv, ft = extract_test_feats(best, val), extract_test_feats(best, test)
    pv = [(fv["image_features"][i:i + 8], fv["labels"][i:i + 8], torch.zeros(1), 1) for i in range(0, 40, 8)]
    pt = [(ft["image_features"][i:i + 8], ft["labels"][i:i + 8], torch.zeros(1), 1) for i in range(0, 48, 8)]
    th, tau, tacc = stages.run_tau_normalization(args, head_b, best, pv, pt, ls, _Lg())
    if abs(tau - float(g["tau_best"])) < 1e-9:
        assert np.allclose(th.linear.weight.detach().cpu().numpy(), g["tau_head"], rtol=1e-5, atol=1e-6)
    assert abs(tacc - float(g["tau_test_acc"])) <= 2 * 100.0 / 48 + 1e-9


def test_train_probing_matches_the_reference_golden(golden_dir):
    """A8 / f.4: swat_amd.training.train_probing (fused head step + device-side validate) on pre-extracted features against the
    REFERENCE's own train_probing (utils/training.py:47-161 — the loop main.py:402-474 run_stage2_probing drives), recorded by
    oracle/make_golden.py `probing`: loss.csv rows, the trained head, best records and the confusion matrix.  The head path is
    fp32 end to end, so the tolerance is rounding of a different summation order, not bf16."""
    from oracle.make_golden import probing_inputs
    from swat_amd import optimizers, training
    from swat_amd.clip_model import CLIP
    from swat_amd.models import MyLinear
    g = np.load(os.path.join(golden_dir, "probing_tiny.npz"), allow_pickle=False)
    train, val, test, W0 = probing_inputs()
    torch.manual_seed(0)
    model = CLIP(vision_cfg=TINY, text_cfg=(8, 64, 64, 1, 1)).to("cuda")
    head = MyLinear(weights=W0.clone()).to("cuda")
    args = _args(method="probing", epochs=3, pre_extracted=True, lr_classifier=1e-2, early_stop=False)
    args.loss = torch.nn.CrossEntropyLoss()
    params, logit_scale = optimizers.set_params(args, model, head, _Lg())
    assert abs(float(logit_scale.reshape(-1)[0]) - 4.60517) < 1e-6 and not logit_scale.requires_grad   # fixed ln(100), ref optimizers.py:58
    opt, sched, _ = optimizers.set_optimizer(args, params, train)
    args.optimizer, args.scheduler = opt, sched
    csv = io.StringIO()
    best_model, best_head, rec, best_ls, vl, tl = training.train_probing(args, _Lg(), csv, model, head, None, None, train, val, test, False)
    assert vl is val and tl is test
    got = [[float(v) for v in line.split(",")] for line in csv.getvalue().strip().split("\n")]
    want = [[float(v) for v in line.split(",")] for line in str(g["csv"]).strip().split("\n")]
    assert len(got) == len(want) == 3
    for a, b in zip(got, want):
        assert a[0] == b[0] and a[1] == b[1]
        assert abs(a[2] - b[2]) < 2e-4 * max(1.0, abs(b[2])) and abs(a[3] - b[3]) < 2e-4 * max(1.0, abs(b[3]))   # train / val loss
        assert a[4] == b[4] and a[5] == b[5]                                                              # val / test accuracy
    ref_head = torch.from_numpy(g["head"])
    assert ((head.linear.weight.cpu() - ref_head).norm() / ref_head.norm()).item() < 1e-5
    assert ((best_head.linear.weight.cpu() - torch.from_numpy(g["best_head"])).norm() / ref_head.norm()).item() < 1e-5
    wb = json.loads(str(g["best"]))
    assert rec["best_val_acc"] == wb[0] and rec["best_epoch"] == wb[1] and rec["best_iter"] == wb[2]
    assert np.array_equal(np.asarray(rec["best_confusion_matrix"]), g["cm"])
    assert abs(rec["best_scores"]["avg_class_accuracy"] - float(g["avg_class_acc"])) < 1e-9


def test_get_text_features_and_cal_prompt_tensors_match_the_reference_golden(golden_dir, tmp_path, monkeypatch):
    """B8 compute branch: swat_amd.features.get_text_features (HIP text tower: encode every template x synonym, L2-normalise,
    mean, L2-normalise) and the UN-CACHED branch of cal_prompt_tensors (prompt_maker -> get_text_features -> both tables
    saved, then the cached branch on the second call) against the reference's own functions (utils/features.py:39-64,
    retrieval/sample_retrieval.py:1418-1454) recorded on the fp32 oracle by oracle/make_golden.py `prompt_tensors`."""
    from oracle import synth
    from oracle.make_golden import PROMPT_TEXT_CFG, prompt_metrics, prompt_tensor_model
    from swat_amd import features
    from swat_amd.clip_model import CLIP
    from swat_amd.prompt import prompt_maker
    from swat_amd.retrieval.sample_retrieval import cal_prompt_tensors
    g = np.load(os.path.join(golden_dir, "prompt_tensors.npz"), allow_pickle=False)
    model = CLIP(vision_cfg=TINY, text_cfg=PROMPT_TEXT_CFG)
    model.load_state_dict(prompt_tensor_model().state_dict())
    model.to("cuda").eval()
    tok = synth.hash_tokenize(PROMPT_TEXT_CFG[0], PROMPT_TEXT_CFG[1])
    metrics = prompt_metrics()

    def close(a, b):  # unit vectors through a bf16 tower: the stated path-A tolerance (0.7 % of max-abs), and still unit-norm
        a = a.detach().float().cpu().numpy()
        assert a.shape == b.shape and np.abs(a - b).max() <= 0.007 * np.abs(b).max() + 2e-3, np.abs(a - b).max()
        assert np.allclose(np.linalg.norm(a, axis=-1), 1.0, atol=1e-4)

    for ds, nt in (("semi-aves", "alternates"), ("dtd", "name")):
        td = features.get_text_features(model, prompt_maker(metrics, ds, nt), tok, "encode")
        keys = json.loads(str(g[f"{ds}|{nt}|keys"]))
        assert list(td.keys()) == keys
        assert [td[k]["all"].shape[0] for k in keys] == g[f"{ds}|{nt}|counts"].tolist()
        close(torch.cat([td[k]["all"] for k in keys]), g[f"{ds}|{nt}|all"])
        close(torch.stack([td[k]["mean"] for k in keys]), g[f"{ds}|{nt}|mean"])
    # the command line's branch: no cache on disk -> compute + save; second call -> load
    monkeypatch.chdir(tmp_path)
    args = types.SimpleNamespace(dataset="semi-aves", model_cfg="vitb32_openclip_laion400m", prompt_name="alternates", recal_prompt=False)
    ptd, tpd = cal_prompt_tensors(args, _Lg(), model, tok, metrics, "/unused", data_dir=str(tmp_path / "data"))
    keys = json.loads(str(g["cal|keys"]))
    assert list(ptd["alternates"].keys()) == keys
    close(torch.stack([ptd["alternates"][k]["mean"] for k in keys]), g["cal|mean"])
    assert sorted(os.listdir(tmp_path / "data" / "semi-aves" / "prompts")) == json.loads(str(g["cal|files"]))
    assert tpd["alternates"][keys[0]]["corpus"] == json.loads(str(g["cal|corpus0"]))
    ptd2, _ = cal_prompt_tensors(args, _Lg(), None, None, metrics, "/unused", data_dir=str(tmp_path / "data"))   # cached: no model needed
    assert torch.equal(ptd2["alternates"][keys[0]]["mean"].cpu(), ptd["alternates"][keys[0]]["mean"].cpu())
