"""oracle/make_golden.py — TEST INFRASTRUCTURE ONLY (runs in the build container only).

Imports the reference (tian1327/SWAT at /root/reference, read-only) through oracle/ref_shims.py,
runs the reference's OWN functions on seeded synthetic inputs and writes their outputs as small
golden fixtures under tests/golden/.  Inputs are never stored: every fixture records the seed
and the generator below (`synth_*`) is re-run by the tests (numpy's legacy RandomState stream is
frozen, so the inputs are reproducible bit for bit).

    python oracle/make_golden.py [retrieval] [host] [loop] [vit] [text] [stages] ...     (no arguments: every section)

The reference source never travels: fixtures are plain .npz/.json data.
"""
import io
import json
import logging
import os
import pickle
import sys
import tempfile
import types
from collections import defaultdict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.environ.get("SWAT_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # (tests regenerate into a scratch directory)
sys.path.insert(0, ROOT)

from oracle import synth  # noqa: E402  (shared seeded input generators, also used by tests)


def _logger():
    lg = logging.getLogger("golden")
    if not lg.handlers:
        lg.addHandler(logging.StreamHandler(io.StringIO()))
    lg.setLevel(logging.INFO)
    return lg


# --------------------------------------------------------------------------------------------
# Path B — retrieval
# --------------------------------------------------------------------------------------------
def _run_ref_samplers(sr, case, k, image_dedup):
    """Drive the reference's samplers on a synthetic mined.pth-shaped dict."""
    mined = {
        "image_features": torch.from_numpy(case["img"]),
        "caption_features": torch.from_numpy(case["cap"]),
        "labels": torch.from_numpy(case["labels"]),
        "filepath": case["paths"],
    }
    pre = sr.transform_extracted_fea(mined)  # :1387-1415
    prompt_tensors = {str(c): {"mean": torch.from_numpy(case["queries"][i])}
                      for i, c in enumerate(case["class_ids"])}
    tmp = tempfile.mkdtemp()
    cap_map = defaultdict(dict)
    for p in case["paths"]:
        cap_map[p.split("/")[-2]][p.split("/")[-1].split(".")[0]] = "caption"
    cap_path = os.path.join(tmp, "cap.map")
    with open(cap_path, "wb") as f:
        pickle.dump(dict(cap_map), f)
    sr.CAPTION_MAP_DICT["synthetic"] = cap_path
    args = types.SimpleNamespace(dataset="synthetic", output_folder=tmp, prefix="golden")
    lg = _logger()
    sr.logger = lg
    sr.shutil = types.SimpleNamespace(copy=lambda *a, **k: None)  # the copy into ../data/ is not part of the result

    dup = defaultdict(set)
    filt = defaultdict(set)
    for p, e in zip(case["paths"], case["excl_dup"]):
        if e:
            dup[p.split("/")[-2]].add(p)
    for p, e in zip(case["paths"], case["excl_filt"]):
        if e:
            filt[p.split("/")[-2]].add(p)

    out = {}
    path_to_row = {p: i for i, p in enumerate(case["paths"])}
    for name, fn in [("t2t", sr.t2t_ranked_sampler), ("t2t_t2i", sr.t2t_ranked_t2i_tshd_sampler),
                     ("t2i", sr.t2i_ranked_sampler)]:
        split, counts = fn(args, lg, prompt_tensors, k, 0.0, pre, dup, filt)
        files = [p for fl in split["file_list"] for p in fl]
        labels = torch.cat(split["label_list"]) if split["label_list"] else torch.zeros(0, dtype=torch.int64)
        feats = torch.cat(split["feature_list"]) if split["feature_list"] else torch.zeros(0, case["img"].shape[1])
        rows = np.array([path_to_row[p] for p in files], np.int64)
        assert torch.equal(feats, torch.from_numpy(case["img"][rows]))
        sr.save_sample_file_list(args, files, labels)  # :1457-1469
        with open(os.path.join(tmp, "golden.txt")) as f:
            txt = f.read()
        out[name] = {"rows": rows, "labels": labels.numpy().astype(np.int64),
                     "counts": {c: int(v) for c, v in counts.items()}, "txt": txt}
    # raw similarities of the reference (torch CPU matmul) per original row
    sims_t2t = np.zeros(len(case["paths"]), np.float32)
    sims_t2i = np.zeros(len(case["paths"]), np.float32)
    for i, c in enumerate(case["class_ids"]):
        rows = np.nonzero(case["labels"] == c)[0]
        q = torch.from_numpy(case["queries"][i]).unsqueeze(0)
        sims_t2t[rows] = np.array(sr.t2t_similarity(q, pre[str(c)]["caption_feats"]), np.float32)
        sims_t2i[rows] = np.array(sr.cal_t2i_similarity(q, pre[str(c)]["feats"]), np.float32)
    out["sims_t2t"], out["sims_t2i"] = sims_t2t, sims_t2i
    # group-by result (class visiting order + row order inside a class)
    classes = sorted(pre.keys(), key=lambda x: int(x))
    out["group_order"] = np.array([path_to_row[p] for c in classes for p in pre[c]["file_paths"]], np.int64)
    out["group_classes"] = np.array([int(c) for c in classes], np.int64)
    out["group_sizes"] = np.array([len(pre[c]["file_paths"]) for c in classes], np.int64)
    if image_dedup:
        d, frac, avg = sr.remove_near_duplicates2(pre)  # :237-275
        out["dedup_rows"] = np.array(sorted(path_to_row[p] for c in d for p in d[c]), np.int64)
        out["dedup_frac"] = np.array(frac, np.float64)
    # p2p similarity (:369-394) on the first class
    c0 = classes[0]
    fs = [case["img"][r] for r in np.nonzero(case["labels"] == int(c0))[0][:16]]
    for mode in ("min", "max", "mean"):
        out[f"p2p_{mode}"] = np.array(sr.i2i_similarity_p2p(fs, pre[c0]["feats"], mode), np.float32)
    return out


def golden_retrieval():
    from oracle import ref_shims, retrieval_oracle as orc
    sr = ref_shims.import_retrieval()
    for kind, spec in synth.RETRIEVAL_CASES.items():
        case = synth.retrieval_case(**spec)
        ref = _run_ref_samplers(sr, case, spec["k"], image_dedup=True)
        # --- cross-check the C oracle against the reference before committing anything ---
        order, cls, off = orc.group_by_label(case["labels"])
        assert np.array_equal(order, ref["group_order"]) and np.array_equal(cls, ref["group_classes"])
        cap_g, img_g = case["cap"][order], case["img"][order]
        excl = (case["excl_dup"] | case["excl_filt"])[order].astype(np.uint8)
        for name, kw in [("t2t", dict(rank_by="t2t", use_t2i_thr=False)),
                         ("t2t_t2i", dict(rank_by="t2t", use_t2i_thr=True)),
                         ("t2i", dict(rank_by="t2i"))]:
            idx, cnt = orc.sample_index(cap_g, img_g, case["queries"], off, spec["k"], excluded=excl, **kw)
            rows = np.concatenate([order[idx[s, :cnt[s]]] for s in range(len(cls))]) if len(cls) else np.zeros(0, np.int64)
            same = np.array_equal(rows, ref[name]["rows"])
            print(f"[retrieval/{kind}/{name}] kept={len(rows)} oracle==reference: {same}")
            assert same, "oracle disagrees with the reference on this fixture"
        payload = {
            "spec": json.dumps(spec),
            "sims_t2t": ref["sims_t2t"], "sims_t2i": ref["sims_t2i"],
            "group_order": ref["group_order"], "group_classes": ref["group_classes"],
            "group_sizes": ref["group_sizes"], "dedup_rows": ref["dedup_rows"], "dedup_frac": ref["dedup_frac"],
            "p2p_min": ref["p2p_min"], "p2p_max": ref["p2p_max"], "p2p_mean": ref["p2p_mean"],
        }
        for name in ("t2t", "t2t_t2i", "t2i"):
            payload[f"{name}_rows"] = ref[name]["rows"]
            payload[f"{name}_labels"] = ref[name]["labels"]
            payload[f"{name}_counts"] = json.dumps(ref[name]["counts"])
            payload[f"{name}_txt"] = ref[name]["txt"]
        np.savez_compressed(os.path.join(GOLD, f"retrieval_{kind}.npz"), **payload)
        print(f"wrote tests/golden/retrieval_{kind}.npz")


def golden_retrieval_cli():
    """B7 / B8 driver level: the reference's own sampling() (sample_retrieval.py:1471-1670) on a mined.pth FILE, with and
    without --image_dedup, and zeroshot_clip_img_filter (:278-329) on the same index; records <prefix>.txt,
    <prefix>_num_imgs_sampled.json and the filtered-path sets."""
    from oracle import ref_shims
    sr = ref_shims.import_retrieval()
    import utils.models as rm
    payload = {}
    for kind in ("float", "grid"):
        spec = synth.RETRIEVAL_CASES[kind]
        case = synth.retrieval_case(**spec)
        root = tempfile.mkdtemp()
        os.makedirs(f"{root}/synthetic")
        torch.save({"image_features": torch.from_numpy(case["img"]), "caption_features": torch.from_numpy(case["cap"]),
                    "labels": torch.from_numpy(case["labels"]), "filepath": case["paths"]},
                   f"{root}/synthetic/synthetic_vitb32_openclip_laion400m_mined.pth")
        cap_map = defaultdict(dict)
        for p in case["paths"]:
            cap_map[p.split("/")[-2]][p.split("/")[-1].split(".")[0]] = "caption"
        with open(f"{root}/cap.map", "wb") as f:
            pickle.dump(dict(cap_map), f)
        sr.CAPTION_MAP_DICT["synthetic"] = f"{root}/cap.map"
        sr.prompt_tensors_dict = {"alternates": {str(c): {"mean": torch.from_numpy(case["queries"][i])}
                                                 for i, c in enumerate(case["class_ids"])}}
        sr.logger = _logger()
        sr.shutil = types.SimpleNamespace(copy=lambda *a, **k: None)
        sr.device = "cpu"
        for method, tag in (("T2T-rank-T2I-tshd", "t2t_t2i"), ("T2T-rank", "t2t"), ("T2I-rank", "t2i")):
            for dedup in (False, True):
                out = tempfile.mkdtemp()
                args = types.SimpleNamespace(dataset="synthetic", model_cfg="vitb32_openclip_laion400m", prefix="golden",
                                             output_folder=out, sampling_method=method, num_samples=spec["k"], image_dedup=dedup,
                                             zeroshot_img_filter=False, prompt_name="alternates", sampling_threshold=0.0)
                path, ct = sr.sampling(args, sr.logger, None, None, None, f"{root}/synthetic")
                key = f"{kind}_{tag}_{'dedup' if dedup else 'plain'}"
                payload[key + "_txt"] = open(path).read()
                payload[key + "_counts"] = open(f"{out}/golden_num_imgs_sampled.json").read()
                payload[key + "_ct"] = np.int64(ct)
                print(f"[retrieval_cli/{key}] sample_ct {ct}")
        # zero-shot filter: head = class prompt means; positions-vs-file-id matching rule of the reference
        pre = sr.transform_extracted_fea(torch.load(f"{root}/synthetic/synthetic_vitb32_openclip_laion400m_mined.pth"))
        C = max(case["class_ids"]) + 1
        W = torch.zeros(C, case["queries"].shape[1])
        for i, c in enumerate(case["class_ids"]):
            W[c] = torch.from_numpy(case["queries"][i])
        head = rm.MyLinear(weights=W)
        os.makedirs(f"{root}/folders", exist_ok=True)
        for c in case["class_ids"]:
            os.makedirs(f"{root}/folders/{c}", exist_ok=True)
        filt = sr.zeroshot_clip_img_filter(None, None, f"{root}/folders", pre_extracted_feats=pre, head=head)
        payload[f"{kind}_zsfilter"] = json.dumps({c: sorted(v) for c, v in filt.items()})
        print(f"[retrieval_cli/{kind}] zero-shot filter keeps out {sum(len(v) for v in filt.values())} files")
    np.savez_compressed(os.path.join(GOLD, "retrieval_cli.npz"), **payload)
    print("wrote tests/golden/retrieval_cli.npz")


SECTIONS = {"retrieval": golden_retrieval, "retrieval_cli": golden_retrieval_cli}


# --------------------------------------------------------------------------------------------
# Path A — host logic the reference owns (LR trace, rand_bbox, CutMix transcript, mixup_criterion,
# calculate_scores / validate) and the loops themselves
# --------------------------------------------------------------------------------------------
def _ref_args(**kw):
    a = types.SimpleNamespace(method="cutmix", device="cpu", freeze_visual=False, lr_classifier=1e-4, lr_backbone=1e-6,
                              temperature=0.07, optim="AdamW", wd=1e-2, epochs=2, early_stop=False, cutmix_beta=1.0,
                              mix_prob=0.5, loss_name="CE", pre_extracted=False, dataset="synthetic", output_dir="/tmp",
                              save_ckpt=False, save_freq=100, stop_epochs=-1, bsz=8)
    a.__dict__.update(kw)
    return a


def golden_host():
    from oracle import ref_shims
    ref_shims.install()
    import utils.training as rt
    import utils.optimizers as ro
    import testing as rtest
    out = {}

    # G1: LR trace of set_optimizer (utils/optimizers.py:22-35) for 3 groups, two T_max values
    traces = {}
    for total_iter_loader, epochs, lrb in [(100, 2, 1e-6), (1388, 50, 1e-6), (20, 3, 1e-4)]:
        ps = [torch.nn.Parameter(torch.zeros(2)) for _ in range(3)]
        params = [{"params": [ps[0]], "lr": 1e-4}, {"params": [ps[1]], "lr": lrb}, {"params": [ps[2]], "lr": 1e-4}]
        args = _ref_args(epochs=epochs, lr_backbone=lrb)
        opt, sched, total = ro.set_optimizer(args, params, [None] * total_iter_loader)
        tr = []
        for it in range(300):
            tr.append([g["lr"] for g in opt.param_groups])
            opt.step()
            sched.step()
        traces[f"{total_iter_loader}x{epochs}_{lrb}"] = {"total_iter": total, "lrs": tr}
    out["lr_trace"] = traces

    # G2: rand_bbox (utils/training.py:1404-1420)
    boxes = []
    rs = np.random.RandomState(0)
    for i in range(300):
        lam = float(rs.beta(1.0, 1.0))
        np.random.seed(1000 + i)
        b = rt.rand_bbox((32, 3, 224, 224), lam)
        boxes.append([lam] + [int(v) for v in b])
    out["rand_bbox"] = boxes

    # G4: mixup_criterion over nn.CrossEntropyLoss (utils/training.py:1157-1158, utils/losses.py:10-11)
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(32, 200, generator=g) * 3
    ya = torch.randint(0, 200, (32,), generator=g)
    yb = torch.randint(0, 200, (32,), generator=g)
    ce = torch.nn.CrossEntropyLoss()
    out["mixup_criterion"] = {"seed": 5, "vals": [[lam, float(rt.mixup_criterion(ce, logits, ya, yb, lam))]
                                                  for lam in (1.0, 0.0, 0.3, 0.7312)]}

    # G6: calculate_scores + validate on pre-extracted features (testing.py:134-186, 403-486)
    g = torch.Generator().manual_seed(6)
    C, E = 7, 16
    W = torch.randn(C, E, generator=g)
    head = rt.build_classifier_head.__globals__["MyLinear"](weights=W)
    feats = [torch.nn.functional.normalize(torch.randn(9, E, generator=g), dim=-1) for _ in range(4)]
    labels = [torch.randint(0, C, (9,), generator=g) for _ in range(4)]
    data = [(f, l, None, None) for f, l in zip(feats, labels)]

    class _M(torch.nn.Module):
        pass

    acc, loss, cm = rtest.validate(_ref_args(), data, _M(), _logger(), torch.nn.CrossEntropyLoss(), torch.tensor([4.60517]),
                                   head, show_confusion_matrix=True, device="cpu", pre_extracted=True)
    scores = rtest.calculate_scores(cm)
    out["validate"] = {"seed": 6, "C": C, "E": E, "acc": acc, "loss": float(loss), "cm": cm.tolist(),
                       "scores": {"acc": float(scores["acc"]), "avg_class_accuracy": float(scores["avg_class_accuracy"]),
                                  "recall": [float(scores["per_class_recall"][i]) for i in range(C)],
                                  "precision": [float(scores["per_class_precision"][i]) for i in range(C)],
                                  "f1": [float(scores["per_class_f1score"][i]) for i in range(C)]}}
    with open(os.path.join(GOLD, "host_logic.json"), "w") as f:
        json.dump(out, f)
    print("wrote tests/golden/host_logic.json")


TINY = (64, 16, 128, 2, 2, 64, "quick_gelu")


def loop_inputs(seed, n_batches, B, C, img=64):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(B, 3, img, img, generator=g), torch.randint(0, C, (B,), generator=g), None,
             torch.ones(B, dtype=torch.long)) for _ in range(n_batches)]


def golden_loop():
    """G3 + G5: the reference's own train_cutmix / train_ce (utils/training.py:1422-1556, 296-412) driven on CPU with the
    fp32 oracle tower as `model`; records loss.csv, the CutMix transcript (lam', target_b, pasted box decoded from the
    images the model received) and the final head / logit_scale."""
    from oracle import ref_shims, vit_oracle as vo
    ref_shims.install()
    import utils.training as rt
    import utils.optimizers as ro
    import utils.models as rm
    res = {}
    for method in ("cutmix", "finetune"):
        B, C, nb, epochs = 8, 10, 4, 2
        torch.manual_seed(0)
        model = vo.OracleCLIP(vision_cfg=TINY, text_width=64, text_heads=1, text_layers=1, ctx=8, vocab=64)
        W = torch.randn(C, TINY[5])
        W /= W.norm(dim=-1, keepdim=True)
        head = rm.MyLinear(weights=W.clone())
        args = _ref_args(method=method, epochs=epochs, bsz=B)
        args.loss = torch.nn.CrossEntropyLoss()
        params, logit_scale = ro.set_params(args, model, head, _logger())
        train = loop_inputs(11, nb, B, C)
        val = loop_inputs(12, 2, B, C)
        test = loop_inputs(13, 2, B, C)
        opt, sched, _ = ro.set_optimizer(args, params, train)
        args.optimizer, args.scheduler = opt, sched
        transcript = []
        orig_mix = rt.mixup_criterion

        def rec_mix(criterion, pred, y_a, y_b, lam):
            transcript.append({"lam": float(lam), "y_b": y_b.tolist()})
            return orig_mix(criterion, pred, y_a, y_b, lam)

        rt.mixup_criterion = rec_mix
        rt.set_training_seed(types.SimpleNamespace(training_seed=1))
        csv = io.StringIO()
        # the reference pastes IN PLACE into the tensors its loader yields; a real DataLoader yields fresh tensors every
        # epoch, so hand it a loader that clones on every iteration
        class Fresh:
            def __len__(self):
                return len(train)

            def __iter__(self):
                for x, y, t, s in train:
                    yield x.clone(), y.clone(), t, s.clone()

        train_c = Fresh()
        fn = rt.train_cutmix if method == "cutmix" else rt.train_ce
        extra = () if method == "cutmix" else (False,)
        best_model, best_head, best_records, best_ls = fn(args, _logger(), csv, model, head, train_c, val, test, *extra)
        rt.mixup_criterion = orig_mix
        res[method] = {"loss_csv": csv.getvalue(), "transcript": transcript,
                       "head": head.linear.weight.detach().numpy().astype(np.float32),
                       "logit_scale": float(logit_scale.item()),
                       "best_val_acc": float(best_records["best_val_acc"]), "best_epoch": int(best_records["best_epoch"]),
                       "best_iter": int(best_records["best_iter"]),
                       "cm": np.asarray(best_records["best_confusion_matrix"])}
        print(f"[loop/{method}] loss.csv:\n{csv.getvalue()}")
    np.savez_compressed(os.path.join(GOLD, "loop_tiny.npz"),
                        cutmix_csv=res["cutmix"]["loss_csv"], finetune_csv=res["finetune"]["loss_csv"],
                        cutmix_transcript=json.dumps(res["cutmix"]["transcript"]),
                        cutmix_head=res["cutmix"]["head"], finetune_head=res["finetune"]["head"],
                        cutmix_ls=res["cutmix"]["logit_scale"], finetune_ls=res["finetune"]["logit_scale"],
                        cutmix_best=json.dumps([res["cutmix"][k] for k in ("best_val_acc", "best_epoch", "best_iter")]),
                        finetune_best=json.dumps([res["finetune"][k] for k in ("best_val_acc", "best_epoch", "best_iter")]),
                        cutmix_cm=res["cutmix"]["cm"], finetune_cm=res["finetune"]["cm"])
    print("wrote tests/golden/loop_tiny.npz")


def probing_inputs(seed=41, C=7, E=64, B=8):
    """Pre-extracted (unit-norm) features of a 7-class toy problem: 4 train / 2 val / 2 test batches of 8."""
    rs = np.random.RandomState(seed)
    centres = synth.unit_rows(rs, C, E)

    def split(nb):
        out = []
        for _ in range(nb):
            y = rs.randint(0, C, size=B)
            x = 0.55 * centres[y] + synth.unit_rows(rs, B, E)
            x /= np.linalg.norm(x, axis=1, keepdims=True)
            out.append((torch.from_numpy(x.astype(np.float32)), torch.from_numpy(y.astype(np.int64)), None, torch.ones(B, dtype=torch.long)))
        return out
    tr, va, te = split(4), split(2), split(2)
    W0 = 0.35 * centres + synth.unit_rows(rs, C, E)   # a head that is partly right: accuracy moves while probing
    W0 /= np.linalg.norm(W0, axis=1, keepdims=True)
    return tr, va, te, torch.from_numpy(W0.astype(np.float32))


def golden_probing():
    """A8 / f.4: the reference's own train_probing (utils/training.py:47-161) on pre-extracted features — the loop that
    run_stage2_probing (main.py:402-474) drives after re-extracting the few-shot features: head-only AdamW with the
    warm-up + cosine LR trace, logit_scale fixed at ln(100), validate + calculate_scores every epoch."""
    from oracle import ref_shims, vit_oracle as vo
    ref_shims.install()
    import utils.training as rt
    import utils.optimizers as ro
    import utils.models as rm
    train, val, test, W0 = probing_inputs()
    torch.manual_seed(0)
    model = vo.OracleCLIP(vision_cfg=TINY, text_width=64, text_heads=1, text_layers=1, ctx=8, vocab=64)
    head = rm.MyLinear(weights=W0.clone())
    args = _ref_args(method="probing", epochs=3, pre_extracted=True, lr_classifier=1e-2, early_stop=False)
    args.loss = torch.nn.CrossEntropyLoss()
    params, logit_scale = ro.set_params(args, model, head, _logger())
    opt, sched, _ = ro.set_optimizer(args, params, train)
    args.optimizer, args.scheduler = opt, sched
    rt.time = types.SimpleNamespace(sleep=lambda s: None, time=__import__("time").time)
    csv = io.StringIO()
    best_model, best_head, rec, best_ls, _, _ = rt.train_probing(args, _logger(), csv, model, head, None, None, train, val, test, False)
    print(f"[probing] loss.csv:\n{csv.getvalue()}")
    np.savez_compressed(os.path.join(GOLD, "probing_tiny.npz"), csv=csv.getvalue(), head=head.linear.weight.detach().numpy(),
                        best_head=best_head.linear.weight.detach().numpy(), logit_scale=np.float32(logit_scale.item()),
                        best=json.dumps([float(rec["best_val_acc"]), int(rec["best_epoch"]), int(rec["best_iter"])]),
                        cm=np.asarray(rec["best_confusion_matrix"]),
                        avg_class_acc=np.float64(rec["best_scores"]["avg_class_accuracy"]))
    print("wrote tests/golden/probing_tiny.npz")


def hf_to_oracle_state(hf):
    """HF CLIPVisionModelWithProjection -> open_clip VisionTransformer key names (SURVEY App. A mapping)."""
    sd = hf.state_dict()
    out = {"conv1.weight": sd["vision_model.embeddings.patch_embedding.weight"],
           "class_embedding": sd["vision_model.embeddings.class_embedding"],
           "positional_embedding": sd["vision_model.embeddings.position_embedding.weight"],
           "ln_pre.weight": sd["vision_model.pre_layrnorm.weight"], "ln_pre.bias": sd["vision_model.pre_layrnorm.bias"],
           "ln_post.weight": sd["vision_model.post_layernorm.weight"], "ln_post.bias": sd["vision_model.post_layernorm.bias"],
           "proj": sd["visual_projection.weight"].t().contiguous()}
    L = hf.config.num_hidden_layers
    for i in range(L):
        h = f"vision_model.encoder.layers.{i}."
        o = f"transformer.resblocks.{i}."
        out[o + "ln_1.weight"], out[o + "ln_1.bias"] = sd[h + "layer_norm1.weight"], sd[h + "layer_norm1.bias"]
        out[o + "ln_2.weight"], out[o + "ln_2.bias"] = sd[h + "layer_norm2.weight"], sd[h + "layer_norm2.bias"]
        out[o + "attn.in_proj_weight"] = torch.cat([sd[h + f"self_attn.{n}_proj.weight"] for n in "qkv"])
        out[o + "attn.in_proj_bias"] = torch.cat([sd[h + f"self_attn.{n}_proj.bias"] for n in "qkv"])
        out[o + "attn.out_proj.weight"], out[o + "attn.out_proj.bias"] = sd[h + "self_attn.out_proj.weight"], sd[h + "self_attn.out_proj.bias"]
        out[o + "mlp.c_fc.weight"], out[o + "mlp.c_fc.bias"] = sd[h + "mlp.fc1.weight"], sd[h + "mlp.fc1.bias"]
        out[o + "mlp.c_proj.weight"], out[o + "mlp.c_proj.bias"] = sd[h + "mlp.fc2.weight"], sd[h + "mlp.fc2.bias"]
    return out


def _digest(t):
    """Order-sensitive digest of an activation / gradient tensor: sum, abs-sum, Frobenius norm, and 16 fixed samples."""
    f = t.detach().double().flatten()
    idx = torch.linspace(0, f.numel() - 1, 16).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item(), f.norm().item()], f[idx].numpy()])


VIT_GRAD_NAMES = ["class_embedding", "ln_post.weight", "proj", "transformer.resblocks.0.attn.in_proj_weight",
                  "transformer.resblocks.1.mlp.c_fc.weight", "conv1.weight"]


VIT_CASES = {"b32_quickgelu": ((224, 32, 768, 12, 12, 512), "quick_gelu"), "tiny_gelu": ((56, 14, 128, 2, 2, 64), "gelu")}


def vit_case(name):
    """One G8 case: build the HF tower and the oracle from the same seeded weights, run forward + backward of
    loss = sum(embeds * w) through both, assert they agree, and return what the fixture stores (from the HF side)."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from oracle import vit_oracle as vo
    cfg, act = VIT_CASES[name]
    recs = {}
    img, patch, width, layers, heads, embed = cfg
    torch.manual_seed(0)
    hf = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers,
                                                        num_attention_heads=heads, patch_size=patch, image_size=img,
                                                        projection_dim=embed, hidden_act=act, layer_norm_eps=1e-5)).eval()
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if n.endswith("bias"):
                p.add_(0.02 * torch.randn_like(p))
    ref = vo.OracleVisionTransformer(img, patch, width, layers, heads, embed, act)
    ref.load_state_dict(hf_to_oracle_state(hf))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, img, img, generator=g)
    w = torch.randn(2, embed, generator=g)
    out = hf(pixel_values=x, output_hidden_states=True)
    a = out.image_embeds
    (a * w).sum().backward()
    hid = []
    b = ref(x, hidden=hid)
    (b * w).sum().backward()
    err = (a - b).abs().max().item()
    print(f"[vit/{name}] oracle vs HF embeds max-abs {err:.3e}")
    assert err < 2e-5
    recs[name + "_embeds"] = a.detach().numpy()
    # HF hidden_states[0] is the embedding BEFORE pre_layrnorm; [i >= 1] is the stream after block i
    for i in range(1, layers + 1):
        e = (out.hidden_states[i] - hid[i]).abs().max().item()
        assert e < 5e-4 * max(1.0, hid[i].abs().max().item()), (i, e)
    recs[name + "_hidden"] = np.stack([_digest(out.hidden_states[i]) for i in range(1, layers + 1)])
    recs[name + "_hidden_oracle"] = np.stack([_digest(h) for h in hid[1:]])
    hf_sd = dict(hf.named_parameters())
    hf_grads = {"class_embedding": hf_sd["vision_model.embeddings.class_embedding"].grad,
                "ln_post.weight": hf_sd["vision_model.post_layernorm.weight"].grad,
                "proj": hf_sd["visual_projection.weight"].grad.t(),
                "transformer.resblocks.0.attn.in_proj_weight": torch.cat([hf_sd[f"vision_model.encoder.layers.0.self_attn.{n}_proj.weight"].grad for n in "qkv"]),
                "transformer.resblocks.1.mlp.c_fc.weight": hf_sd["vision_model.encoder.layers.1.mlp.fc1.weight"].grad,
                "conv1.weight": hf_sd["vision_model.embeddings.patch_embedding.weight"].grad}
    mine = dict(ref.named_parameters())
    for n in VIT_GRAD_NAMES:
        r = ((mine[n].grad - hf_grads[n]).norm() / hf_grads[n].norm()).item()
        print(f"[vit/{name}] grad {n}: oracle vs HF relative L2 {r:.3e}")
        assert r < 1e-4, (n, r)
        recs[f"{name}_grad_{n}"] = _digest(hf_grads[n])
        recs[f"{name}_grad_oracle_{n}"] = _digest(mine[n].grad)
    return recs


def golden_vit():
    """G8: the oracle tower vs an INDEPENDENT implementation of the same architecture (HF transformers CLIP): embeddings,
    the residual stream after every block, and the gradients of named parameters of loss = sum(embeds * w)."""
    recs = {}
    for name in VIT_CASES:
        recs.update(vit_case(name))
    np.savez_compressed(os.path.join(GOLD, "vit_hf.npz"), **recs)
    print("wrote tests/golden/vit_hf.npz")


def hf_text_to_oracle_state(hf):
    """HF CLIPTextModelWithProjection -> open_clip CLIP text-tower key names."""
    sd = hf.state_dict()
    out = {"token_embedding.weight": sd["text_model.embeddings.token_embedding.weight"],
           "positional_embedding": sd["text_model.embeddings.position_embedding.weight"],
           "ln_final.weight": sd["text_model.final_layer_norm.weight"], "ln_final.bias": sd["text_model.final_layer_norm.bias"],
           "text_projection": sd["text_projection.weight"].t().contiguous()}
    for i in range(hf.config.num_hidden_layers):
        h = f"text_model.encoder.layers.{i}."
        o = f"transformer.resblocks.{i}."
        out[o + "ln_1.weight"], out[o + "ln_1.bias"] = sd[h + "layer_norm1.weight"], sd[h + "layer_norm1.bias"]
        out[o + "ln_2.weight"], out[o + "ln_2.bias"] = sd[h + "layer_norm2.weight"], sd[h + "layer_norm2.bias"]
        out[o + "attn.in_proj_weight"] = torch.cat([sd[h + f"self_attn.{n}_proj.weight"] for n in "qkv"])
        out[o + "attn.in_proj_bias"] = torch.cat([sd[h + f"self_attn.{n}_proj.bias"] for n in "qkv"])
        out[o + "attn.out_proj.weight"], out[o + "attn.out_proj.bias"] = sd[h + "self_attn.out_proj.weight"], sd[h + "self_attn.out_proj.bias"]
        out[o + "mlp.c_fc.weight"], out[o + "mlp.c_fc.bias"] = sd[h + "mlp.fc1.weight"], sd[h + "mlp.fc1.bias"]
        out[o + "mlp.c_proj.weight"], out[o + "mlp.c_proj.bias"] = sd[h + "mlp.fc2.weight"], sd[h + "mlp.fc2.bias"]
    return out


TEXT_CASES = {  # name: (ctx, vocab, width, heads, layers, embed, act)
    "b32_text": (77, 49408, 512, 8, 12, 512, "quick_gelu"),
    "tiny_text": (16, 100, 64, 1, 2, 32, "gelu"),
}


def make_tokens(ctx, vocab, seed):
    """CLIP-shaped token rows: SOT, a few words, EOT (= the highest id, what argmax pooling keys on), zero padding."""
    g = torch.Generator().manual_seed(seed)
    tok = torch.zeros(3, ctx, dtype=torch.long)
    for r, n in enumerate((3, ctx - 2, 7 if ctx > 9 else 2)):
        tok[r, 0] = vocab - 2
        tok[r, 1:1 + n] = torch.randint(1, vocab - 2, (n,), generator=g)
        tok[r, 1 + n] = vocab - 1
    return tok


def text_case(name):
    """The oracle's encode_text (causal mask, EOT pooling by argmax, text_projection) vs HF CLIPTextModelWithProjection on
    the same seeded weights: forward + backward of loss = sum(text_embeds * w)."""
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    from oracle import vit_oracle as vo
    ctx, vocab, width, heads, layers, embed, act = TEXT_CASES[name]
    recs = {}
    torch.manual_seed(0)
    # eos_token_id = 2 selects HF's legacy pooling = argmax over token ids, which is open_clip's rule
    hf = CLIPTextModelWithProjection(CLIPTextConfig(vocab_size=vocab, hidden_size=width, intermediate_size=4 * width,
                                                    num_hidden_layers=layers, num_attention_heads=heads,
                                                    max_position_embeddings=ctx, projection_dim=embed, hidden_act=act,
                                                    layer_norm_eps=1e-5, eos_token_id=2, bos_token_id=0, pad_token_id=1)).eval()
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if n.endswith("bias"):
                p.add_(0.02 * torch.randn_like(p))
    ref = vo.OracleCLIP(vision_cfg=(32, 16, 64, 1, 1, embed, act), text_width=width, text_heads=heads, text_layers=layers,
                        ctx=ctx, vocab=vocab)
    missing = ref.load_state_dict(hf_text_to_oracle_state(hf), strict=False)
    assert all(k.startswith("visual.") or k == "logit_scale" for k in missing.missing_keys), missing
    tok = make_tokens(ctx, vocab, seed=5)
    w = torch.randn(3, embed, generator=torch.Generator().manual_seed(6))
    out = hf(input_ids=tok, output_hidden_states=True)
    a = out.text_embeds
    (a * w).sum().backward()
    hid = []
    b = ref.encode_text(tok, hidden=hid)
    (b * w).sum().backward()
    err = (a - b).abs().max().item()
    print(f"[text/{name}] oracle vs HF text_embeds max-abs {err:.3e}")
    assert err < 2e-5
    for i in range(layers + 1):
        e = (out.hidden_states[i] - hid[i]).abs().max().item()
        assert e < 5e-4 * max(1.0, hid[i].abs().max().item()), (i, e)
    recs[name + "_tokens"] = tok.numpy()
    recs[name + "_embeds"] = a.detach().numpy()
    recs[name + "_hidden"] = np.stack([_digest(h) for h in out.hidden_states])
    recs[name + "_hidden_oracle"] = np.stack([_digest(h) for h in hid])
    hf_sd = dict(hf.named_parameters())
    pairs = {"text_projection": hf_sd["text_projection.weight"].grad.t(),
             "ln_final.weight": hf_sd["text_model.final_layer_norm.weight"].grad,
             "positional_embedding": hf_sd["text_model.embeddings.position_embedding.weight"].grad,
             "transformer.resblocks.0.attn.in_proj_weight": torch.cat([hf_sd[f"text_model.encoder.layers.0.self_attn.{n}_proj.weight"].grad for n in "qkv"])}
    mine = dict(ref.named_parameters())
    for n, gref in pairs.items():
        r = ((mine[n].grad - gref).norm() / gref.norm()).item()
        print(f"[text/{name}] grad {n}: oracle vs HF relative L2 {r:.3e}")
        assert r < 1e-4, (n, r)
        recs[f"{name}_grad_{n}"] = _digest(gref)
        recs[f"{name}_grad_oracle_{n}"] = _digest(mine[n].grad)
    return recs


def golden_text():
    recs = {}
    for name in TEXT_CASES:
        recs.update(text_case(name))
    np.savez_compressed(os.path.join(GOLD, "text_hf.npz"), **recs)
    print("wrote tests/golden/text_hf.npz")


def export_hf_state(kind, name, path):
    """Weights of a golden case in the oracle's / open_clip's key names, as a plain tensor dict (tests on the GPU box load
    them into the HIP towers and compare with what the fixture recorded from the HF side)."""
    if kind == "vit":
        from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
        (img, patch, width, layers, heads, embed), act = VIT_CASES[name]
        torch.manual_seed(0)
        hf = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers,
                                                            num_attention_heads=heads, patch_size=patch, image_size=img,
                                                            projection_dim=embed, hidden_act=act, layer_norm_eps=1e-5)).eval()
        conv = hf_to_oracle_state
    else:
        from transformers import CLIPTextConfig, CLIPTextModelWithProjection
        ctx, vocab, width, heads, layers, embed, act = TEXT_CASES[name]
        torch.manual_seed(0)
        hf = CLIPTextModelWithProjection(CLIPTextConfig(vocab_size=vocab, hidden_size=width, intermediate_size=4 * width,
                                                        num_hidden_layers=layers, num_attention_heads=heads,
                                                        max_position_embeddings=ctx, projection_dim=embed, hidden_act=act,
                                                        layer_norm_eps=1e-5, eos_token_id=2, bos_token_id=0, pad_token_id=1)).eval()
        conv = hf_text_to_oracle_state
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if n.endswith("bias"):
                p.add_(0.02 * torch.randn_like(p))
    torch.save({k: v.detach().clone() for k, v in conv(hf).items()}, path)


def stage_models(seed_best=11, seed_zero=12):
    """Two TINY oracle CLIPs ("finetuned" and "zero-shot") + heads, seeded: inputs of the stages golden (re-made by the tests)."""
    from oracle import vit_oracle as vo
    out = []
    for seed in (seed_best, seed_zero):
        torch.manual_seed(seed)
        m = vo.OracleCLIP(vision_cfg=TINY, text_width=64, text_heads=1, text_layers=1, ctx=8, vocab=64)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n.endswith("bias"):
                    p.add_(0.02 * torch.randn_like(p))
        W = torch.randn(7, TINY[5]) * (0.5 + torch.rand(7, 1))  # rows of different norms: tau-normalisation matters
        out.append((m, W))
    return out


def stage_loader(seed, n, bsz, C=7, img=64, teacher=None):
    """Seeded image batches; with a teacher (model, W) the labels are its predictions with every fourth one re-drawn, so
    accuracies are neither 0 nor 100 and depend on the head's row norms (what tau-normalisation changes)."""
    g = torch.Generator().manual_seed(seed)
    x, y = torch.randn(n, 3, img, img, generator=g), torch.randint(0, C, (n,), generator=g)
    if teacher is not None:
        model, W = teacher
        with torch.no_grad():
            f = model.encode_image(x)
            pred = ((f / f.norm(dim=-1, keepdim=True)) @ W.t()).argmax(1)
        keep = torch.arange(n) % 4 != 3
        y = torch.where(keep, pred, y)
    return [(x[i:i + bsz], y[i:i + bsz], None, torch.ones(len(y[i:i + bsz]), dtype=torch.long)) for i in range(0, n, bsz)]


def golden_stages():
    """A10 / A11 / f.4: the reference's own extract_test_feats (utils/features.py:122-148), ensemble_model, run_wsft_alpha
    and run_tau_normalization (main.py:27-195) on the fp32 oracle TINY model (CPU)."""
    from oracle import ref_shims
    ref_shims.install()
    import main as ref_main
    import utils.features as rf
    import utils.models as rm
    rf.tqdm = lambda it, *a, **k: it
    (best, Wb), (zero, Wz) = stage_models()
    recs = {}
    # ---- extract_test_feats: batched encode_image + L2 normalise, loader order ----
    loader = stage_loader(21, 22, 8)
    feats = rf.extract_test_feats(best, loader)
    recs["feats_image_features"] = feats["image_features"].numpy()
    recs["feats_labels"] = feats["labels"].numpy()
    # ---- ensemble_model(alpha) ----
    for alpha in (0.5, 0.3):
        w = ref_main.ensemble_model(best, zero, alpha)
        sd = w.state_dict()
        recs[f"wsft_{alpha}_digest"] = np.stack([_digest(sd[k].float()) for k in sorted(sd) if sd[k].dtype.is_floating_point])
        with torch.no_grad():
            recs[f"wsft_{alpha}_embeds"] = w.encode_image(loader[0][0]).numpy()
    # ---- run_wsft_alpha and run_tau_normalization on image / pre-extracted loaders ----
    args = _ref_args(device="cpu", dataset="synthetic", output_dir=tempfile.mkdtemp(), pre_extracted=False, freeze_visual=False,
                     tau_norm=True, loss=torch.nn.CrossEntropyLoss())
    ls = torch.tensor(4.0)
    head_b, head_z = rm.MyLinear(weights=Wb.clone()), rm.MyLinear(weights=Wz.clone())
    Wn = Wb / Wb.norm(dim=-1, keepdim=True)  # the teacher ranks by cosine: the un-normalised head only agrees with it at tau = 1
    val, test = stage_loader(31, 40, 8, teacher=(best, Wn)), stage_loader(32, 48, 8, teacher=(best, Wn))
    wm, wh, wacc = ref_main.run_wsft_alpha(args, best, head_b, val, test, zero, head_z, ls, _logger())
    recs["wsft_alpha_test_acc"] = np.float64(wacc)
    recs["wsft_alpha_head"] = wh.linear.weight.detach().numpy()
    # tau-norm runs on pre-extracted (already normalised) features
    fv, ft = rf.extract_test_feats(best, val), rf.extract_test_feats(best, test)
    pv = [(fv["image_features"][i:i + 8], fv["labels"][i:i + 8], None, None) for i in range(0, 40, 8)]
    pt = [(ft["image_features"][i:i + 8], ft["labels"][i:i + 8], None, None) for i in range(0, 48, 8)]
    th, tau, tacc = ref_main.run_tau_normalization(args, head_b, best, pv, pt, ls, _logger())
    recs["tau_best"] = np.float64(tau)
    recs["tau_test_acc"] = np.float64(tacc)
    recs["tau_head"] = th.linear.weight.detach().numpy()
    print(f"[stages] wsft alpha test acc {wacc:.3f}; best tau {tau:.1f} test acc {tacc:.3f}")
    np.savez_compressed(os.path.join(GOLD, "stages_tiny.npz"), **recs)
    print("wrote tests/golden/stages_tiny.npz")


def prompt_metrics():
    """Synthetic <dataset>_metrics-LAION400M.json content: keys out of numeric order, synonyms with counts, Semi-Aves fields."""
    names = [("7", "marsh wren", "Cistothorus palustris"), ("0", "barn owl", "Tyto alba"), ("12", "cedar waxwing", "Bombycilla cedrorum"),
             ("3", "dark-eyed junco", "Junco hyemalis")]
    return {k: {"name": n, "most_common_name": n.title(), "alternates": {n: 10, n + " bird": 3, s: 1}, "class_id": int(k), "species": s,
                "common_name": n, "most_freq_synonym": n + " bird"} for k, n, s in names}


def golden_prompts():
    """prompt_maker / prompt_maker_aves (utils/prompt.py:83-148) on synthetic metrics for every dataset whose template list the
    build carries — the expected prompts double as the check of those lists."""
    from oracle import ref_shims
    ref_shims.install()
    import utils.prompt as rp
    m = prompt_metrics()
    out = {"prompt_maker": {}, "prompt_maker_aves": {}}
    for ds in ("flowers102", "food101", "oxford_pets", "fgvc-aircraft", "eurosat", "sun397", "dtd", "dtd_selected", "stanford_cars", "semi-aves"):
        for nt in ("name", "most_common_name", "alternates"):
            out["prompt_maker"][f"{ds}|{nt}"] = rp.prompt_maker(m, ds, nt)
    import contextlib, io
    for nt in ("s-name", "c-name", "t-name", "f-name"):
        with contextlib.redirect_stdout(io.StringIO()):
            out["prompt_maker_aves"][nt] = rp.prompt_maker_aves(m, "semi-aves", nt)
    with open(os.path.join(GOLD, "prompts.json"), "w") as f:
        json.dump(out, f)
    print("wrote tests/golden/prompts.json")


PROMPT_TEXT_CFG = (16, 128, 64, 2, 2)  # ctx, vocab, width, heads, layers of the toy text tower of the B8 golden


def prompt_tensor_model(seed=21):
    from oracle import vit_oracle as vo
    torch.manual_seed(seed)
    ctx, vocab, tw, th, tl = PROMPT_TEXT_CFG
    return vo.OracleCLIP(vision_cfg=TINY, text_width=tw, text_heads=th, text_layers=tl, ctx=ctx, vocab=vocab)


def golden_prompt_tensors():
    """B8 compute branch: the reference's own get_text_features (utils/features.py:39-64: encode every template x synonym,
    L2-normalise each, mean, L2-normalise again -> {'all', 'mean'}) and the un-cached branch of cal_prompt_tensors
    (retrieval/sample_retrieval.py:1418-1454: prompt_maker -> get_text_features -> torch.save of both tables) with the fp32
    oracle text tower as `model` and oracle/synth.hash_tokenize as the tokenizer."""
    from oracle import ref_shims
    sr = ref_shims.import_retrieval()
    import utils.features as rf
    import utils.prompt as rp
    model = prompt_tensor_model()
    tok = synth.hash_tokenize(PROMPT_TEXT_CFG[0], PROMPT_TEXT_CFG[1])
    metrics = prompt_metrics()
    recs = {}
    for ds, nt in (("semi-aves", "alternates"), ("dtd", "name")):
        prompts = rp.prompt_maker(metrics, ds, nt)
        td = rf.get_text_features(model, prompts, tok, "encode")
        keys = list(td.keys())
        recs[f"{ds}|{nt}|keys"] = json.dumps(keys)
        recs[f"{ds}|{nt}|counts"] = np.array([td[k]["all"].shape[0] for k in keys], np.int64)
        recs[f"{ds}|{nt}|all"] = torch.cat([td[k]["all"] for k in keys]).numpy()
        recs[f"{ds}|{nt}|mean"] = torch.stack([td[k]["mean"] for k in keys]).numpy()
        print(f"[prompt_tensors/{ds}|{nt}] {len(keys)} classes, {recs[f'{ds}|{nt}|all'].shape[0]} prompts")
    # cal_prompt_tensors, un-cached: runs from a scratch `retrieval/` so that its '../data/<dataset>/prompts/' lands in a temp tree
    tmp = tempfile.mkdtemp()
    os.makedirs(f"{tmp}/retrieval")
    cwd = os.getcwd()
    os.chdir(f"{tmp}/retrieval")
    try:
        args = types.SimpleNamespace(dataset="semi-aves", model_cfg="vitb32_openclip_laion400m", prompt_name="alternates", recal_prompt=False)
        ptd, tpd = sr.cal_prompt_tensors(args, _logger(), model, tok, metrics, "/unused")
        files = sorted(os.listdir(f"{tmp}/data/semi-aves/prompts"))
    finally:
        os.chdir(cwd)
    keys = list(ptd["alternates"].keys())
    recs["cal|keys"] = json.dumps(keys)
    recs["cal|mean"] = torch.stack([ptd["alternates"][k]["mean"] for k in keys]).numpy()
    recs["cal|files"] = json.dumps(files)
    recs["cal|corpus0"] = json.dumps(tpd["alternates"][keys[0]]["corpus"])
    np.savez_compressed(os.path.join(GOLD, "prompt_tensors.npz"), **recs)
    print("wrote tests/golden/prompt_tensors.npz", files)


SECTIONS.update({"probing": golden_probing, "prompt_tensors": golden_prompt_tensors})
SECTIONS.update({"prompts": golden_prompts, "host": golden_host, "loop": golden_loop, "vit": golden_vit, "text": golden_text, "stages": golden_stages})


def _import_hf_first():
    """The tower sections pin the oracle against HF `transformers`, whose availability probes (`importlib.util.find_spec("torchvision")`)
    raise on the spec-less torchvision stand-in that ref_shims installs for the reference's imports: when several sections run in one
    process, transformers and the CLIP classes it loads lazily must be imported BEFORE the first section installs the shims."""
    try:
        from transformers import (CLIPTextConfig, CLIPTextModelWithProjection, CLIPVisionConfig,  # noqa: F401
                                  CLIPVisionModelWithProjection)
    except ImportError:
        pass   # the sections that need it will say so


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    todo = sys.argv[1:] or list(SECTIONS)
    unknown = [s for s in todo if s not in SECTIONS]
    if unknown:
        raise SystemExit(f"unknown section(s) {unknown}; choose from {sorted(SECTIONS)}")
    _import_hf_first()
    for s in todo:
        SECTIONS[s]()
