"""oracle/make_golden.py — TEST INFRASTRUCTURE ONLY (runs in the build container only).

Imports the reference (tian1327/SWAT at /root/reference, read-only) through oracle/ref_shims.py,
runs the reference's OWN functions on seeded synthetic inputs and writes their outputs as small
golden fixtures under tests/golden/.  Inputs are never stored: every fixture records the seed
and the generator below (`synth_*`) is re-run by the tests (numpy's legacy RandomState stream is
frozen, so the inputs are reproducible bit for bit).

    python oracle/make_golden.py [retrieval] [host] [loop] [vit]

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
GOLD = os.path.join(ROOT, "tests", "golden")
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


SECTIONS = {"retrieval": golden_retrieval}


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        SECTIONS[s]()
