"""Class-grouped retrieval index resident in HBM.

Replaces the reference's O(N) Python loop `transform_extracted_fea`
(ref: retrieval/sample_retrieval.py:1387-1415) by a stable group-by: rows are permuted once so
that each class is one contiguous segment (classes ascending by integer id, file order kept
inside a class — the order the reference's samplers visit them, :738-739 / :789-790) and the
samplers work on CSR offsets.
"""
import json
import os

import numpy as np
import torch

from . import ops


def group_by_label(labels):
    """-> (order [N] int64, class_ids [C] int64, offsets [C+1] int64); stable, any device."""
    labels = labels.to(torch.int64)
    sorted_labels, order = torch.sort(labels, stable=True)
    class_ids, counts = torch.unique_consecutive(sorted_labels, return_counts=True)
    offsets = torch.zeros(class_ids.numel() + 1, dtype=torch.int64, device=labels.device)
    offsets[1:] = torch.cumsum(counts, 0)
    return order, class_ids, offsets


class SegmentedIndex:
    """Device-resident, class-grouped copy of a `mined.pth`-shaped dict
    (ref: retrieval/extract_mined_feature.py:270-279: image_features, caption_features, labels, filepath)."""

    def __init__(self, image_features, caption_features, labels, filepath=None, device="cuda"):
        order, class_ids, offsets = group_by_label(labels.to(device))
        self.order = order  # grouped row -> original row
        self.class_ids = class_ids
        self.offsets = offsets
        self.image_features = image_features.to(device)[order].contiguous() if image_features is not None else None
        self.caption_features = caption_features.to(device)[order].contiguous() if caption_features is not None else None
        counts = offsets[1:] - offsets[:-1]
        self.row_seg = torch.repeat_interleave(torch.arange(class_ids.numel(), device=device, dtype=torch.int32),
                                               counts)
        self.filepath = None
        if filepath is not None:
            # one numpy fancy-index instead of a Python loop over N rows; `filepath` (list view) is made on demand
            self._paths = np.asarray(filepath)[order.cpu().numpy()]
        self.n_rows = int(order.numel())
        self.n_seg = int(class_ids.numel())

    def paths_array(self):
        """numpy unicode array [N] of the grouped rows' file paths (vectorised membership tests, slicing by row list)."""
        return self._paths

    def file_ids(self):
        """int64 [N]: int(<file name without extension>) of every grouped row — what the reference's duplicate / zero-shot
        filters compare POSITIONS against (retrieval/sample_retrieval.py:262-267, :312-316); numpy string ops, cached."""
        if getattr(self, "_file_ids", None) is None:
            base = np.char.rpartition(self._paths, "/")[:, 2]
            self._file_ids = np.char.partition(base, ".")[:, 0].astype(np.int64)
        return self._file_ids

    @property
    def filepath(self):
        return self._paths

    @filepath.setter
    def filepath(self, v):
        self._paths = None if v is None else np.asarray(v)
        self._file_ids = None

    def queries_from_prompt_tensors(self, prompt_tensors):
        """Stack prompt_tensors[str(cls)]['mean'] in segment order (ref: :749-750, :801-802)."""
        rows = [prompt_tensors[str(int(c))]["mean"].reshape(-1) for c in self.class_ids.tolist()]
        return torch.stack(rows).to(self.offsets.device, torch.float32).contiguous()

    def sample(self, queries, k, rank_by="t2t", use_t2i_thr=False, thr=0.0, t2i_thr=0.25, excluded=None):
        """rank_by='t2t' [+use_t2i_thr] : t2t_ranked_sampler / t2t_ranked_t2i_tshd_sampler (:724-825)
        rank_by='t2i'                  : t2i_ranked_sampler (:1195-1243)
        excluded: uint8 [N] over GROUPED rows (dup-set / zero-shot-filter-set membership).
        Returns idx [S,k] (grouped rows, -1 padded), val [S,k], cnt [S], sims, t2i_sims."""
        t2i = None
        if rank_by == "t2t":
            sims = ops.similarity_segmented(self.caption_features, queries, self.row_seg)
            if use_t2i_thr:
                t2i = ops.similarity_segmented(self.image_features, queries, self.row_seg)
        elif rank_by == "t2i":
            sims = ops.similarity_segmented(self.image_features, queries, self.row_seg)
        else:
            raise NotImplementedError(rank_by)
        idx, val, cnt = ops.segment_topk(sims, self.offsets, k, thr=thr, t2i=t2i, t2i_thr=t2i_thr, excluded=excluded)
        return idx, val, cnt, sims, t2i


    # ---- on-disk form: one directory of raw .npy arrays in GROUPED row order, so a rank can memory-map
    # the files and move only its own row range [lo, hi) to HBM (the reference keeps one pickled dict, mined.pth,
    # retrieval/extract_mined_feature.py:270-279, which every process would have to unpickle whole) ----
    def save(self, path):
        os.makedirs(path, exist_ok=True)
        np.save(os.path.join(path, "order.npy"), self.order.cpu().numpy())
        np.save(os.path.join(path, "class_ids.npy"), self.class_ids.cpu().numpy())
        np.save(os.path.join(path, "offsets.npy"), self.offsets.cpu().numpy())
        for name in ("image_features", "caption_features"):
            t = getattr(self, name)
            if t is not None:
                np.save(os.path.join(path, name + ".npy"), t.cpu().numpy())
        if self._paths is not None:
            np.save(os.path.join(path, "filepath.npy"), self._paths)
        meta = {"format": "swat_amd.SegmentedIndex/1", "n_rows": self.n_rows, "n_seg": self.n_seg,
                "dim": int(self.caption_features.shape[1] if self.caption_features is not None else self.image_features.shape[1]),
                "dtype": "float32", "row_order": "grouped by class id ascending, file order inside a class"}
        with open(os.path.join(path, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1)

    @classmethod
    def load(cls, path, device="cuda", rows=None):
        """rows = (lo, hi): keep only that range of grouped rows (a shard); offsets are clipped and rebased to it."""
        meta = json.load(open(os.path.join(path, "meta.json")))
        assert meta["format"] == "swat_amd.SegmentedIndex/1", meta["format"]
        lo, hi = rows if rows is not None else (0, meta["n_rows"])
        self = cls.__new__(cls)
        self.order = torch.from_numpy(np.load(os.path.join(path, "order.npy"), mmap_mode="r")[lo:hi].copy()).to(device)
        self.class_ids = torch.from_numpy(np.load(os.path.join(path, "class_ids.npy"))).to(device)
        offsets = torch.from_numpy(np.load(os.path.join(path, "offsets.npy")))
        self.offsets = (offsets.clamp(lo, hi) - lo).to(device).contiguous()
        for name in ("image_features", "caption_features"):
            f = os.path.join(path, name + ".npy")
            setattr(self, name, torch.from_numpy(np.load(f, mmap_mode="r")[lo:hi].copy()).to(device) if os.path.exists(f) else None)
        counts = self.offsets[1:] - self.offsets[:-1]
        self.row_seg = torch.repeat_interleave(torch.arange(self.class_ids.numel(), device=device, dtype=torch.int32), counts)
        self._paths = None
        fp = os.path.join(path, "filepath.npy")
        if os.path.exists(fp):  # fixed-width unicode array: a shard memory-maps it and copies only its own rows
            self._paths = np.load(fp, mmap_mode="r")[lo:hi].copy()
        elif os.path.exists(os.path.join(path, "filepath.txt")):
            self._paths = np.asarray(open(os.path.join(path, "filepath.txt")).read().split("\n")[lo:hi])
        self.n_rows = hi - lo
        self.n_seg = int(self.class_ids.numel())
        self.row_base = lo
        return self
