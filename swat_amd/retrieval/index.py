"""Class-grouped retrieval index resident in HBM.

Replaces the reference's O(N) Python loop `transform_extracted_fea`
(ref: retrieval/sample_retrieval.py:1387-1415) by a stable group-by: rows are permuted once so
that each class is one contiguous segment (classes ascending by integer id, file order kept
inside a class — the order the reference's samplers visit them, :738-739 / :789-790) and the
samplers work on CSR offsets.
"""
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
            o = order.tolist()
            self.filepath = [filepath[i] for i in o]
        self.n_rows = int(order.numel())
        self.n_seg = int(class_ids.numel())

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
