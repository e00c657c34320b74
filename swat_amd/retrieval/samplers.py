"""`--sampling_method` plug-ins and the `sampling()` driver (mirror of the reference's retrieval/sample_retrieval.py).

ref: transform_extracted_fea :1387-1415, t2t_ranked_sampler :724-771, t2t_ranked_t2i_tshd_sampler :774-825,
     t2i_ranked_sampler :1195-1243, add_to_split :439-482, add_t2t_ranked_t2i_tshd_to_split :492-540,
     remove_near_duplicates2 :237-275, save_sample_file_list :1457-1469, sampling :1471-1670.

Same call signatures and return values ((mined_split, num_imgs_sampled_dict)); `pre_extracted_feats` may be the
reference's per-class dict (transform_extracted_fea output) or a swat_amd SegmentedIndex.  The per-class Python loop
(GEMV + CPython sort + per-row dict lookups) is replaced by one similarity scan + one filtered segmented top-k on the GPU.
"""
import json
import os
from collections import defaultdict

import torch

from . import dense
from .index import SegmentedIndex


def transform_extracted_fea(pre_extracted_feats, device="cuda"):
    """mined.pth dict -> class-grouped device index (the reference builds a dict of per-class stacks, :1387-1415)."""
    return SegmentedIndex(pre_extracted_feats['image_features'], pre_extracted_feats['caption_features'],
                          pre_extracted_feats['labels'], pre_extracted_feats['filepath'], device=device)


def _as_index(pre, device="cuda"):
    if isinstance(pre, SegmentedIndex):
        return pre
    classes = sorted(pre.keys(), key=lambda x: int(x))
    img = torch.cat([pre[c]['feats'] for c in classes])
    cap = torch.cat([pre[c]['caption_feats'] for c in classes])
    labels = torch.cat([torch.full((len(pre[c]['file_paths']),), int(c), dtype=torch.int64) for c in classes])
    paths = [p for c in classes for p in pre[c]['file_paths']]
    return SegmentedIndex(img, cap, labels, paths, device=device)


def _excluded_mask(index, duplicates_dict, filtered_images_dict):
    """uint8 [N] over grouped rows: 1 iff the row's path is in its class's duplicate / zero-shot-filter set (what the
    reference checks per row inside its Python walk, :463-469 / :507-527).  One vectorised membership test per class that
    HAS exclusions, on the index's numpy path array — never a Python loop over all N rows."""
    if not duplicates_dict and not filtered_images_dict:
        return None
    import numpy as np
    paths = index.paths_array()
    off = index.offsets.tolist()
    mask = np.zeros(index.n_rows, dtype=np.uint8)
    for s, c in enumerate(index.class_ids.tolist()):
        excl = set(duplicates_dict.get(str(c), ())) | set(filtered_images_dict.get(str(c), ()))
        if excl:
            mask[off[s]:off[s + 1]] = np.isin(paths[off[s]:off[s + 1]], np.array(sorted(excl), dtype=paths.dtype))
    return torch.from_numpy(mask).to(index.offsets.device)


def _run(args, logger, prompt_tensors, num_samples, threshold, pre, duplicates_dict, filtered_images_dict, rank_by,
         use_t2i_thr, t2i_threshold=0.25, audit_prefix=None):
    index = _as_index(pre)
    queries = index.queries_from_prompt_tensors(prompt_tensors)
    excl = _excluded_mask(index, duplicates_dict or {}, filtered_images_dict or {})
    idx, val, cnt, sims, t2i = index.sample(queries, num_samples, rank_by=rank_by, use_t2i_thr=use_t2i_thr, thr=threshold,
                                            t2i_thr=t2i_threshold, excluded=excl)
    idx_c, cnt_c = idx.cpu(), cnt.cpu().tolist()
    mined_split = {'feature_list': [], 'label_list': [], 'file_list': []}
    num_imgs_sampled_dict = {}
    sampled_list = []
    for s, c in enumerate(index.class_ids.tolist()):
        n = cnt_c[s]
        num_imgs_sampled_dict[str(c)] = n
        if n == 0:
            continue
        rows = idx_c[s, :n]
        mined_split['feature_list'].append(index.image_features[rows.to(index.image_features.device)].cpu())
        mined_split['label_list'].append(torch.full((n,), int(c), dtype=torch.int64))
        files = index.paths_array()[rows.numpy()].tolist()
        mined_split['file_list'].append(files)
        sv = val[s, :n].cpu().tolist()
        sampled_list += [f'{round(v, 4)}/{threshold}, {f}' for v, f in zip(sv, files)]
    out_dir = getattr(args, 'output_folder', None)
    if out_dir and audit_prefix is not None:
        with open(f'{out_dir}/{audit_prefix}sampled_list.txt', 'w') as f:
            f.write('\n'.join(sampled_list))
    logger.info(f'len(sampled_list): {len(sampled_list)}')
    return mined_split, num_imgs_sampled_dict


def t2t_ranked_sampler(args, logger, prompt_tensors, num_samples, threshold, pre_extracted_feats,
                       duplicates_dict=None, filtered_images_dict=None):
    return _run(args, logger, prompt_tensors, num_samples, threshold, pre_extracted_feats, duplicates_dict, filtered_images_dict,
                't2t', False, audit_prefix=f'{getattr(args, "prefix", "")}_')


def t2t_ranked_t2i_tshd_sampler(args, logger, prompt_tensors, num_samples, threshold, pre_extracted_feats,
                                duplicates_dict=None, filtered_images_dict=None, t2i_threshold=0.25):
    return _run(args, logger, prompt_tensors, num_samples, threshold, pre_extracted_feats, duplicates_dict, filtered_images_dict,
                't2t', True, t2i_threshold, audit_prefix='')


def t2i_ranked_sampler(args, logger, prompt_tensors, num_samples, threshold, pre_extracted_feats,
                       duplicates_dict=None, filtered_images_dict=None):
    return _run(args, logger, prompt_tensors, num_samples, threshold, pre_extracted_feats, duplicates_dict, filtered_images_dict,
                't2i', False, audit_prefix=f'{getattr(args, "prefix", "")}_')


def remove_near_duplicates2(pre_extracted_feats):
    return dense.remove_near_duplicates2(_as_index(pre_extracted_feats))


def save_sample_file_list(args, final_file_list, label_tensor, logger=None):
    """'{path} {label} 0' per kept row (0 = retrieved, 1 = few-shot) — ref :1457-1469."""
    fn = f'{args.output_folder}/{args.prefix}.txt'
    with open(fn, 'w') as f:
        for i, file_path in enumerate(final_file_list):
            f.write(f'{file_path} {label_tensor[i].item()} {0}\n')
    if logger:
        logger.info(f'Saved file_list to: {fn}')
    return fn


SAMPLERS = {'T2T-rank': t2t_ranked_sampler, 'T2T-rank-T2I-tshd': t2t_ranked_t2i_tshd_sampler, 'T2I-rank': t2i_ranked_sampler}


def sampling(args, logger, pre_extracted_feats, prompt_tensors, filtered_images_dict=None):
    """ref :1471-1670 for the implemented sampling methods -> (file_list_path, sample_ct)."""
    index = pre_extracted_feats if isinstance(pre_extracted_feats, SegmentedIndex) else transform_extracted_fea(pre_extracted_feats)
    filtered_images_dict = filtered_images_dict if filtered_images_dict is not None else defaultdict(set)
    duplicate_images_dict = defaultdict(set)
    if getattr(args, 'image_dedup', False):
        duplicate_images_dict, fracs, avg = dense.remove_near_duplicates2(index)
        logger.info(f'avg_dup_images_fraction: {avg}')
    if args.sampling_method not in SAMPLERS:
        raise NotImplementedError(f'sampling_method {args.sampling_method}: ablation samplers are out of scope (SURVEY §2 row 15)')
    mined_split, num_imgs_sampled_dict = SAMPLERS[args.sampling_method](args, logger, prompt_tensors, args.num_samples, 0.0, index,
                                                                      duplicate_images_dict, filtered_images_dict)
    final_file_list = [p for fl in mined_split['file_list'] for p in fl]
    labels_tensor = torch.cat(mined_split['label_list'], dim=0) if mined_split['label_list'] else torch.zeros(0, dtype=torch.int64)
    path = save_sample_file_list(args, final_file_list, labels_tensor, logger)
    with open(f'{args.output_folder}/{args.prefix}_num_imgs_sampled.json', 'w') as f:
        json.dump(num_imgs_sampled_dict, f, indent=4)
    return path, len(final_file_list)
