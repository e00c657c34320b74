"""Path B command line: rank the retrieved LAION rows per class by CLIP similarity, filter, keep the top `--num_samples`
and write `<prefix>.txt` (mirror of the reference's retrieval/sample_retrieval.py __main__ :1673-1748: same flags, same
output files, same log lines where they carry results).

    python -m swat_amd.retrieval.sample_retrieval --prefix T2T500+T2I0.25 --num_samples 500 \
           --sampling_method T2T-rank-T2I-tshd --dataset semi-aves

ref: cal_prompt_tensors :1418-1454, zeroshot_clip_img_filter :278-329, sampling :1471-1670 (-> swat_amd/retrieval/samplers.py),
save_sample_file_list :1457-1469.  The per-class Python loop is one similarity scan + one filtered segmented top-k on the GPU
(csrc/retrieval.hip); prompt embeddings come from the HIP text tower.
"""
import argparse
import json
import os
import random
import shutil
from collections import defaultdict
from time import time

import torch

from .. import _lib, features
from ..logger import get_logger
from ..models import MyLinear
from ..prompt import prompt_maker
from . import samplers
from .extract_mined_feature import _retrieved_root, mined_dataset_root


def cal_prompt_tensors(args, logger, model, tokenizer, metrics, dataset_root, data_dir='../data'):
    """Per-class prompt embeddings ({str(cls): {'all' [P,E], 'mean' [E]}}) for --prompt_name, cached under
    <data_dir>/<dataset>/prompts/ exactly like the reference (:1418-1454) -> (prompt_tensors_dict, text_prompts_dict)."""
    prompts_dir = os.path.join(data_dir, args.dataset, 'prompts/')
    if not os.path.exists(prompts_dir):
        os.makedirs(prompts_dir)
        logger.info(f'Created directory: {prompts_dir}')
    prompt_tensors_dict, text_prompts_dict = {}, {}
    for label_type in [args.prompt_name]:
        prompt_tensors_filename = f"{prompts_dir}{args.dataset}_{args.model_cfg}_{label_type}_prompt_tensors.pth"
        text_prompts_filename = f"{prompts_dir}{args.dataset}_{args.model_cfg}_{label_type}_text_prompts.pth"
        if not args.recal_prompt and os.path.exists(prompt_tensors_filename):
            prompt_tensors_dict[label_type] = torch.load(prompt_tensors_filename, weights_only=False)
            text_prompts_dict[label_type] = torch.load(text_prompts_filename, weights_only=False)
            logger.info(f'Loaded prompt tensors from: {prompt_tensors_filename}')
            logger.info(f'Loaded text prompts from: {text_prompts_filename}')
        else:
            text_prompts = prompt_maker(metrics=metrics, dataset_name=args.dataset, name_type=label_type)
            text_prompts_dict[label_type] = text_prompts
            logger.info(f'Calculating prompt tensors for {label_type} ...')
            prompt_tensors = features.get_text_features(model, text_prompts, tokenizer, 'encode')
            prompt_tensors_dict[label_type] = prompt_tensors
            torch.save(prompt_tensors, prompt_tensors_filename)
            logger.info(f'Saved prompt tensors to: {prompt_tensors_filename}')
            torch.save(text_prompts, text_prompts_filename)
            logger.info(f'Saved text prompts to: {text_prompts_filename}')
    return prompt_tensors_dict, text_prompts_dict


@_lib.retry_op_by_op
def zeroshot_clip_img_filter(model, preprocess, root_folder, pre_extracted_feats=None, head=None):
    """ref :278-329: per class, the rows whose zero-shot prediction (argmax of head(image embedding)) is another class
    are collected as a set of POSITIONS; a file is filtered iff int(<image id of its file name>) is in that set — the
    reference's own matching rule (positions against file ids), kept as it is.  `pre_extracted_feats`: a SegmentedIndex or
    the reference's per-class dict.  -> {str(cls): set(paths)}"""
    if pre_extracted_feats is None:
        raise ValueError('Pre-extracted features are required for zeroshot filtering.')
    index = samplers._as_index(pre_extracted_feats)
    filtered_images_dict = defaultdict(set)
    W = head.linear.weight.detach().to(index.image_features.device, torch.float32)
    # logits[c, i] = <W[c], f_i> in the library's canonical order (one fp32 fma chain over k ascending on the f32 matrix
    # cores, csrc/retrieval_mfma.hip — the same bits as the oracle's restatement), 32 768 rows at a time
    from . import dense
    W = W.contiguous()
    feats = index.image_features
    preds = torch.empty(feats.shape[0], dtype=torch.int64)
    for lo in range(0, feats.shape[0], 32768):
        S = dense.sim_matrix(W, feats[lo:lo + 32768].contiguous())   # [C, n]
        preds[lo:lo + S.shape[1]] = torch.argmax(S, dim=0).cpu()
    off = index.offsets.tolist()
    paths, ids = index.paths_array(), index.file_ids()
    import numpy as np
    fractions = []
    for s, c in enumerate(index.class_ids.tolist()):
        b, e = off[s], off[s + 1]
        if e == b:
            continue
        to_remove = (preds[b:e] != int(c)).nonzero().flatten().numpy()
        hit = np.isin(ids[b:e], to_remove)
        if hit.any():
            filtered_images_dict[str(c)].update(paths[b:e][hit].tolist())
        fractions.append((e - b - len(to_remove)) / (e - b))
    print(f'Average unique images: {round(sum(fractions) / max(1, len(fractions)), 4)}')
    return filtered_images_dict


def sampling(args, logger, model, preprocess, metrics, dataset_root, prompt_tensors_dict, data_dir='../data'):
    """ref :1471-1670 -> (file_list_path, sample_ct); reads <dataset_root>/<dataset>_<model_cfg>_mined.pth."""
    fn = f'{dataset_root}/{args.dataset}_{args.model_cfg}_mined.pth'
    if not os.path.exists(fn):
        logger.info(f'Error: Pre-extracted features not found. {fn}')
        raise NotImplementedError
    pre = torch.load(fn, map_location='cpu', weights_only=False)
    logger.info(f'Loaded pre-extracted mined features from: {fn}')
    index = samplers.transform_extracted_fea(pre)
    filtered_images_dict = defaultdict(set)
    if args.zeroshot_img_filter:
        # (the reference's call site passes an undefined `root_folder`, :1493-1497 — the function itself only needs the features)
        logger.info('Doing zeroshot image filtering!')
        zeroshot_weights = features.prompt_sampler(prompt_tensors_dict[args.prompt_name], sample_by='mean')
        head = MyLinear(weights=zeroshot_weights, bias=False).to('cuda').eval()
        filtered_images_dict = zeroshot_clip_img_filter(model, preprocess, mined_dataset_root(args.dataset, args.root), index, head)
    else:
        logger.info('No zeroshot image filtering!')
    path, sample_ct = samplers.sampling(args, logger, index, prompt_tensors_dict[args.prompt_name],
                                        filtered_images_dict=filtered_images_dict)
    dst = f'{data_dir}/{args.dataset}/'
    os.makedirs(dst, exist_ok=True)
    shutil.copy(path, dst)  # the training CLI reads --retrieval_split from data/<dataset>/ (ref :1466-1467)
    logger.info(f'Copied file to: {dst}')
    return path, sample_ct


def main(argv=None):
    time_start = time()
    parser = argparse.ArgumentParser(description='Arguments for script.')
    parser.add_argument('--prefix', type=str, default=None, help='prefix.txt and foldername')
    parser.add_argument('--dataset', type=str, default='semi-aves', help='Dataset name.')
    parser.add_argument('--root', type=str, default=_retrieved_root(), help='Root directory for storing mined data.')
    parser.add_argument('--model_cfg', type=str, default='vitb32_openclip_laion400m',
                        choices=['vitb32_openclip_laion400m', 'vitb32_openclip_laion2b', 'vitb16_openclip_laion400m',
                                 'vitl14_openclip_laion400m'], help='ViT Transformer arch.')
    parser.add_argument('--database', type=str, default='LAION400M', help='Database from which images are mined.')
    parser.add_argument('--prompt_name', type=str, default='alternates', choices=['most_common_name', 'alternates', 'name'],
                        help='What label to use for making text prompts.')
    parser.add_argument('--sampling_method', type=str, default='T2T-rank', choices=list(samplers.SAMPLERS),
                        help='Ranked sampling by CLIP similarity (the ablation samplers of the reference are out of scope).')
    parser.add_argument('--sampling_threshold', type=float, default=0.0)
    parser.add_argument('--num_samples', type=int, default=500, help='Number of images that are to be sampled.')
    parser.add_argument('--zeroshot_img_filter', action='store_true', default=False, help='zeroshot CLIP image filtering.')
    parser.add_argument('--image_dedup', action='store_true', default=False,
                        help='CLIP image deduplication by filtering duplicates with img sim > 0.9.')
    parser.add_argument('--recal_prompt', action='store_true', default=False, help='Recalculate the prompt tensors.')
    parser.add_argument('--seed', type=int, default=0, help='Seed for random number generator.')
    parser.add_argument('--log_mode', type=str, default='both', choices=['console', 'file', 'both'], help='where to log.')
    parser.add_argument('--data_dir', type=str, default='../data', help='where data/<dataset>/ lives (the reference runs from retrieval/)')
    args = parser.parse_args(argv)
    os.makedirs('output', exist_ok=True)
    random.seed(args.seed)
    torch.manual_seed(args.seed)
    args.case_name = f'{args.dataset}_{args.model_cfg}_{args.prefix}'
    args.output_folder = f'output/{args.case_name}'
    os.makedirs(args.output_folder, exist_ok=True)
    logger = get_logger(args.output_folder, 'sampling', args.log_mode)
    logger.info(f'case_name: {args.case_name}')
    for arg in vars(args):
        logger.info(f'{arg} = {getattr(args, arg)}')
    dataset_root = f'{args.root}/{args.dataset}'
    metric_fn = f'{args.data_dir}/{args.dataset}/{args.dataset}_metrics-{args.database.upper()}.json'
    with open(metric_fn, 'r') as f:
        metrics = json.load(f)
    logger.info(f'Loaded metrics from: {metric_fn}')
    logger.info(f'len(metrics): {len(metrics)}')
    from ..engine import get_engine
    model, preprocess, tokenizer = get_engine(model_cfg=args.model_cfg, device='cuda')
    model.eval()
    prompt_tensors_dict, _ = cal_prompt_tensors(args, logger, model, tokenizer, metrics, dataset_root, data_dir=args.data_dir)
    file_list_path, sample_ct = sampling(args, logger, model, preprocess, metrics, dataset_root, prompt_tensors_dict, data_dir=args.data_dir)
    logger.info(f'sample_ct: {sample_ct}')
    logger.info(f'file_list_path: {file_list_path}')
    logger.info(f'Done, time: {round(time() - time_start)} seconds.')
    return file_list_path, sample_ct


if __name__ == '__main__':
    main()
