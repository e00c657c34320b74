"""Text-prompt / image feature helpers (mirror of the reference's utils/features.py).

ref: prompt_sampler :12-23, operate_on_prompt :27-36, get_text_features :39-64, extract_test_feats :122-148.
"""
import random

import torch

from . import _lib, ops


def prompt_sampler(prompt_tensors, sample_by='mean'):
    sampled_prompts = []
    for i in prompt_tensors.keys():
        if sample_by == 'mean':
            sampled_prompts.append(prompt_tensors[i]['mean'])
        elif sample_by == 'random':
            sampled_prompts.append(random.choice(prompt_tensors[i]['all']))
        else:
            raise NotImplementedError
    return torch.stack(sampled_prompts, dim=0)


def l2_normalize(x):
    """x / x.norm(dim=-1, keepdim=True) on the GPU kernel (A3)."""
    y, _ = ops.l2norm_fwd(x.contiguous().float())
    return y


def operate_on_prompt(model, text, operation, tokenize):
    if operation == 'encode':
        dev = next(model.parameters()).device
        feats = model.encode_text(tokenize(text).to(dev))
        return l2_normalize(feats)
    elif operation == 'tokenize':
        return tokenize(text)


def get_text_features(model, prompt_dict, tokenize, operation='encode'):
    tensor_dict = {}
    model.eval()
    with torch.no_grad():
        for key, info in prompt_dict.items():
            source = {}
            prompts = [p for p in info['corpus']]
            stacked_tensor = operate_on_prompt(model, prompts, operation, tokenize)
            source['all'] = stacked_tensor
            if operation == 'encode':
                mean_tensor = torch.mean(stacked_tensor, dim=0)
                mean_tensor /= mean_tensor.norm(dim=-1, keepdim=True)
                source['mean'] = mean_tensor
            tensor_dict[key] = source
    return tensor_dict


@_lib.retry_op_by_op
def extract_test_feats(model, dataloader):
    """ref :122-148 — batched encode_image + L2-normalise -> {'image_features' [N,E], 'labels' [N]} (CPU tensors)."""
    img_feats_lst, labels_lst = [], []
    dev = next(model.parameters()).device
    model.eval()
    for data in dataloader:
        imgs, labels, text, source = data
        with torch.no_grad():
            img_feats = l2_normalize(model.visual.forward_raw(imgs.to(dev)))
        img_feats_lst.append(img_feats.cpu())
        labels_lst.append(labels.long().cpu())
    return {'image_features': torch.cat(img_feats_lst, dim=0), 'labels': torch.cat(labels_lst, dim=0)}
