"""Prompt templates -> per-class text embeddings (mirror of the reference's utils/prompt.py:6-105, reduced).

The reference ships per-dataset template tables (utils/prompt.py:373-389, 80 ImageNet templates etc.) and
`data/<ds>/<ds>_metrics-LAION400M.json`; those assets are not part of this repository.  `set_prompt` reads the
metrics JSON when it exists and templates from `SWAT_TEMPLATES_JSON` ({dataset: [template, ...]}), defaulting to
'a photo of a {}.'; with `--cls_init random` (or --synthetic) no text is needed and it returns empty tables."""
import json
import os

import torch

from . import features

DEFAULT_TEMPLATES = ['a photo of a {}.']


def load_templates(dataset):
    path = os.environ.get('SWAT_TEMPLATES_JSON')
    if path and os.path.exists(path):
        table = json.load(open(path))
        return table.get(dataset, DEFAULT_TEMPLATES)
    return DEFAULT_TEMPLATES


def prompt_maker(metrics, dataset_name, name_type='name'):
    """-> {class_id: {'corpus': [prompt, ...]}} (ref :83-105): templates x names ('alternates' = all synonyms)."""
    templates = load_templates(dataset_name)
    prompts = {}
    for i, (key, info) in enumerate(metrics.items()):
        if name_type == 'alternates':
            names = list(info['alternates'].keys())
        else:
            names = [info[name_type if name_type in info else 'name']]
        prompts[str(i)] = {'corpus': [t.format(n) for n in names for t in templates]}
    return prompts


def set_prompt(args, model, tokenizer, logger):
    """ref :71-80 -> (prompt_tensors, text_prompts, tokenized_text_prompts, prompt_tensors_dict)."""
    if args.cls_init == 'random':
        logger.info('cls_init random: no text prompts needed.')
        return None, None, None, {}
    metric_fn = f'{args.dataset_root}/{args.dataset}_metrics-{args.database.upper()}.json'
    with open(metric_fn) as f:
        metrics = json.load(f)
    text_prompts = prompt_maker(metrics, args.dataset, args.prompt_name)
    prompt_tensors = features.get_text_features(model, text_prompts, tokenizer, 'encode')
    tokenized = features.get_text_features(model, text_prompts, tokenizer, 'tokenize')
    return prompt_tensors, text_prompts, tokenized, {args.prompt_name: prompt_tensors}
