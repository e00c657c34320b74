"""Prompt templates -> per-class text embeddings (mirror of the reference's utils/prompt.py:6-150).

`prompt_maker` / `prompt_maker_aves` / `get_prompts_tensors` / `set_prompt` keep the reference's signatures, key order
(metrics key resp. class_id, sorted numerically) and the on-disk cache `data/<dataset>/prompts/<dataset>_<model_cfg>_<name>_*.pth`.
The per-dataset template tables (ref :152-389) are data: the short dataset-specific ones are built in; the two long lists
(the 80 OpenAI ImageNet templates, the 34 Caltech-101 ones) are read from `SWAT_TEMPLATES_JSON` ({dataset: [template, ...]} or,
for semi-aves, {dataset: {name_type: [...]}}), which also overrides any built-in entry.  With `--cls_init random` (or
--synthetic) no text is needed and `set_prompt` returns empty tables."""
import json
import os

import torch

from . import features

DEFAULT_TEMPLATES = ['a photo of a {}.']


def _pair(*stems):  # 'a photo of a {} X.' and 'a photo of the {} X.' families
    return [f'a photo of {art} {{}} {x}.' for art in ('a', 'the') for x in stems]


_BIRD = ['a photo of a {}, a type of bird.']
TEMPLATES_DIC = {
    'flowers102': ['a photo of a {}, a type of flower.'],
    'food101': ['a photo of {}, a type of food.'],
    'oxford_pets': ['a photo of a {}, a type of pet.'],
    'fgvc-aircraft': [f'a photo of {art} {{}}, a type of aircraft.' for art in ('a', 'the')],
    'eurosat': [f'a centered satellite photo of {art}{{}}.' for art in ('', 'a ', 'the ')],
    'sun397': [f'a photo of {art} {{}}.' for art in ('a', 'the')],
    'dtd': _pair('texture', 'pattern', 'thing', 'object'),
    'stanford_cars': ['a photo of a {}.', 'a photo of the {}.', 'a photo of my {}.', 'i love my {}!'] +
                     [f'a photo of my {adj} {{}}.' for adj in ('dirty', 'clean', 'new', 'old')],
    'semi-aves': {**{n: _BIRD for n in ('s-name', 'c-name', 'f-name', 'most_common_name', 'alternates', 'most_common_name_REAL', 'name')},
                  't-name': ['a photo of a {}, a type of bird, commonally known as {}.']},
}
TEMPLATES_DIC['dtd_selected'] = TEMPLATES_DIC['dtd']


def load_templates(dataset, name_type=None):
    """Templates of `dataset` (for semi-aves: of its `name_type`): SWAT_TEMPLATES_JSON first, then the built-in table."""
    table = dict(TEMPLATES_DIC)
    path = os.environ.get('SWAT_TEMPLATES_JSON')
    if path and os.path.exists(path):
        table.update(json.load(open(path)))
    entry = table.get(dataset)
    if entry is None:
        if dataset in ('imagenet', 'imagenet_1k', 'imagenet_1k_mined', 'imagenet_v2', 'caltech101'):
            raise KeyError(f"the {dataset} template list is not built in: point SWAT_TEMPLATES_JSON at a JSON file with a "
                           f"'{dataset}' entry (ref utils/prompt.py:152-188, 240-358)")
        return DEFAULT_TEMPLATES
    if isinstance(entry, dict):
        return entry[name_type]
    return entry


def prompt_maker(metrics, dataset_name, name_type='most_common_name'):
    """ref :83-105 -> {metrics key: {'corpus': [prompt, ...]}}, sorted by int(key); 'alternates' = templates for every synonym."""
    templates = load_templates(dataset_name, name_type)
    prompts = {}
    for key in metrics.keys():
        label = metrics[key][name_type]
        if name_type == 'alternates':
            prompts[key] = {'corpus': [t.format(alt) for alt in label.keys() for t in templates]}
        else:
            prompts[key] = {'corpus': [t.format(label) for t in templates]}
    return dict(sorted(prompts.items(), key=lambda x: int(x[0])))


def prompt_maker_aves(metrics, dataset_name, name_type='s-name'):
    """ref :108-148 -> {class_id: {'corpus': [...]}}: scientific / common / both / most frequent synonym names of Semi-Aves."""
    templates = load_templates(dataset_name, name_type.split('_')[0])
    prompts = {}
    for key in metrics.keys():
        m = metrics[key]
        s_name, c_name = m['species'], m['common_name']
        if name_type == 's-name':
            lst = [t.format(s_name) for t in templates]
        elif name_type in ('c-name', 'c-name-80prompts'):
            lst = [t.format(c_name) for t in templates]
        elif name_type == 't-name':
            lst = [t.format(s_name, c_name) for t in templates]
        elif name_type == 'f-name':
            lst = [t.format(m['most_freq_synonym']) for t in templates]
        else:
            raise NotImplementedError(f'prompt_maker_aves: name_type {name_type} (the attribute prompts need assets outside this repository)')
        prompts[str(m['class_id'])] = {'corpus': lst}
    return dict(sorted(prompts.items(), key=lambda x: int(x[0])))


def get_prompts_tensors(args, model, tokenizer, logger):
    """ref :6-68 -> (prompt_tensors_dict, text_prompts_dict, tokenized_text_prompts_dict) for args.prompt_name, cached under
    data/<dataset>/prompts/ unless --recal_prompt."""
    root = getattr(args, 'dataset_root', None) or f'data/{args.dataset}'
    metric_fn = f'{root}/{args.dataset}_metrics-{args.database.upper()}.json'
    if args.dataset == 'dtd_selected':
        metric_fn = f'{root}/{args.dataset}_metrics-{args.database.upper()}_new_idx.json'
    with open(metric_fn) as f:
        metrics = json.load(f)
    logger.info(f'Loaded metrics from: {metric_fn}')
    logger.info(f'len(metrics): {len(metrics)}')
    prompts_dir = os.path.join(root, 'prompts/')
    os.makedirs(prompts_dir, exist_ok=True)
    tensors, texts, tokens = {}, {}, {}
    for label_type in [args.prompt_name]:
        stem = f'{prompts_dir}{args.dataset}_{args.model_cfg}_{label_type}'
        fn_t, fn_x, fn_k = f'{stem}_prompt_tensors.pth', f'{stem}_text_prompts.pth', f'{stem}_tokenized_text_prompts.pth'
        if not getattr(args, 'recal_prompt', False) and os.path.exists(fn_k):
            logger.info(f'Loading prompt tensors from {fn_t}')
            tensors[label_type], texts[label_type], tokens[label_type] = torch.load(fn_t), torch.load(fn_x), torch.load(fn_k)
            continue
        logger.info(f'Calculating prompt tensors for {label_type} ...')
        text_prompts = prompt_maker(metrics, args.dataset, label_type)
        texts[label_type] = text_prompts
        torch.save(text_prompts, fn_x)
        tokens[label_type] = features.get_text_features(model, text_prompts, tokenizer, 'tokenize')
        torch.save(tokens[label_type], fn_k)
        tensors[label_type] = features.get_text_features(model, text_prompts, tokenizer, 'encode')
        torch.save(tensors[label_type], fn_t)
        logger.info(f'Saved prompt tensors to {fn_t}')
    return tensors, texts, tokens


def set_prompt(args, model, tokenizer, logger):
    """ref :71-80 -> (prompt_tensors, text_prompts, tokenized_text_prompts, prompt_tensors_dict)."""
    if args.cls_init == 'random':
        logger.info('cls_init random: no text prompts needed.')
        return None, None, None, {}
    tensors, texts, tokens = get_prompts_tensors(args, model, tokenizer, logger)
    return tensors[args.prompt_name], texts[args.prompt_name], tokens[args.prompt_name], tensors
