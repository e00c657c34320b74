"""`--model_cfg` plug-in: get_engine(model_cfg, device, mode) -> (model, preprocess, tokenizer)
(mirror of the reference's utils/extras.py:138-171 with OPENCLIP_MODEL_DIC :97-114).

The returned model is swat_amd.clip_model.CLIP (open_clip state-dict key names, HIP vision tower).  There is no
network in this environment: weights are random-init open_clip-style unless `SWAT_WEIGHTS_DIR/<tag>/<arch>.pt`
(an open_clip state_dict) exists, in which case it is loaded.
"""
import os

import torch

from .clip_model import CLIP

OpenCLIP_model_dic_LAION400M = {
    'ViT-B/32': ('ViT-B-32-quickgelu', 'laion400m_e32'),
    'ViT-B/16': ('ViT-B-16', 'laion400m_e32'),
    'ViT-L/14': ('ViT-L-14', 'laion400m_e32'),
}
OpenCLIP_model_dic_LAION2B = {
    'ViT-B/32': ('ViT-B-32', 'laion2b_s34b_b79k'),
    'ViT-B/16': ('ViT-B-16', 'laion2b_s34b_b88k'),
    'ViT-L/14': ('ViT-L-14', 'laion2b_s32b_b82k'),
}
ARCH_DIC = {'vitb32': 'ViT-B/32', 'vitb16': 'ViT-B/16', 'vitl14': 'ViT-L/14'}
OPENCLIP_MODEL_DIC = {
    'laion400m': {k: (v[1], v[0]) for k, v in OpenCLIP_model_dic_LAION400M.items()},
    'laion2b': {k: (v[1], v[0]) for k, v in OpenCLIP_model_dic_LAION2B.items()},
}
MODEL_CFG_CHOICES = ['vitb32_openclip_laion400m', 'vitb16_openclip_laion400m', 'vitb32_openclip_laion2b',
                     'vitl14_openclip_laion400m']  # vitl14 extends the reference's list (BASELINE config 5)


def parse_model_cfg(model_cfg):
    cfgs = model_cfg.split('_')
    arch, model_name = cfgs[0], cfgs[1]
    corpus = cfgs[2] if len(cfgs) == 3 else None
    if model_name != 'openclip':
        raise NotImplementedError(f"{model_cfg}: only the open_clip ViT family has a HIP tower "
                                  "(OpenAI clip.load / ResNet-50 are out of scope, SURVEY App. A)")
    corpus_config, model_arch = OPENCLIP_MODEL_DIC[corpus][ARCH_DIC[arch]]
    return model_arch, corpus_config


def get_engine(model_cfg, device='cuda', mode='val', seed=0):
    model_arch, corpus_config = parse_model_cfg(model_cfg)
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)  # deterministic open_clip-style random init without disturbing the caller's stream
    model = CLIP(model_arch)
    torch.random.set_rng_state(gen_state)
    wdir = os.environ.get('SWAT_WEIGHTS_DIR')
    if wdir:
        path = os.path.join(wdir, corpus_config, f'{model_arch}.pt')
        if os.path.exists(path):
            model.load_state_dict(torch.load(path, map_location='cpu'))
    model.float()
    model.to(device)
    from .transforms import build_transform
    from .tokenizer import tokenize
    preprocess = build_transform(224, 'val')
    if mode == 'train':
        return model, build_transform(224, 'train'), preprocess, tokenize
    elif mode == 'val':
        return model, preprocess, tokenize
    raise NotImplementedError
