"""Image transforms without torchvision (ref: utils/extras.py:66-85 `transform`, open_clip's val preprocess).

train: RandomResizedCrop(n, scale (0.9,1), ratio (3/4,4/3), bicubic) + RandomHorizontalFlip + ToTensor + Normalize
val  : Resize(n, bicubic) + CenterCrop(n) + ToTensor + Normalize
Randomness comes from torch's global CPU generator, as torchvision's does.
"""
import math

import numpy as np
import torch

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def _to_tensor_normalize(img):
    a = np.asarray(img.convert('RGB'), dtype=np.float32) / 255.0
    t = torch.from_numpy(a).permute(2, 0, 1).contiguous()
    return (t - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)


def _rrc_params(w, h, scale, ratio):
    area = h * w
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect = math.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1]).item())
        cw, ch = int(round(math.sqrt(target_area * aspect))), int(round(math.sqrt(target_area / aspect)))
        if 0 < cw <= w and 0 < ch <= h:
            i = torch.randint(0, h - ch + 1, size=(1,)).item()
            j = torch.randint(0, w - cw + 1, size=(1,)).item()
            return i, j, ch, cw
    in_ratio = w / h
    if in_ratio < ratio[0]:
        cw, ch = w, int(round(w / ratio[0]))
    elif in_ratio > ratio[1]:
        ch, cw = h, int(round(h * ratio[1]))
    else:
        cw, ch = w, h
    return (h - ch) // 2, (w - cw) // 2, ch, cw


def build_transform(n_px, mode='train'):
    from PIL import Image

    def train(img):
        w, h = img.size
        i, j, ch, cw = _rrc_params(w, h, (0.9, 1.0), (0.75, 1.3333))
        img = img.crop((j, i, j + cw, i + ch)).resize((n_px, n_px), Image.BICUBIC)
        if torch.rand(1).item() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        return _to_tensor_normalize(img)

    def val(img):
        w, h = img.size
        s = n_px / min(w, h)
        nw, nh = (n_px, max(n_px, int(round(h * s)))) if w <= h else (max(n_px, int(round(w * s))), n_px)
        img = img.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))
        return _to_tensor_normalize(img.crop((left, top, left + n_px, top + n_px)))

    return train if mode == 'train' else val
