"""oracle/image_oracle.py — TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

CPU restatement of the image side of the reference's input pipeline:
  train: RandomResizedCrop(224, scale (0.9, 1), ratio (3/4, 4/3), bicubic) + RandomHorizontalFlip + ToTensor + Normalize
         (ref: utils/extras.py:66-77)
  val  : Resize(224, bicubic) + CenterCrop(224) + ToTensor + Normalize (ref: utils/extras.py:78-85, open_clip preprocess)
The arithmetic lives in a third-party dependency that is not under /root/reference: torchvision hands PIL images to
Pillow (`Pillow` in requirements.txt; 12.2.0 in this image), whose 8-bit resampler is restated here from its published
algorithm: per axis a table of antialiased bicubic (a = -0.5) weights in double precision, normalised, rounded to 22-bit
fixed point; a horizontal pass then a vertical pass, each accumulating in int32 from 1 << 21 and rounding to uint8.
Pinned: tests/test_oracle_image.py checks this file against Pillow itself (installed here and on the GPU box) on random
images for down-scaling, up-scaling, identity axes and one-pixel-wide crops — bit-exact.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def ksize_of(in_size, out_size):
    filterscale = max(float(in_size) / out_size, 1.0)
    return int(math.ceil(2.0 * filterscale)) * 2 + 1


def axis_table(in_size, out_size):
    """-> bounds int32 [out, 2] (first source index, count), weights int32 [out, ksize]"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ks = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ks), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(src, bounds, kk):
    """resample axis 1 of src uint8 [rows, in, 3] -> uint8 [rows, out, 3]"""
    out = np.empty((src.shape[0], bounds.shape[0], 3), np.uint8)
    s = src.astype(np.int64)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(s[:, x0:x0 + n, :], kk[xx, :n].astype(np.int64), axes=([1], [0]))
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize(img, out_w, out_h):
    """Pillow's Image.resize((out_w, out_h), BICUBIC) on a uint8 [H, W, 3] array"""
    h, w = img.shape[:2]
    bh, kh = axis_table(w, out_w)
    bv, kv = axis_table(h, out_h)
    first, last = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
    if out_w != w:
        img = _pass(img[first:last], bh, kh)
        bv = bv.copy()
        bv[:, 0] -= first
    if out_h != h:
        img = _pass(img.transpose(1, 0, 2), bv, kv).transpose(1, 0, 2)
    return np.ascontiguousarray(img)


def normalize_lut():
    """float32 [3, 256]: (v / 255 - mean) / std with every step rounded to float32, as ToTensor + Normalize do"""
    v = np.arange(256, dtype=np.float32) / np.float32(255.0)
    return np.stack([(v - np.float32(MEAN[c])) / np.float32(STD[c]) for c in range(3)]).astype(np.float32)


def val_geometry(w, h, n_px):
    s = n_px / min(w, h)
    nw, nh = (n_px, max(n_px, int(round(h * s)))) if w <= h else (max(n_px, int(round(w * s))), n_px)
    return nw, nh, int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))


def apply(img, n_px, box=None, flip=False, mode='train'):
    """uint8 [H, W, 3] -> float32 [3, n_px, n_px].  train: box = (top, left, crop_h, crop_w)"""
    if mode == 'train':
        i, j, ch, cw = box
        out = resize(img[i:i + ch, j:j + cw], n_px, n_px)
        if flip:
            out = out[:, ::-1]
    else:
        h, w = img.shape[:2]
        nw, nh, left, top = val_geometry(w, h, n_px)
        out = resize(img, nw, nh)[top:top + n_px, left:left + n_px]
    lut = normalize_lut()
    return np.stack([lut[c][out[:, :, c]] for c in range(3)])
