"""Baseline-JPEG decode on host threads + GPU (include/swat_hip.h `swat_jpeg_*`, csrc/jpeg.hip): the loaders' per-image
`Image.open(path).convert('RGB')` (ref: utils/datasets/dataset_utils.py:130-180, retrieval/extract_mined_feature.py:103-117) for a
whole batch, bytes identical to Pillow's.  Files of OTHER FORMATS (progressive / arithmetic-coded / CMYK JPEG, unusual sampling, PNG,
...) are decoded by the reference's own decoder, Pillow, on the host and uploaded into the same buffer, so a caller always gets
every image — counted in `stats`, announced once, and refused with SWAT_JPEG_STRICT=1.  This is format coverage, not a fallback for
the HIP library: without libswat_hip.so every call here raises.
"""
import os
import warnings
import ctypes
import io

import numpy as np
import torch

from . import _lib


def info(data):
    """(width, height) of a baseline JPEG this path decodes, else None.  `data`: bytes / uint8 array of the file."""
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    w, h = ctypes.c_int32(), ctypes.c_int32()
    rc = _lib.lib().swat_jpeg_info(a.ctypes.data_as(ctypes.c_void_p), a.size, ctypes.byref(w), ctypes.byref(h))
    return (w.value, h.value) if rc == 0 else None


stats = {"device": 0, "host_other_format": 0}   # images decoded by the HIP path / handed to Pillow because of their format
_warned = False


def _pil_rgb(data):
    global _warned
    if os.environ.get("SWAT_JPEG_STRICT", "0") == "1":
        raise _lib.SwatHipError("swat_amd.jpeg: a file is not a baseline JPEG the HIP decoder takes (SWAT_JPEG_STRICT=1)")
    if not _warned:
        warnings.warn("swat_amd.jpeg: a file is not a baseline JPEG the HIP decoder takes (progressive / CMYK / another format): "
                      "decoded by Pillow on the host — counted in swat_amd.jpeg.stats")
        _warned = True
    stats["host_other_format"] += 1
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(bytes(data))).convert('RGB'), dtype=np.uint8)


def decode_batch(files, device='cuda', sizes=None, threads=0, stream=None, tail_bytes=8):
    """files: list of bytes / uint8 arrays (whole image files).  -> (pixels uint8 [total + tail] on `device`: image i = HWC RGB at byte
    offsets[i]; offsets int64 numpy [n]; shapes [(h, w)]).  sizes: optional [(w, h)] the caller already parsed (swat_amd.jpeg.info)."""
    lib = _lib.lib()
    dev = torch.device(device)
    arrs = [np.frombuffer(f, dtype=np.uint8) if not isinstance(f, np.ndarray) else np.ascontiguousarray(f, dtype=np.uint8) for f in files]
    n = len(arrs)
    fallback = {}
    shapes = []
    for i, a in enumerate(arrs):
        wh = sizes[i] if sizes is not None and sizes[i] is not None else info(a)
        if wh is None:
            fallback[i] = _pil_rgb(a)
            wh = (fallback[i].shape[1], fallback[i].shape[0])
        shapes.append((wh[1], wh[0]))
    offsets = np.zeros(n, np.int64)
    total = 0
    for i, (h, w) in enumerate(shapes):
        offsets[i] = total
        total += 3 * h * w
    pixels = torch.empty(total + tail_bytes, dtype=torch.uint8, device=dev)
    if n == 0:
        return pixels, offsets, shapes
    ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data if i not in fallback else None for i, a in enumerate(arrs)])
    nbytes = (ctypes.c_int64 * n)(*[a.size for a in arrs])
    status = (ctypes.c_int32 * n)()
    with torch.cuda.device(dev):
        s = stream if stream is not None else torch.cuda.current_stream(dev)
        with torch.cuda.stream(s):
            _lib.check(lib.swat_jpeg_decode_batch(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(nbytes, ctypes.c_void_p), n, _lib.ptr(pixels),
                                                  offsets.ctypes.data_as(ctypes.c_void_p), ctypes.cast(status, ctypes.c_void_p), int(threads),
                                                  s.cuda_stream), "swat_jpeg_decode_batch")
            for i in range(n):
                if status[i] != 0 and i not in fallback:   # the headers parsed, the scan did not (or the caller's sizes were wrong)
                    fallback[i] = _pil_rgb(arrs[i])
                    if fallback[i].shape[:2] != shapes[i]:
                        raise _lib.SwatHipError(f"image {i}: size {fallback[i].shape[:2]} differs from the announced {shapes[i]}")
            stats["device"] += n - len(fallback)
            for i, a in fallback.items():
                pixels[offsets[i]:offsets[i] + a.size].copy_(torch.from_numpy(np.ascontiguousarray(a).reshape(-1)), non_blocking=False)
    return pixels, offsets, shapes


def decode(data, device='cuda'):
    """One file -> uint8 [H, W, 3] tensor on the device."""
    pixels, offsets, shapes = decode_batch([data], device)
    h, w = shapes[0]
    return pixels[:3 * h * w].view(h, w, 3)
