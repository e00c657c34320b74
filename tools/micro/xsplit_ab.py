#!/usr/bin/env python3
"""tools/micro/xsplit_ab.py — the persistent forward / backward of ViT-B/32 at B = 32 under the three team <-> XCD placements
(swat_vit_set_mega_xsplit: 1, 2, 4 XCDs per team), alternating in one process; bytes must not depend on the placement."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swat_amd import _lib  # noqa: E402
from swat_amd.clip_model import CLIP  # noqa: E402


def main():
    B = 32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = CLIP("ViT-B-32-quickgelu").to(dev)
    x = torch.randn(B, 3, 224, 224, device=dev)
    dfeats = torch.randn(B, model.visual.output_dim, device=dev) * 0.05
    L = _lib.lib()
    v = model.visual
    ref = None
    for rep in range(3):
        for xs in (1, 2, 4, 8):
            L.swat_vit_set_mega_xsplit(xs)
            for _ in range(5):
                f = v.forward_raw(x, save=True)
                v.backward_raw(dfeats)
            torch.cuda.synchronize()
            n = 40
            t0 = time.perf_counter()
            for _ in range(n):
                f = v.forward_raw(x, save=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                v.forward_raw(x, save=True)
                v.backward_raw(dfeats)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            g = v._last_ws[3].grad
            key = (f.clone(), g.clone())
            if ref is None:
                ref = key
            same = torch.equal(ref[0], key[0]) and torch.equal(ref[1], key[1])
            fw = 1e3 * (t1 - t0) / n
            print(f"xsplit={xs}: forward {fw:.3f} ms, forward+backward {1e3 * (t2 - t1) / n:.3f} ms (backward {1e3 * (t2 - t1) / n - fw:.3f}); bytes equal to xsplit=1: {same}; status {L.swat_vit_mega_status():#x}")


if __name__ == "__main__":
    main()
