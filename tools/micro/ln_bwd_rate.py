#!/usr/bin/env python3
"""tools/micro/ln_bwd_rate.py — swat_layernorm_bwd alone at the many-row shapes (ViT-B/16: 6 304 x 768, ViT-L/14: 8 224 x 1 024) and the
ViT-B/32 shape: time per call and bytes moved / time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swat_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
p = lambda t: t.data_ptr()
for M, d in ((1600, 768), (6304, 768), (8224, 1024)):
    dy = torch.randn(M, d, device=dev); x = torch.randn(M, d, device=dev); g = torch.randn(d, device=dev)
    mean = torch.randn(M, device=dev); rstd = torch.rand(M, device=dev) + 0.5
    dx = torch.randn(M, d, device=dev); dxb = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    dg = torch.empty(d, device=dev); db = torch.empty(d, device=dev); dcs = torch.empty(d, device=dev)
    ws = torch.empty(lib.swat_layernorm_bwd_workspace_bytes(M, d), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    def call():
        _lib.check(lib.swat_layernorm_bwd(p(dy), d, p(x), d, p(g), p(mean), p(rstd), p(dx), d, p(dx), d, p(dxb), d, p(dg), p(db), p(dcs), p(ws), M, d, s), "ln_bwd")
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    byts = M * d * (3 * 4 + 4 + 2)
    print(f"rows {M:5d} x {d}: {us:6.1f} us per call (incl. the reduce launch)   {byts / us / 1e6:.2f} TB/s")
