#!/usr/bin/env python3
"""tools/micro/adamw_under_forward.py — can the optimizer run UNDER the next step's persistent forward?  The forward (csrc/vit_mega.hip)
holds every CU with 156 KB of LDS and 8 waves; built with a 224-register budget (MG_FWD_VGPR=112) it leaves 64 registers per SIMD — one
56-register AdamW wave.  Times the ViT-B/32 forward at B = 32 alone, swat_adamw over an 88 M-parameter arena alone, and both at once on
two streams (a separate arena: no data hazard, timing only)."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swat_amd import _lib  # noqa: E402
from swat_amd.clip_model import CLIP  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = CLIP("ViT-B-32-quickgelu").to(dev)
v = model.visual
x = torch.randn(32, 3, 224, 224, device=dev)
n = 85_000_000 // 1024 * 1024
p, g, m, vv = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
vv.abs_()
sh = torch.empty(n, dtype=torch.bfloat16, device=dev)
lrs = torch.tensor([1e-6], device=dev)
bc = torch.tensor([0.1, 0.001], device=dev)
L = _lib.lib()
s_opt = torch.cuda.Stream()
nblk = 12
per = n // nblk // 1024 * 1024


def adamw_chain():
    for b in range(nblk):
        seg = (ctypes.c_int64 * 2)(0, per)
        o = b * per
        _lib.check(L.swat_adamw(p.data_ptr() + 4 * o, g.data_ptr() + 4 * o, m.data_ptr() + 4 * o, vv.data_ptr() + 4 * o, sh.data_ptr() + 2 * o, per, seg, 1,
                                lrs.data_ptr(), 0.9, 0.999, 1e-8, 0.01, bc.data_ptr(), 1.0, s_opt.cuda_stream), "adamw")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def fwd():
    v.forward_raw(x, save=True)


def both():
    s_opt.wait_stream(torch.cuda.current_stream())
    adamw_chain()
    v.forward_raw(x, save=True)
    torch.cuda.current_stream().wait_stream(s_opt)


def serial():
    adamw_chain()
    torch.cuda.current_stream().wait_stream(s_opt)
    v.forward_raw(x, save=True)
    s_opt.wait_stream(torch.cuda.current_stream())


print(f"forward alone      {timed(fwd):.3f} ms")
print(f"adamw chain alone  {timed(lambda: (adamw_chain(), torch.cuda.current_stream().wait_stream(s_opt))):.3f} ms")
print(f"one after the other {timed(serial):.3f} ms")
print(f"both at once (optimizer queued first)  {timed(both):.3f} ms   status {L.swat_vit_mega_status():#x}")


def both2():   # the forward's workgroups are resident first; the optimizer's find one wave slot per SIMD beside them
    v.forward_raw(x, save=True)
    adamw_chain()
    torch.cuda.current_stream().wait_stream(s_opt)
    s_opt.wait_stream(torch.cuda.current_stream())


print(f"both at once (forward queued first)    {timed(both2):.3f} ms   status {L.swat_vit_mega_status():#x}")
