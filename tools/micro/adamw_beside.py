#!/usr/bin/env python3
"""tools/micro/adamw_beside.py — swat_adamw over the ViT-B/32 arena (88 M parameters) alone, and beside a kernel that only occupies
every CU the way the weight-gradient launch does (tools/micro/_hog/hog.hip: 12 waves, 144 KB LDS, 152 / 128 / 104 registers).
Build the occupant first (here, it travels to the GPU box):
    (cd tools/micro/_hog && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o libhog.so hog.hip)"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swat_amd import _lib  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
hog = ctypes.CDLL(os.path.join(here, "_hog", "libhog.so"))
hog.launch_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
n = 88_000_000 // 1024 * 1024
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v.abs_()
sh = torch.empty(n, dtype=torch.bfloat16, device=dev)
lrs = torch.tensor([1e-6], device=dev)
bc = torch.tensor([0.1, 0.001], device=dev)
seg = (ctypes.c_int64 * 2)(0, n)
L = _lib.lib()
s_opt, s_hog = torch.cuda.Stream(), torch.cuda.Stream()


def adamw():
    _lib.check(L.swat_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr(), n, seg, 1, lrs.data_ptr(), 0.9, 0.999, 1e-8,
                            0.01, bc.data_ptr(), 1.0, s_opt.cuda_stream), "adamw")


for regs, blocks in ((0, 0), (152, 256), (128, 256), (104, 256), (152, 216), (128, 216), (0, 0)):
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        if regs:
            hog.launch_hog(regs, 2500, blocks, s_hog.cuda_stream)
            time.sleep(0.0005)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s_opt)
        adamw()
        e1.record(s_opt)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[1:])
    print(f"hog regs={regs:3d} blocks={blocks:3d}: adamw {ts[len(ts) // 2]:.3f} ms  ({30 * n / ts[len(ts) // 2] / 1e9:.2f} TB/s)  all {[round(t, 3) for t in ts]}")
