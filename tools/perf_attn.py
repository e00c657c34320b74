"""Dev tool: the tiled attention kernels (64 < T <= 320) alone, at the ViT-L/14 and ViT-B/16 shapes of a B = 32 step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
for name, B, H, T in (("ViT-L/14", 32, 16, 257), ("ViT-B/16", 32, 12, 197)):
    d = 64 * H
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn(B * T, 3 * d, device=dev, generator=g).bfloat16()
    do = torch.randn(B * T, d, device=dev, generator=g).bfloat16()
    out = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
    dqkv = torch.empty(B * T, 3 * d, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * H, T, device=dev)
    ws = torch.empty(B * H * T * 4 + 1024, device=dev, dtype=torch.uint8)
    s = _lib.stream()

    def fwd():
        _lib.check(L.swat_attn_fwd_tiled(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(lse), B, T, d, H, 0, s), "fwd")

    def bwd():
        _lib.check(L.swat_attn_bwd_tiled(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(do), _lib.ptr(lse), _lib.ptr(dqkv), B, T, d, H, _lib.ptr(ws), s), "bwd")

    res = []
    for fn in (fwd, bwd):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 4.0 * T * T * 64 * B * H
    print(f"{name}: T={T} B*H={B * H}  fwd {res[0]:7.1f} us ({fl / res[0] / 1e6:5.0f} TF)   bwd {res[1]:7.1f} us ({2.5 * fl / res[1] / 1e6:5.0f} TF)")
