"""Dev tool: repeat-run bit equality of the tiled attention forward (with / without the lse output) at the ViT-B/16 and ViT-L/14
shapes — how the inline-asm MFMA hazard of round 3 was pinned down."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from swat_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
for B, H, T in ((2, 12, 197), (2, 16, 257), (32, 16, 257)):
    d = 64 * H
    qkv = torch.randn(B * T, 3 * d, device=dev).bfloat16()
    outs = []
    for lse_on in (1, 0, 1, 0):
        out = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(B * H, T, device=dev)
        _lib.check(L.swat_attn_fwd_tiled(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(lse) if lse_on else None, B, T, d, H, 0, _lib.stream()), "fwd")
        torch.cuda.synchronize()
        outs.append(out)
    print(B, H, T, [torch.equal(outs[0], o) for o in outs[1:]], [(outs[0].float() - o.float()).abs().max().item() for o in outs[1:]])
