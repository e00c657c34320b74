"""Dev tool: every row-count class of the tiled attention kernels (64 < T <= 320) against torch autograd, over poisoned memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import _lib
lib = _lib.lib(); dev = torch.device("cuda:0"); BF = torch.bfloat16
bad = 0
for T in (65, 66, 79, 80, 81, 96, 112, 127, 128, 129, 143, 144, 145, 160, 191, 192, 193, 208, 240, 255, 256, 257, 272, 300, 319, 320):
    for B, H in ((1, 1), (3, 5)):
        for causal in (0, 1):
            poison = torch.full((32 << 20,), -1, dtype=torch.int8, device=dev); del poison
            d = 64 * H
            g = torch.Generator(device=dev).manual_seed(T * 7 + B)
            qkv = torch.randn(B * T, 3 * d, device=dev, generator=g).to(BF)
            do = torch.randn(B * T, d, device=dev, generator=g).to(BF)
            out = torch.full((B * T, d), float("nan"), device=dev, dtype=BF)
            dqkv = torch.full((B * T, 3 * d), float("nan"), device=dev, dtype=BF)
            lse = torch.full((B * H, T), float("nan"), device=dev)
            _lib.check(lib.swat_attn_fwd_tiled(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(lse), B, T, d, H, causal, _lib.stream()), "fwd")
            q, k, v = [t.reshape(B, T, H, 64).permute(0, 2, 1, 3).float().requires_grad_(True) for t in qkv.float().split(d, dim=1)]
            s = (q @ k.transpose(-1, -2)) / 8.0
            if causal:
                s = s.masked_fill(torch.ones(T, T, device=dev).triu(1).bool(), float("-inf"))
            o = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, d)
            e1 = (out.float() - o).abs().max().item() / (o.abs().max().item() + 1e-6)
            e2 = (lse - torch.logsumexp(s, -1).reshape(B * H, T)).abs().max().item()
            e3 = 0.0
            if not causal:
                ws = torch.empty(lib.swat_attn_bwd_tiled_workspace_bytes(B, T, H), device=dev, dtype=torch.uint8)
                _lib.check(lib.swat_attn_bwd_tiled(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(do), _lib.ptr(lse), _lib.ptr(dqkv), B, T, d, H, _lib.ptr(ws), _lib.stream()), "bwd")
                o.backward(do.float())
                ref = torch.cat([t.grad.permute(0, 2, 1, 3).reshape(B * T, d) for t in (q, k, v)], dim=1)
                e3 = (dqkv.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
            ok = e1 <= 2.0 ** -7 + 1e-3 and e2 < 5e-3 and e3 <= 0.03 and not (out != out).any().item() and (causal or not (dqkv != dqkv).any().item())
            if not ok:
                bad += 1
                print("BAD", T, B, H, causal, e1, e2, e3, flush=True)
print("bad:", bad)
