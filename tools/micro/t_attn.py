import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from swat_amd import _lib
lib = _lib.lib()
torch.manual_seed(0)
for (B, H, T, causal) in [(2, 3, 65, 0), (2, 12, 197, 0), (1, 16, 257, 0), (3, 8, 77, 1), (2, 2, 320, 0), (1, 2, 128, 1), (2, 4, 50, 0)]:
    d = 64 * H
    qkv = torch.randn(B * T, 3 * d, device="cuda").bfloat16()
    do = torch.randn(B * T, d, device="cuda").bfloat16()
    out = torch.full((B * T, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    dqkv = torch.full((B * T, 3 * d), float("nan"), device="cuda", dtype=torch.bfloat16)
    lse = torch.full((B * H, T), float("nan"), device="cuda")
    _lib.check(lib.swat_attn_fwd_tiled(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(lse), B, T, d, H, causal, _lib.stream()), "fwd")
    q, k, v = [t.reshape(B, T, H, 64).permute(0, 2, 1, 3).float().requires_grad_(True) for t in qkv.float().split(d, dim=1)]
    s = (q @ k.transpose(-1, -2)) / 8.0
    if causal:
        s = s.masked_fill(torch.ones(T, T, device="cuda").triu(1).bool(), float("-inf"))
    p = torch.softmax(s, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * T, d)
    err = (out.float() - o).abs().max().item() / o.abs().max().item()
    lerr = (lse - torch.logsumexp(s, -1).reshape(B * H, T)).abs().max().item()
    msg = f"{B} {H} {T} causal={causal}: out rel err {err:.2e}  lse err {lerr:.2e}"
    if not causal:
        _lib.check(lib.swat_attn_bwd_tiled(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(do), _lib.ptr(lse), _lib.ptr(dqkv), B, T, d, H, _lib.stream()), "bwd")
        o.backward(do.float())
        ref = torch.cat([g_.grad.permute(0, 2, 1, 3).reshape(B * T, d) for g_ in (q, k, v)], dim=1)
        parts = [(dqkv.float()[:, i * d:(i + 1) * d] - ref[:, i * d:(i + 1) * d]).abs().max().item() / ref.abs().max().item() for i in range(3)]
        msg += "  dq/dk/dv rel err " + " ".join(f"{x:.2e}" for x in parts)
    print(msg)
