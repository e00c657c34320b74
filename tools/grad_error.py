"""Dev tool: measured error of the HIP vision tower against the fp32 oracle per architecture (what tests/test_gpu_vit.py's
tolerances are set from): embedding max-abs error / max-abs, worst per-tensor gradient relative L2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import test_gpu_vit as T
for name, cfg, B in [("TINY", T.TINY, 5), ("TINY_GELU", T.TINY_GELU, 3), ("B32", T.B32, 4), ("B16", T.B16, 2), ("L14", T.L14, 2)]:
    worst_e, worst_g, worst_n = 0.0, 0.0, ""
    for seed in (0, 1):
        ref, mine = T.make_pair(cfg, seed=seed)
        g = torch.Generator().manual_seed(1 + seed)
        x = torch.randn(B, 3, cfg[0], cfg[0], generator=g); w = torch.randn(B, cfg[5], generator=g)
        f_ref = ref(x); (f_ref * w).sum().backward()
        f = mine.forward_raw(x.to(T.dev()), save=True)
        worst_e = max(worst_e, (f.cpu() - f_ref.detach()).abs().max().item() / f_ref.abs().max().item())
        mine.backward_raw(w.to(T.dev()).contiguous())
        arena = mine.arena(); rp = dict(ref.named_parameters())
        for n, p in mine.named_parameters():
            o = arena.offset_of(p)
            r = T.rel(arena.grad[o:o + p.numel()].view(p.shape), rp[n].grad)
            if r > worst_g: worst_g, worst_n = r, n
    print(f"{name:10s} embedding err / max-abs {worst_e:.4f}   worst gradient relative L2 {worst_g:.4f} ({worst_n})", flush=True)
