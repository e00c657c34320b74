#!/usr/bin/env python3
"""tools/mega_bwd_check.py — the persistent backward of the blocks (csrc/vit_mega.hip: vit_bwd_mega_kernel) against the op-by-op
backward on ViT-B/32: every gradient of the arena, the hand-off status, repeatability, the time of both and (--trace) its phases."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swat_amd import _lib  # noqa: E402
from swat_amd.clip_model import CLIP  # noqa: E402


def run(model, x, dfeats, mega_bwd):
    L = _lib.lib()
    L.swat_vit_set_mega(1)
    L.swat_vit_set_mega_backward(mega_bwd)
    v = model.visual
    v.forward_raw(x, save=True)
    a = v._last_ws[3]
    if a.grad is not None:
        a.grad.zero_()
    n0 = L.swat_vit_mega_backward_launches()
    v.backward_raw(dfeats)
    torch.cuda.synchronize()
    return a.grad.clone(), L.swat_vit_mega_backward_launches() - n0


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = CLIP("ViT-B-32-quickgelu").to(dev)
    x = torch.randn(B, 3, 224, 224, device=dev)
    dfeats = torch.randn(B, model.visual.output_dim, device=dev) * 0.05
    L = _lib.lib()
    if B not in (29, 30, 31, 32, 61, 62, 63, 64):
        L.swat_vit_set_mega(2)
    g0, n0 = run(model, x, dfeats, 0)
    g1, n1 = run(model, x, dfeats, 1)
    print(f"B={B}: persistent backward launches: op-by-op run {n0}, persistent run {n1}; status {L.swat_vit_mega_status():#x}")
    v = model.visual
    a = v._last_ws[3]
    worst = 0.0
    for name, p in v.named_parameters():
        lo = a.offset_of(p) if hasattr(a, "offset_of") else None
        if lo is None:
            break
        n = p.numel()
        d0, d1 = g0[lo:lo + n], g1[lo:lo + n]
        rel = ((d0 - d1).norm() / (d0.norm() + 1e-30)).item()
        worst = max(worst, rel)
        if rel > 2e-3 or "--all" in sys.argv:
            print(f"  {name:50s} |g| {d0.norm().item():.3e}  rel diff {rel:.3e}")
    tot = ((g0 - g1).norm() / g0.norm()).item()
    print(f"whole arena: rel diff {tot:.3e}; worst tensor {worst:.3e}; finite {bool(torch.isfinite(g1).all())}")
    bad = 0
    for _ in range(10):
        g2, _ = run(model, x, dfeats, 1)
        bad += int(not torch.equal(g2, g1))
    print("repeat mismatches:", bad, "status", hex(L.swat_vit_mega_status()))
    for mb in (0, 1):
        L.swat_vit_set_mega_backward(mb)
        v.forward_raw(x, save=True)
        for _ in range(5):
            v.backward_raw(dfeats)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            v.backward_raw(dfeats)
        torch.cuda.synchronize()
        print(f"mega_bwd={mb}: backward {1e3 * (time.perf_counter() - t0) / n:.3f} ms")
    if "--trace" in sys.argv:
        L.swat_vit_set_mega_backward(1)
        tr = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
        L.swat_vit_mega_backward_trace(ctypes.c_void_p(tr.data_ptr()))
        v.backward_raw(dfeats)
        torch.cuda.synchronize()
        L.swat_vit_mega_backward_trace(None)
        t = tr.view(256, 64).cpu()
        names = ["entry", "blocks above", "PRJ'", "FC'", "LN2'", "OUT'+attn", "IN'", "LN1'", "exit"]
        for i in range(1, 9):
            d = (t[:, i] - t[:, i - 1]).float() / 100.0
            print(f"  {names[i]:12s}: median {d.median():7.2f} us  min {d.min():7.2f}  max {d.max():7.2f}" + ("" if i in (1, 8) else "  (middle block, incl. hand-off)"))
        print("  total median %.1f us" % (float((t[:, 8] - t[:, 0]).float().median()) / 100.0))
        base = t[:, 0].min()
        names = names[1:]
        for k in range(6):
            s4 = (t[:, 8 + 4 * k: 12 + 4 * k] - base).double() / 100.0
            start = (t[:, k + 1] - base).double() / 100.0
            line = f"  {names[k + 1]:10s} work {float((s4[:, 0] - start).median()):6.2f}  drain {float((s4[:, 1] - s4[:, 0]).median()):5.2f}  wait-for-team {float((s4[:, 2] - s4[:, 1]).median()):5.2f} (min {float((s4[:, 2] - s4[:, 1]).min()):5.2f})"
            if k in (0, 1, 3, 4):
                gd = (t[:, 32 + k] - base).double() / 100.0
                sel = gd > 0
                if sel.any():
                    line += f"   gemm {float((gd - start)[sel].median()):6.2f}  rest {float((s4[:, 0] - gd)[sel].median()):5.2f}"
            print(line)


if __name__ == "__main__":
    main()
