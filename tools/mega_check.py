#!/usr/bin/env python3
"""tools/mega_check.py — the persistent forward (csrc/vit_mega.hip) against the op-by-op forward on ViT-B/32: features,
every saved activation of the workspace, the hand-off status word, and the time of both (forward only and whole step)."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swat_amd import _lib  # noqa: E402
from swat_amd.clip_model import CLIP  # noqa: E402


def fwd(model, x, mega, save=True, perm=None, box=None):
    L = _lib.lib()
    L.swat_vit_set_mega(mega)
    model.visual._ws = {}  # fresh, zeroed workspace
    f = model.visual.forward_raw(x, save=save, perm=perm, box=box)
    torch.cuda.synchronize()
    ws = model.visual._ws[(x.shape[0], save)].clone()
    return f.clone(), ws


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = CLIP("ViT-B-32-quickgelu").to(dev)
    x = torch.randn(B, 3, 224, 224, device=dev)
    L = _lib.lib()
    f0, w0 = fwd(model, x, 0)
    f1, w1 = fwd(model, x, 1)
    st = L.swat_vit_mega_status()
    print(f"B={B} status={st:#x}")
    print("feats: max|d| = %.3e  (max|f| = %.3e)" % ((f0 - f1).abs().max().item(), f0.abs().max().item()))
    nb = w0.numel()
    diff = (w0 != w1)
    print(f"workspace bytes {nb}, differing bytes {int(diff.sum())} ({100.0 * diff.float().mean().item():.4f} %)")
    if diff.any():
        idx = diff.nonzero().flatten()
        print("first / last differing byte:", int(idx[0]), int(idx[-1]))
    # round 6: fixed wave roles + register-carried weight stages against the kernel of rounds 3-5: the same bytes
    if hasattr(L, "swat_vit_set_mega_roles"):
        L.swat_vit_set_mega_roles(1)
        f1o, w1o = fwd(model, x, 1)
        L.swat_vit_set_mega_roles(2)
        print("roles 2 vs roles 1: feats equal", bool(torch.equal(f1o, f1)), " workspace equal", bool(torch.equal(w1o, w1)), "status", hex(L.swat_vit_mega_status()))
    # repeatability of the persistent forward (hand-offs under load): 20 runs, identical bytes
    bad = 0
    for _ in range(20):
        f2, w2 = fwd(model, x, 1)
        bad += int(not torch.equal(w2, w1)) + int(not torch.equal(f2, f1))
    print("repeat mismatches:", bad, "status", hex(L.swat_vit_mega_status()))
    for mega in (0, 1, 2):
        L.swat_vit_set_mega(min(mega, 1))
        if hasattr(L, "swat_vit_set_mega_roles"):
            L.swat_vit_set_mega_roles(1 if mega == 1 else 2)
        for _ in range(5):
            model.visual.forward_raw(x, save=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            model.visual.forward_raw(x, save=True)
        torch.cuda.synchronize()
        print(f"mega={min(mega, 1)} roles={1 if mega == 1 else 2}: forward {1e3 * (time.perf_counter() - t0) / n:.3f} ms")
    if "--trace" in sys.argv:
        L.swat_vit_set_mega(1)
        if hasattr(L, "swat_vit_set_mega_roles"):
            L.swat_vit_set_mega_roles(2 if "--roles2" in sys.argv else 1)
        tr = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
        L.swat_vit_mega_trace(ctypes.c_void_p(tr.data_ptr()))
        model.visual.forward_raw(x, save=True)
        torch.cuda.synchronize()
        L.swat_vit_mega_trace(None)
        t = tr.view(256, 64).cpu()
        base = t[:, 0].min()
        names = ["entry", "LN1", "QKV+attn", "OUT", "LN2", "FC", "PRJ", "exit"]
        for i in range(1, 8):
            d = (t[:, i] - t[:, i - 1]).float() / 100.0   # us
            print(f"  {names[i]:9s}: median {d.median():7.2f} us  min {d.min():7.2f}  max {d.max():7.2f}" + ("   (whole rest of the tower)" if i == 7 else "  (layer 0)"))
        ph = ["LN1", "QKV", "OUT", "LN2", "FC", "PRJ"]
        for k in range(6):
            s4 = (t[:, 8 + 4 * k: 12 + 4 * k] - base).double() / 100.0
            start = (t[:, k] - base).double() / 100.0   # phase start = previous stamp
            line = f"  {ph[k]:4s} work {float((s4[:, 0] - start).median()):6.2f}  drain {float((s4[:, 1] - s4[:, 0]).median()):5.2f}  wait-for-team {float((s4[:, 2] - s4[:, 1]).median()):5.2f} (min {float((s4[:, 2] - s4[:, 1]).min()):5.2f})  acquire {float((s4[:, 3] - s4[:, 2]).median()):5.2f}"
            gi = {1: 0, 2: 1, 4: 2, 5: 3}.get(k)
            if gi is not None:
                gd = (t[:, 32 + gi] - base).double() / 100.0
                line += f"   gemm {float((gd - start).median()):6.2f}  epilogue {float((s4[:, 0] - gd).median()):5.2f}"
            print(line)
        print("  entry spread %.2f us, total median %.1f us" % (float(t[:, 0].max() - base) / 100.0, float((t[:, 7] - t[:, 0]).float().median()) / 100.0))


if __name__ == "__main__":
    main()
