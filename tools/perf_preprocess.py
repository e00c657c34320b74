"""Times the GPU image transform (swat_preprocess_batch) beside the per-image Pillow path a DataLoader worker runs.
usage: python tools/perf_preprocess.py [B] [H] [W]"""
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from swat_amd import _lib, transforms as T  # noqa: E402

B, H, W = (int(x) for x in (sys.argv[1:4] + ["32", "375", "500"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(B)]
for mode in ("train", "val"):
    tf = T.DeviceBatchTransform(224, mode)
    torch.manual_seed(0)
    geo = [tf.geometry(W, H) for _ in range(B)]
    for _ in range(3):
        out = tf(imgs, geometry=geo)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = tf(imgs, geometry=geo)
    torch.cuda.synchronize()
    whole = (time.perf_counter() - t0) / 10
    # kernels alone, inputs resident
    lib = _lib.lib()
    desc = np.array([(W, H) + g[:8] + (g[8], 0) for g in geo], np.int32)
    offs = torch.tensor([i * H * W * 3 for i in range(B)], dtype=torch.int64, device="cuda")
    rows = torch.tensor(np.concatenate([[0], np.cumsum(desc[:, 5])]), dtype=torch.int64, device="cuda")
    d = torch.from_numpy(desc).cuda()
    pix = torch.from_numpy(np.concatenate([a.reshape(-1) for a in imgs] + [np.zeros(4, np.uint8)])).cuda()
    ks = max(max(T._axis_ksize(g[2], g[4]), T._axis_ksize(g[3], g[5])) for g in geo)
    ws = torch.empty(lib.swat_preprocess_workspace_bytes(B, 224, ks, int(rows[-1])), dtype=torch.uint8, device="cuda")
    o = torch.empty(B, 3, 224, 224, device="cuda")
    s = torch.cuda.current_stream()

    def run():
        _lib.check(lib.swat_preprocess_batch(pix.data_ptr(), offs.data_ptr(), d.data_ptr(), rows.data_ptr(), tf._lut.data_ptr(), o.data_ptr(),
                                             B, 224, int(desc[:, 4].max()), int(desc[:, 5].max()), ks, ws.data_ptr(), ws.numel(), s.cuda_stream), "pp")
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    kern = e0.elapsed_time(e1) / 50 * 1e-3
    assert torch.equal(o, out)
    nbytes = sum(3 * g[2] * g[3] for g in geo) + B * 12 * 224 * 224
    host_tf = T.build_transform(224, mode)
    pil = [Image.fromarray(a) for a in imgs[:8]]
    t0 = time.perf_counter()
    for im in pil:
        host_tf(im)
    host = (time.perf_counter() - t0) / len(pil)
    print(f"{mode}: B={B} {H}x{W}  kernels {kern * 1e6:.1f} us = {B / kern:,.0f} img/s, {nbytes / kern / 1e9:.0f} GB/s algorithmic; "
          f"with host packing + H2D {whole * 1e3:.2f} ms = {B / whole:,.0f} img/s; Pillow on one host thread {host * 1e3:.2f} ms/img = {1 / host:,.0f} img/s")
