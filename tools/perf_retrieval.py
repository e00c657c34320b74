"""Quick on-GPU timing of the path B kernels (dev tool; bench.py is the judged entry)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd.retrieval import ops

dev = torch.device("cuda:0")
n, d, nseg, k = int(os.environ.get("N", 1_250_000)), 512, 125, 500
feats = torch.randn(n, d, device=dev); feats /= feats.norm(dim=-1, keepdim=True)
q = torch.randn(nseg, d, device=dev); q /= q.norm(dim=-1, keepdim=True)
per = n // nseg
off = torch.arange(0, per * nseg + 1, per, device=dev, dtype=torch.int64)
row_seg = torch.repeat_interleave(torch.arange(nseg, device=dev, dtype=torch.int32), per)
n = per * nseg
feats = feats[:n].contiguous()
sims = torch.empty(n, device=dev)
for name, fn in [("sim", lambda: ops.similarity_segmented(feats, q, row_seg, out=sims)),
                 ("topk", lambda: ops.segment_topk(sims, off, k, thr=0.0)),
                 ("copy", lambda: feats.clone())]:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = n * d * 4 / 1e9
    print(f"{name}: {ms:.3f} ms  ({gb/ms*1e3:.0f} GB/s on {gb:.2f} GB matrix)")
