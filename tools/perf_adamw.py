"""Dev tool: swat_adamw alone over an 88 M-parameter arena (ViT-B/32 + head), wide and thin (256-block) launch shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
n = int(os.environ.get("N", 88_000_000)) // 4 * 4
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3
m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev); sh = torch.empty(n, dtype=torch.bfloat16, device=dev)
lrs = torch.tensor([1e-4, 1e-6, 1e-4], device=dev); bc = torch.tensor([0.1, 0.001], device=dev)
segs = [0, 102400, n - 4]
def run(k=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): ops.adamw_step(p, g, m, v, sh, segs, lrs, bc)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
for blocks in (0, 256):
    if hasattr(L, "swat_adamw_set_blocks"): L.swat_adamw_set_blocks(blocks)
    run(3)
    ts = sorted(run() for _ in range(5))
    print(f"blocks {blocks}: {ts[2] * 1e3:.1f} us per launch = {n * 30 / ts[2] / 1e9:.2f} TB/s of the 30 B per parameter")
