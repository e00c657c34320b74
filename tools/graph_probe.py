"""Does replaying the fused step as ONE captured graph shorten it on the GPU side?  (probe: the replayed step repeats the same
CutMix draw and LR values — timing only).  Captures forward + head + trailing backward (lane fork / join, per-stage AdamW)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from swat_amd.training import cutmix_draw
dev = torch.device("cuda", 0)
args, model, head, step, sched = bench.build_job(dev, 1, 32)
g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn(32, 3, 224, 224, device=dev, generator=g)
labels = torch.randint(0, 200, (32,), device=dev, generator=g)
np.random.seed(1); torch.manual_seed(1)
bench.run_steps(args, step, sched, images, labels, 10)
torch.cuda.synchronize()
t0 = time.perf_counter(); bench.run_steps(args, step, sched, images, labels, 50); torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step")
perm = torch.randperm(32, device=dev); yb = labels[perm]
box, lam = (30, 40, 130, 160), 0.7
opt = step.opt
opt.begin_step()
b0, e0 = opt.begin_step, opt.end_step
opt.begin_step = lambda: None
opt.end_step = lambda: None
try:
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(images, labels, yb, lam, perm=perm, box=box)
    torch.cuda.synchronize()
    for _ in range(5): graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): graph.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step, loss {step.loss.item():.4f}")
except Exception as e:
    print("capture failed:", repr(e)[:500])
finally:
    opt.begin_step, opt.end_step = b0, e0
