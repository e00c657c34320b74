"""Does replaying the fused step as ONE captured graph shorten it on the GPU side?  (probe: the replayed step repeats the same
CutMix draw and LR values — timing only).  Two forms: (a) the default step — forward + head + trailing backward (lane fork /
join, per-stage AdamW on the lane): a multi-stream capture; (b) the single-stream form (`--overlap 0`, monolithic backward and
one AdamW launch): a linear chain of kernel nodes, the case the MI355X guide measures as eager == graph at kernel boundaries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from swat_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn(32, 3, 224, 224, device=dev, generator=g)
labels = torch.randint(0, 200, (32,), device=dev, generator=g)
perm = torch.randperm(32, device=dev); yb = labels[perm]
box, lam = (30, 40, 130, 160), 0.7


def probe(tag, trail, overlap):
    args, model, head, step, sched = bench.build_job(dev, 1, 32, trail=trail)
    L.swat_vit_set_overlap(overlap)
    np.random.seed(1); torch.manual_seed(1)
    bench.run_steps(args, step, sched, images, labels, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); bench.run_steps(args, step, sched, images, labels, 50); torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50 * 1e3
    t0 = time.perf_counter(); bench.run_steps(args, step, sched, images, labels, 50)
    host = (time.perf_counter() - t0) / 50 * 1e3   # enqueue only (no synchronisation inside)
    torch.cuda.synchronize()
    opt = step.opt
    opt.begin_step()
    b0, e0 = opt.begin_step, opt.end_step
    opt.begin_step = lambda: None
    opt.end_step = lambda: None
    msg = ""
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
        msg = f"graph replay {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step (loss {step.loss.item():.4f})"
    except Exception as e:
        msg = "capture failed: " + repr(e)[:300]
    finally:
        opt.begin_step, opt.end_step = b0, e0
    print(f"{tag}: eager {eager:.3f} ms/step, host enqueue {host:.3f} ms/step, {msg}", flush=True)
    L.swat_vit_set_overlap(1)
    del step, model, head
    torch.cuda.empty_cache()


probe("single stream (overlap 0, monolithic backward + one AdamW)", False, 0)
probe("default (lane + trailing per-stage AdamW)", True, 1)
