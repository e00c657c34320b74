"""Fused ViT-B/32 step at several per-GPU batch sizes: img/s and a finite loss (robustness + where the step scales)."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
dev = torch.device("cuda", 0)
for B in (1, 7, 16, 32, 64, 128, 256):
    args, model, head, step, sched = bench.build_job(dev, 1, B)
    g = torch.Generator(device=dev).manual_seed(B)
    images = torch.randn(B, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 200, (B,), device=dev, generator=g)
    np.random.seed(1); torch.manual_seed(1)
    bench.run_steps(args, step, sched, images, labels, 5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.run_steps(args, step, sched, images, labels, 20)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    loss = step.loss.item()
    assert math.isfinite(loss), (B, loss)
    print(f"B={B:4d}: {B / dt:9.1f} img/s  {dt * 1e3:8.3f} ms/step  loss {loss:.4f}", flush=True)
    del step, model, head
