"""Is the fused step host-bound?  Host enqueue time per step (no sync inside the loop) vs wall time per step, and the
host cost of the pieces around the C calls (tuning aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from swat_amd import _lib
from swat_amd.training import cutmix_draw
dev = torch.device("cuda", 0)
ov = int(os.environ.get("OVERLAP", 2))
args, model, head, step, sched = bench.build_job(dev, 1, 32, trail=ov == 2)
_lib.lib().swat_vit_set_overlap(min(ov, 1))
g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn(32, 3, 224, 224, device=dev, generator=g)
labels = torch.randint(0, 200, (32,), device=dev, generator=g)
np.random.seed(1); torch.manual_seed(1)
bench.run_steps(args, step, sched, images, labels, 10)
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
bench.run_steps(args, step, sched, images, labels, K)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"overlap {ov}: host enqueue {1e3*(t1-t0)/K:.3f} ms/step, wall {1e3*(t2-t0)/K:.3f} ms/step, drain after last enqueue {1e3*(t2-t1):.3f} ms")

def t(name, fn, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print(f"{name:40s} {dt:8.1f} us", flush=True)
t("cutmix_draw", lambda: cutmix_draw(args, images.size()))
perm = torch.randperm(32)
t("perm.to(dev)", lambda: perm.to(dev, non_blocking=True))
pd = perm.to(dev)
t("labels[perm]", lambda: labels[pd])
t("sched.step", lambda: sched.step())
t("step() no cutmix", lambda: step(images, labels, labels, 1.0), 50)
t("step() cutmix", lambda: step(images, labels, labels[pd], 0.6, perm=pd, box=(30, 40, 130, 160)), 50)
