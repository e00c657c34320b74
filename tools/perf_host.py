"""Host cost of the individual calls in the trailing step (tuning aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from swat_amd import _lib
dev = torch.device("cuda", 0)
args, model, head, step, sched = bench.build_job(dev, 1, 32, trail=True)
lib = _lib.lib()
g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn(32, 3, 224, 224, device=dev, generator=g)
labels = torch.randint(0, 200, (32,), device=dev, generator=g)
bench.run_steps(args, step, sched, images, labels, 3)
torch.cuda.synchronize()
sl = step._slice_args()
side = step._opt_stream.cuda_stream
main = torch.cuda.current_stream().cuda_stream
ws, dfeats = step._head_ws(32, dev)

def t(name, fn, n=50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print(f"{name:40s} {dt:8.1f} us", flush=True)

t("adamw slice on main", lambda: lib.swat_adamw(*sl["stages"][5], main))
t("adamw slice on side", lambda: lib.swat_adamw(*sl["stages"][5], side))
t("lane_order(side)", lambda: lib.swat_vit_lane_order(side))
for mode in (0, 1, 2):
    lib.swat_vit_set_overlap(mode)
    t(f"backward stage 5, mode {mode}", lambda: step.visual.backward_raw(dfeats, stage_hi=5, stage_lo=5))
    lib.swat_vit_backward_join(main)
lib.swat_vit_set_overlap(2)
def both():
    step.visual.backward_raw(dfeats, stage_hi=5, stage_lo=5)
    lib.swat_vit_lane_order(side)
    lib.swat_adamw(*sl["stages"][5], side)
t("stage 5 + order + adamw (mode 2)", both)
lib.swat_vit_backward_join(main)
lib.swat_vit_set_overlap(1)
t("forward", lambda: step.visual.forward_raw(images, save=True), 20)
t("opt.begin_step", lambda: step.opt.begin_step(), 20)
s2 = step._opt_stream
t("side.wait_stream(main)", lambda: s2.wait_stream(torch.cuda.current_stream()), 20)
