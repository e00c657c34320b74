"""Dev tool: the benchmark's step as a data-parallel rank runs it (gradient buckets trailing the weight gradients, all-reduce + AdamW on
their own streams, per-step flag exchange) inside a 1-RANK RCCL group — every collective is a local copy — against the one-GPU step,
same process, alternating.  What it shows: the cost of the N > 1 step's structure itself (what the links add comes on top).
  python3 tools/perf_dp_step.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import bench
from swat_amd import _lib
from swat_amd.step import FusedTrainStep

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L = _lib.lib()

def job(dp, persistent=True):
    os.environ["SWAT_DP_PERSISTENT"] = "1" if persistent else "0"
    args, model, head, step, sched = bench.build_job(dev, 1, 32)
    if dp:
        step = FusedTrainStep(model, head, step.logit_scale, step.opt, world_size=1, staged=True)
        step.reduce = True
    g = torch.Generator(device=dev).manual_seed(1234)
    images = torch.randn(32, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 200, (32,), device=dev, generator=g)
    np.random.seed(1); torch.manual_seed(1)
    bench.run_steps(args, step, sched, images, labels, 20)
    torch.cuda.synchronize()
    b0 = L.swat_vit_mega_backward_launches()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        bench.run_steps(args, step, sched, images, labels, N)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / N * 1e3)
    return min(ts), (L.swat_vit_mega_backward_launches() - b0) / (3 * N)

for name, dp, pers in (("one GPU", False, True), ("data-parallel rank, persistent backward + trailing buckets", True, True),
                       ("data-parallel rank, stage-by-stage backward (rounds 2-4)", True, False), ("one GPU", False, True)):
    ms, pb = job(dp, pers)
    print(f"{name:62s} {ms:.3f} ms/step = {32 / ms * 1e3:,.0f} img/s   persistent backward launches per step {pb:.2f}")
assert L.swat_vit_mega_status() == 0
dist.destroy_process_group()
