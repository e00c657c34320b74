"""Step time of the fused finetuning step on another architecture and overlap mode: python tools/perf_arch.py ViT-L-14 [mode]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from swat_amd import _lib
arch = sys.argv[1] if len(sys.argv) > 1 else "ViT-L-14"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
args, model, head, step, sched = bench.build_job(dev, 1, 32, arch=arch, trail=mode == 2)
_lib.lib().swat_vit_set_overlap(min(mode, 1))
g = torch.Generator(device=dev).manual_seed(4321)
images = torch.randn(32, 3, 224, 224, device=dev, generator=g)
labels = torch.randint(0, 200, (32,), device=dev, generator=g)
bench.run_steps(args, step, sched, images, labels, 4)
torch.cuda.synchronize()
t0 = time.perf_counter()
bench.run_steps(args, step, sched, images, labels, 10)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(arch, "mode", mode, round(32 / dt, 1), "img/s", round(dt * 1e3, 3), "ms/step")
