"""Host vs device time of the fused step in the three overlap modes (tuning aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from swat_amd import _lib

dev = torch.device("cuda", 0)
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for mode in (2, 2):
    args, model, head, step, sched = bench.build_job(dev, 1, BS, trail=mode == 2)
    _lib.lib().swat_vit_set_overlap(min(mode, 1))
    _lib.lib().swat_gemm_set_exp(int(os.environ.get('SWAT_GEMM_EXP', '0')))
    g = torch.Generator(device=dev).manual_seed(1234)
    images = torch.randn(BS, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 200, (BS,), device=dev, generator=g)
    np.random.seed(1); torch.manual_seed(1)
    bench.run_steps(args, step, sched, images, labels, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_steps(args, step, sched, images, labels, 100)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs = []
    for _ in range(5):
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        bench.run_steps(args, step, sched, images, labels, 4)
        hs.append((time.perf_counter() - t3) / 4 * 1e3)
    torch.cuda.synchronize()
    print(f"mode {mode}: pure host (4 steps from an empty queue) {min(hs):.3f} ms/step")
    print(f"mode {mode}: host {1e1*(t1-t0):.3f} ms/step  total {1e1*(t2-t0):.3f} ms/step", flush=True)
    del step, model, head
    torch.cuda.empty_cache()
