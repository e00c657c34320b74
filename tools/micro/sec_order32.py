import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
def ft(tag):
    args, model, head, step, sched = bench.build_job(dev, 1, 32, arch="ViT-B-32-quickgelu")
    g = torch.Generator(device=dev).manual_seed(4321)
    images = torch.randn(32, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 200, (32,), device=dev, generator=g)
    bench.run_steps(args, step, sched, images, labels, 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.run_steps(args, step, sched, images, labels, 40)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    a = step.arena
    ws = model.visual._ws[(32, True)]
    ptrs = {"flat": a.flat.data_ptr(), "shadow": a.shadow.data_ptr(), "grad": a.grad.data_ptr(), "m": a.exp_avg.data_ptr(), "v": a.exp_avg_sq.data_ptr(),
            "ws": ws.data_ptr(), "img": images.data_ptr()}
    print(f"{tag:28s} {dt*1e3:7.3f} ms  " + " ".join(f"{k}={v:#x}" for k, v in ptrs.items()), flush=True)
    del step, model, head, args, sched
ft("fresh")
ft("again")
torch.cuda.empty_cache(); ft("after empty_cache")
x = torch.empty(6 * 2**30, dtype=torch.uint8, device=dev); del x; ft("after 6 GB alloc+free")
torch.cuda.empty_cache(); ft("after empty_cache")
xs = [torch.empty(s * 2**20, dtype=torch.uint8, device=dev) for s in (3, 17, 129, 1, 700)]; ft("odd live blocks")
del xs; torch.cuda.empty_cache(); ft("after empty_cache")
