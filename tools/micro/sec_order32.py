"""The same ViT-B/32 job built repeatedly in one process: step time per 20-step window (allocation-state experiment)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
def ft(tag, windows=6, sleep=0.0):
    args, model, head, step, sched = bench.build_job(dev, 1, 32, arch="ViT-B-32-quickgelu")
    g = torch.Generator(device=dev).manual_seed(4321)
    images = torch.randn(32, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 200, (32,), device=dev, generator=g)
    torch.cuda.synchronize()
    if sleep:
        time.sleep(sleep)
    out = []
    for w in range(windows):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bench.run_steps(args, step, sched, images, labels, 20)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 20 * 1e3)
    # probes on this job's own buffers: streaming copy, strided row gather (one 128-B line per 6 KB), GEMM on the workspace
    a = step.arena
    def timeit(fn, n=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    t_copy = timeit(lambda: a.exp_avg.copy_(a.flat))
    v = a.flat[: (a.numel // 1536) * 1536].view(-1, 1536)
    t_gather = timeit(lambda: v[:, :32].sum())
    ws = model.visual._ws[(32, True)]
    wv = ws[: 4096 * 4096 * 2 * 2].view(torch.bfloat16).view(2, 4096, 4096)
    t_mm = timeit(lambda: torch.matmul(wv[0], wv[1]))
    roof = bench.gemm_roofline(args, step, sched, images, labels, 5, 32, 2)
    kinds = " ".join(f"{k}:{v['avg_us']:.1f}" for k, v in roof["per_instantiation"].items() if v["launches_per_step"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step.trail = False
    from swat_amd import _lib
    _lib.lib().swat_vit_set_overlap(0)
    bench.run_steps(args, step, sched, images, labels, 20)
    torch.cuda.synchronize(); one_stream = (time.perf_counter() - t0) / 20 * 1e3
    _lib.lib().swat_vit_set_overlap(1)
    print(f"{tag:34s} one-stream {one_stream:6.2f} ms, GEMM {roof['gemm_ms_per_step']:.2f} ms/step [{kinds}]")
    print(f"{tag:34s} " + " ".join(f"{x:6.2f}" for x in out) + f"  ms/step | copy {t_copy:7.1f} us  strided {t_gather:7.1f} us  matmul {t_mm:7.1f} us", flush=True)
    del step, model, head, args, sched
ft("fresh")
ft("again")
torch.cuda.empty_cache(); ft("after empty_cache")
torch.cuda.empty_cache(); ft("after empty_cache, sleep 3 s", sleep=3.0)
torch.cuda.empty_cache(); ft("after empty_cache")
ft("again")
