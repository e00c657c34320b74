"""HIP stream -> hardware-queue placement and the fused step (DESIGN §4 'Streams and hardware queues').

Each case runs in a fresh process: K torch streams are created BEFORE the library creates its own (main = torch's
default stream, lane, aux), optionally with GPU_MAX_HW_QUEUES set, and the trailing ViT-B/32 step is timed.
    python tools/micro/stream_queue.py            # the sweep
    python tools/micro/stream_queue.py K          # one case (internal)
"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def one(k):
    import torch, bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    keep = [torch.cuda.Stream() for _ in range(k)]          # user streams created first
    for s in keep:
        with torch.cuda.stream(s):
            torch.zeros(1, device=dev)
    args, model, head, step, sched = bench.build_job(dev, 1, 32)
    g = torch.Generator(device=dev).manual_seed(1)
    images = torch.randn(32, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 200, (32,), device=dev, generator=g)
    bench.run_steps(args, step, sched, images, labels, 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.run_steps(args, step, sched, images, labels, 40)
    torch.cuda.synchronize()
    print(f"pre-created torch streams {k}  GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: "
          f"{(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step", flush=True)

if len(sys.argv) > 1:
    one(int(sys.argv[1]))
else:
    for q in (None, "2", "8"):
        for k in (0, 1, 2, 3, 4):
            env = dict(os.environ)
            if q:
                env["GPU_MAX_HW_QUEUES"] = q
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(k)], env=env, capture_output=True, text=True, timeout=300)
            print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
