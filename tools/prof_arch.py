"""Steps of the fused finetuning step on another architecture (run under rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
arch = sys.argv[1] if len(sys.argv) > 1 else "ViT-L-14"
dev = torch.device("cuda", 0)
args, model, head, step, sched = bench.build_job(dev, 1, 32, arch=arch)
g = torch.Generator(device=dev).manual_seed(1)
images = torch.randn(32, 3, 224, 224, device=dev, generator=g)
labels = torch.randint(0, 200, (32,), device=dev, generator=g)
bench.run_steps(args, step, sched, images, labels, 6)
torch.cuda.synchronize()
