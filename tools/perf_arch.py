"""Step time of the fused finetuning step on another architecture: python tools/perf_arch.py ViT-L-14"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
arch = sys.argv[1] if len(sys.argv) > 1 else "ViT-L-14"
r = bench.finetune_secondary(torch.device("cuda", 0), arch, 1.0, steps=10, warmup=4)
print(arch, r["value"], "img/s", r["ms_per_step"], "ms/step")
