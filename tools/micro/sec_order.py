import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
def ft(a, g):
    r = bench.finetune_secondary(dev, a, g); print(a, r["value"], r["ms_per_step"], "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 1), flush=True)
ft("ViT-B-16", 105.0)
r = bench.retrieval_secondary(dev); print("scan", r["value"], flush=True)
ft("ViT-B-16", 105.0)
torch.cuda.empty_cache()
ft("ViT-B-16", 105.0)
x = torch.empty(6 * 2**30, dtype=torch.uint8, device=dev); del x
ft("ViT-B-16", 105.0)
torch.cuda.empty_cache()
ft("ViT-B-16", 105.0)
