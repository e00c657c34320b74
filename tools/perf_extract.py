"""Dev tool: mined-feature extraction forward (path B1: encode_image + encode_text + l2 normalise per batch of --bsz pairs,
ref retrieval/extract_mined_feature.py:103-213, bsz 1024) on synthetic pixels / token ids: pairs per second and the share of
the bf16 MFMA peak the two towers' algorithmic flops reach."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd.engine import get_engine
from swat_amd.features import l2_normalize

cfg = sys.argv[1] if len(sys.argv) > 1 else "vitb32_openclip_laion400m"
bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
model, preprocess, tokenize = get_engine(cfg, device=dev)
model.eval()
g = torch.Generator().manual_seed(3)
imgs = torch.randn(bsz, 3, 224, 224, generator=g).to(dev)
tok = torch.randint(1, 49000, (bsz, 77), generator=g)
tok[:, 0] = 49406
eot = torch.randint(8, 77, (bsz,), generator=g)
for i in range(bsz):
    tok[i, eot[i]] = 49407
    tok[i, eot[i] + 1:] = 0
tok = tok.to(dev)


def timed(fn, n=5):
    with torch.no_grad():
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


v = model.visual
T, d, L, mlp = v.tokens, v.width, v.layers, v.mlp_width
gf_img = (2 * 3 * v.patch_size ** 2 * d * (T - 1) + L * (2 * T * (4 * d * d + 2 * d * mlp) + 4 * T * T * d)) / 1e9
tw = model.positional_embedding.shape[1]
gf_txt = (12 * (2 * 77 * 12 * tw * tw + 4 * 77 * 77 * tw)) / 1e9
ti = timed(lambda: l2_normalize(model.visual.forward_raw(imgs)))
tt = timed(lambda: l2_normalize(model.encode_text(tok)))
print(f"{cfg} bsz {bsz}: encode_image {ti * 1e3:.2f} ms = {bsz / ti:,.0f} img/s ({gf_img * bsz / ti / 1e3:.0f} TF/s, {gf_img * bsz / ti / 2.5e6:.3f} of bf16 MFMA peak); "
      f"encode_text {tt * 1e3:.2f} ms = {bsz / tt:,.0f} captions/s ({gf_txt * bsz / tt / 1e3:.0f} TF/s, {gf_txt * bsz / tt / 2.5e6:.3f}); "
      f"pairs/s {bsz / (ti + tt):,.0f}")
