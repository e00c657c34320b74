"""CPU: the fp32 ViT oracle against an INDEPENDENT implementation of the same architecture (HF transformers CLIP)
and against the embeddings recorded in tests/golden/vit_hf.npz."""
import os
import sys
import subprocess

import numpy as np


def test_oracle_tower_matches_hf_clip(golden_dir):
    # own process: transformers' import probes torchvision, which other tests may have stubbed
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
from oracle import vit_oracle as vo
from oracle.make_golden import hf_to_oracle_state
g = np.load(%r)
img, patch, width, layers, heads, embed, act = 56, 14, 128, 2, 2, 64, "gelu"
torch.manual_seed(0)
hf = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers,
     num_attention_heads=heads, patch_size=patch, image_size=img, projection_dim=embed, hidden_act=act, layer_norm_eps=1e-5)).eval()
with torch.no_grad():
    for n, p in hf.named_parameters():
        if n.endswith("bias"):
            p.add_(0.02 * torch.randn_like(p))
ref = vo.OracleVisionTransformer(img, patch, width, layers, heads, embed, act)
ref.load_state_dict(hf_to_oracle_state(hf))
x = torch.randn(2, 3, img, img, generator=torch.Generator().manual_seed(1))
with torch.no_grad():
    a, b = hf(pixel_values=x).image_embeds, ref(x)
assert (a - b).abs().max().item() < 2e-5, (a - b).abs().max().item()
assert np.abs(b.numpy() - g["tiny_gelu_embeds"]).max() < 2e-5
print("ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(golden_dir, "vit_hf.npz"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_oracle_lr_closed_form_is_monotone_after_warmup():
    from oracle import vit_oracle as vo
    lrs = [vo.lr_at(i, 1e-4, 1000) for i in range(400)]
    assert lrs[0] == 1e-6 and abs(lrs[50] - 1e-4) < 1e-18
    assert all(a < b for a, b in zip(lrs[1:50], lrs[2:51]))
    assert all(a > b for a, b in zip(lrs[50:399], lrs[51:400]))
