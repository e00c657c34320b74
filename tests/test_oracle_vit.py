"""CPU: the fp32 ViT / text-tower oracle against an INDEPENDENT implementation of the same architecture (HF transformers
CLIPVisionModelWithProjection / CLIPTextModelWithProjection) and against what tests/golden/{vit,text}_hf.npz recorded
from the HF side: embeddings, per-block residual-stream digests and the gradients of named parameters."""
import os
import sys
import subprocess

import numpy as np
import pytest

CODE = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle import make_golden as mg
g = np.load(%r)
recs = getattr(mg, %r)(%r)          # asserts HF == oracle on embeddings, every block's output, the named gradients
for k, v in recs.items():
    ref = g[k]
    if k.endswith("_tokens"):
        assert (v == ref).all(), k
        continue
    tol = 2e-5 if k.endswith("_embeds") else 2e-4
    scale = max(1.0, float(np.abs(ref).max()))
    assert v.shape == ref.shape and np.abs(v - ref).max() <= tol * scale, (k, float(np.abs(v - ref).max()))
print("ok", len(recs))
"""


@pytest.mark.parametrize("fn,case,fixture", [("vit_case", "tiny_gelu", "vit_hf.npz"), ("vit_case", "b32_quickgelu", "vit_hf.npz"),
                                             ("text_case", "tiny_text", "text_hf.npz"), ("text_case", "b32_text", "text_hf.npz")])
def test_oracle_towers_match_hf_clip_and_the_golden_vectors(golden_dir, fn, case, fixture):
    # own process: transformers' import probes torchvision, which other tests may have stubbed
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", CODE % (root, os.path.join(golden_dir, fixture), fn, case)], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_oracle_lr_closed_form_is_monotone_after_warmup():
    from oracle import vit_oracle as vo
    lrs = [vo.lr_at(i, 1e-4, 1000) for i in range(400)]
    assert lrs[0] == 1e-6 and abs(lrs[50] - 1e-4) < 1e-18
    assert all(a < b for a, b in zip(lrs[1:50], lrs[2:51]))
    assert all(a > b for a, b in zip(lrs[50:399], lrs[51:400]))
