"""Guarded cross-check of the oracle against open_clip itself.  open_clip (open-clip-torch 2.26.1, requirements.txt:117) is an
un-vendored dependency that is absent from this image and from /root/reference, so the towers are pinned to HF CLIP instead
(tests/golden/vit_hf.npz, text_hf.npz) and DESIGN.md section 2 says "parity unpinned w.r.t. open_clip".  On the first box that HAS open_clip
this test compares oracle/vit_oracle.py with `open_clip.create_model('ViT-B-32-quickgelu')` — the only binding the reference has
(utils/extras.py:155) — at seeded random weights: same state-dict keys, same image and text embeddings."""
import pytest
import torch


def test_oracle_clip_equals_open_clip_at_seeded_weights():
    open_clip = pytest.importorskip("open_clip")
    from oracle import vit_oracle as vo
    torch.manual_seed(0)
    ref = open_clip.create_model("ViT-B-32-quickgelu", pretrained=None, precision="fp32").eval()
    sd = ref.state_dict()
    mine = vo.OracleCLIP("ViT-B-32-quickgelu").eval()
    msd = mine.state_dict()
    missing = [k for k in msd if k not in sd]
    assert not missing, f"oracle keys open_clip does not have: {missing[:5]}"
    mine.load_state_dict({k: sd[k] for k in msd})
    x = torch.randn(2, 3, 224, 224)
    tok = torch.zeros(2, 77, dtype=torch.long)
    tok[:, 0] = 49406
    tok[0, 1:5] = torch.tensor([320, 1125, 539, 320])
    tok[0, 5] = 49407
    tok[1, 1:3] = torch.tensor([1929, 2368])
    tok[1, 3] = 49407
    with torch.no_grad():
        a, b = ref.encode_image(x), mine.encode_image(x)
        assert (a - b).abs().max().item() <= 1e-4 * a.abs().max().item() + 1e-5
        c, d = ref.encode_text(tok), mine.encode_text(tok)
        assert (c - d).abs().max().item() <= 1e-4 * c.abs().max().item() + 1e-5
