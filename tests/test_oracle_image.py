"""CPU: oracle/image_oracle.py (the restated 8-bit bicubic resampler + ToTensor/Normalize) against Pillow itself, the
library torchvision's transforms call in the reference (utils/extras.py:66-85), and against the host transforms of
swat_amd/transforms.py that a DataLoader worker runs."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import image_oracle as io
from swat_amd import transforms as T

CASES = [(375, 500, 224, 224), (500, 375, 224, 224), (100, 80, 224, 224), (224, 300, 224, 224), (300, 224, 224, 224),
         (224, 224, 224, 224), (1, 50, 224, 224), (50, 1, 224, 224), (1500, 2000, 224, 224), (333, 500, 224, 336),
         (64, 64, 32, 32), (7, 9, 32, 32), (2, 2, 224, 224), (225, 223, 224, 224)]


def _image(rng, h, w, smooth=False):
    if not smooth:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([127 + 120 * np.sin(x / 17.0 + c) * np.cos(y / 23.0 - c) for c in range(3)], -1)
    return np.clip(a + rng.normal(0, 4, a.shape), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w,ow,oh", CASES)
def test_resize_matches_pillow(h, w, ow, oh):
    rng = np.random.default_rng(h * 7919 + w)
    for smooth in (False, True):
        a = _image(rng, h, w, smooth)
        ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(io.resize(a, ow, oh), ref)


def test_extreme_values_clamp_like_pillow():
    a = np.zeros((90, 120, 3), np.uint8)
    a[::2, ::3] = 255  # ringing of the negative lobes drives sums below 0 and above 255
    ref = np.asarray(Image.fromarray(a).resize((224, 224), Image.BICUBIC))
    assert np.array_equal(io.resize(a, 224, 224), ref)
    ref = np.asarray(Image.fromarray(a).resize((40, 30), Image.BICUBIC))
    assert np.array_equal(io.resize(a, 40, 30), ref)


@pytest.mark.parametrize("mode", ["train", "val"])
def test_whole_transform_matches_host_path(mode):
    rng = np.random.default_rng(3)
    tf = T.build_transform(224, mode)
    dt = T.DeviceBatchTransform.__new__(T.DeviceBatchTransform)  # geometry() only: no GPU, no library
    dt.n_px, dt.mode = 224, mode
    for h, w in [(375, 500), (500, 333), (224, 224), (180, 260), (640, 480)]:
        a = _image(rng, h, w, smooth=True)
        torch.manual_seed(h + w)
        ref = tf(Image.fromarray(a)).numpy()
        torch.manual_seed(h + w)
        x0, y0, in_w, in_h, out_w, out_h, off_x, off_y, flip = dt.geometry(w, h)
        if mode == "train":
            got = io.apply(a, 224, box=(y0, x0, in_h, in_w), flip=bool(flip), mode="train")
        else:
            assert (out_w, out_h, off_x, off_y) == io.val_geometry(w, h, 224)
            got = io.apply(a, 224, mode="val")
        assert got.dtype == np.float32 and np.array_equal(got, ref)


def _write_split(tmp_path, rng, n=10):
    import os
    os.makedirs(tmp_path / "img", exist_ok=True)
    lines = []
    for i in range(n):
        h, w = int(rng.integers(120, 400)), int(rng.integers(120, 400))
        Image.fromarray(_image(rng, h, w, smooth=True)).save(tmp_path / "img" / f"{i}.png")
        lines.append(f"img/{i}.png {i % 3} {i % 2}\n")
    (tmp_path / "split.txt").write_text("".join(lines))
    return [["split.txt"], [str(tmp_path)]]


class _HostStandIn:
    """what DeviceBatchTransform computes, through the oracle (no GPU in this test)"""

    def __init__(self, n_px, mode, device):
        self.n_px, self.mode = n_px, mode

    def __call__(self, imgs, geometry):
        out = []
        for a, g in zip(imgs, geometry):
            x0, y0, in_w, in_h, out_w, out_h, off_x, off_y, flip = g
            if self.mode == "train":
                out.append(io.apply(a, self.n_px, box=(y0, x0, in_h, in_w), flip=bool(flip), mode="train"))
            else:
                out.append(io.apply(a, self.n_px, mode="val"))
        return torch.from_numpy(np.stack(out))


@pytest.mark.parametrize("mode", ["train", "val"])
def test_decode_only_loader_yields_the_host_loaders_batches(tmp_path, monkeypatch, mode):
    from torch.utils.data import DataLoader
    from swat_amd import dataloader as dl
    monkeypatch.setattr(dl, "DeviceBatchTransform", _HostStandIn)
    split = _write_split(tmp_path, np.random.default_rng(5))
    toks = {str(c): {"all": torch.arange(12).view(4, 1, 3) + 100 * c} for c in range(3)}
    host = dl.MyDataset(str(tmp_path), split, T.build_transform(224, mode), toks)
    raw = dl.MyDataset(str(tmp_path), split, T.DecodeOnly(224, mode), toks)
    torch.manual_seed(9)
    want = list(DataLoader(host, batch_size=4, shuffle=(mode == "train"), num_workers=0))
    torch.manual_seed(9)
    got = list(dl.DeviceTransformLoader(raw, 224, mode, "cpu", batch_size=4, shuffle=(mode == "train"), num_workers=0))
    assert len(got) == len(want) == 3
    for g, w in zip(got, want):
        assert len(g) == 4 and all(torch.equal(a, b) for a, b in zip(g, w))


def test_weight_range_over_a_size_sweep():
    """normalised fixed-point weights stay within 23 bits + sign for every region / output size tried (the largest
    occur where an edge cuts the kernel of an up-scaled axis: about 1.125 x 2^22)"""
    worst = 0
    for out in (224, 336, 32, 30):
        for ins in list(range(1, 48)) + [63, 100, 223, 224, 225, 300, 448, 500, 1000, 2000, 4000]:
            b, k = io.axis_table(ins, out)
            assert (b[:, 1] <= io.ksize_of(ins, out)).all() and (b[:, 1] >= 1).all()
            worst = max(worst, int(np.abs(k).max()))
    assert worst < (1 << 23)
