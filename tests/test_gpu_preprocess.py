"""GPU: swat_preprocess_batch (through the C ABI, via swat_amd.transforms.DeviceBatchTransform) against Pillow — the
library behind the reference's torchvision transforms (utils/extras.py:66-85) — and against oracle/image_oracle.py.
Byte / index work: the bar is bit-exact."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import image_oracle as io
from swat_amd import transforms as T

pytestmark = pytest.mark.gpu


def _image(rng, h, w, smooth=True):
    if not smooth:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([127 + 120 * np.sin(x / 17.0 + c) * np.cos(y / 23.0 - c) for c in range(3)], -1)
    return np.clip(a + rng.normal(0, 6, a.shape), 0, 255).astype(np.uint8)


# one group per kernel instantiation: the tap count of a batch (its largest reduction) selects the unrolled form
GROUPS = {"taps5": [(100, 80), (224, 224), (50, 40), (180, 260)], "taps9": [(300, 400), (97, 131), (224, 301), (301, 224)],
          "taps13": [(375, 500), (500, 333), (333, 500), (640, 480)], "taps17": [(800, 600), (375, 500)],
          "taps25": [(768, 1024), (1100, 900), (224, 224)], "generic": [(1200, 1600), (375, 500), (2000, 1500)]}


@pytest.mark.parametrize("group", list(GROUPS))
@pytest.mark.parametrize("mode", ["train", "val"])
def test_batch_matches_host_transform(mode, group):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    rng = np.random.default_rng(11)
    imgs = [_image(rng, h, w, smooth=(i % 2 == 0)) for i, (h, w) in enumerate(GROUPS[group])]
    tf = T.build_transform(224, mode)
    torch.manual_seed(5)
    ref = torch.stack([tf(Image.fromarray(a)) for a in imgs])
    torch.manual_seed(5)
    got = T.DeviceBatchTransform(224, mode)([Image.fromarray(a) for a in imgs])
    assert got.shape == (len(imgs), 3, 224, 224) and got.dtype == torch.float32
    assert torch.equal(got.cpu(), ref)


def test_full_batch_of_mixed_sizes():
    rng = np.random.default_rng(14)
    sizes = [(int(rng.integers(200, 700)), int(rng.integers(200, 700))) for _ in range(32)]
    imgs = [_image(rng, h, w, smooth=False) for h, w in sizes]
    tf = T.build_transform(224, "train")
    torch.manual_seed(6)
    ref = torch.stack([tf(Image.fromarray(a)) for a in imgs])
    torch.manual_seed(6)
    assert torch.equal(T.DeviceBatchTransform(224, "train")(imgs).cpu(), ref)


def test_edge_geometries_match_oracle():
    """one-pixel regions, up-scaling, identity axes, odd byte offsets, forced flips, a 4000-pixel-wide panorama"""
    rng = np.random.default_rng(12)
    small = [((1, 1), (0, 0, 1, 1)), ((3, 5), (1, 2, 1, 3)), ((224, 224), (0, 0, 224, 224)), ((224, 500), (0, 100, 224, 333)),
             ((500, 224), (13, 0, 401, 224)), ((31, 17), (0, 0, 31, 17)), ((227, 229), (1, 1, 225, 227))]
    _check_cases(rng, small)  # unrolled kernels
    _check_cases(rng, small + [((600, 4000), (7, 3, 590, 3990)), ((2000, 90), (100, 1, 1800, 88))])  # run-time tap loops


def _check_cases(rng, cases):
    imgs, geo = [], []
    for i, ((h, w), (y0, x0, ch, cw)) in enumerate(cases):
        imgs.append(_image(rng, h, w, smooth=False))
        geo.append((x0, y0, cw, ch, 224, 224, 0, 0, i % 2))
    got = T.DeviceBatchTransform(224, "train")(imgs, geometry=geo).cpu().numpy()
    for i, (a, g) in enumerate(zip(imgs, geo)):
        x0, y0, cw, ch = g[:4]
        want = io.apply(a, 224, box=(y0, x0, ch, cw), flip=bool(g[8]), mode="train")
        assert np.array_equal(got[i], want), f"case {i}: {cases[i]}"
        pil = np.asarray(Image.fromarray(a).crop((x0, y0, x0 + cw, y0 + ch)).resize((224, 224), Image.BICUBIC))
        if g[8]:
            pil = pil[:, ::-1]
        assert np.array_equal(got[i], np.stack([io.normalize_lut()[c][pil[:, :, c]] for c in range(3)]))


def test_other_resolution_and_single_image():
    rng = np.random.default_rng(13)
    a = _image(rng, 300, 420)
    tf = T.build_transform(336, "val")
    got = T.DeviceBatchTransform(336, "val")([a])
    assert torch.equal(got.cpu()[0], tf(Image.fromarray(a)))
    assert T.DeviceBatchTransform(224, "val")([]).shape == (0, 3, 224, 224)
    for n in (30, 32):  # 30: not a multiple of 4 -> the run-time tap loops
        assert torch.equal(T.DeviceBatchTransform(n, "val")([a]).cpu()[0], T.build_transform(n, "val")(Image.fromarray(a)))
        torch.manual_seed(1)
        ref = T.build_transform(n, "train")(Image.fromarray(a[:40, :50]))
        torch.manual_seed(1)
        assert torch.equal(T.DeviceBatchTransform(n, "train")([a[:40, :50]]).cpu()[0], ref)


def test_bad_geometry_is_refused():
    a = np.zeros((10, 10, 3), np.uint8)
    with pytest.raises(ValueError):
        T.DeviceBatchTransform(224, "train")([a], geometry=[(5, 5, 10, 10, 224, 224, 0, 0, 0)])
    with pytest.raises(ValueError):
        T.DeviceBatchTransform(224, "train")([np.zeros((10, 10), np.uint8)])


def test_loader_with_gpu_transform_matches_host_loader(tmp_path):
    """real files, worker processes that only decode, the batch transformed by the HIP kernels: the same batches as
    the reference-shaped loader (utils/dataloader.py:107-135) whose workers run Pillow"""
    import os
    from types import SimpleNamespace
    from swat_amd import dataloader as dl
    rng = np.random.default_rng(21)
    os.makedirs(tmp_path / "img")
    lines = []
    for i in range(12):
        h, w = int(rng.integers(120, 500)), int(rng.integers(120, 500))
        Image.fromarray(_image(rng, h, w)).save(tmp_path / "img" / f"{i}.png")
        lines.append(f"img/{i}.png {i % 3} {i % 2}\n")
    (tmp_path / "split.txt").write_text("".join(lines))
    split = [["split.txt"], [str(tmp_path)]]
    args = SimpleNamespace(dataset_root=str(tmp_path), bsz=4, num_workers=0, device="cuda:0", host_transform=True)
    val_tf = T.build_transform(224, "val")
    torch.manual_seed(3)
    want = [list(l) for l in dl.get_dataloader(args, split, split, split, None, val_tf)]
    args.host_transform, args.host_decode = False, True
    torch.manual_seed(3)
    loaders = dl.get_dataloader(args, split, split, split, None, val_tf)
    assert all(isinstance(l, dl.DeviceTransformLoader) for l in loaders)
    got = [list(l) for l in loaders]
    for lw, lg in zip(want, got):
        assert len(lw) == len(lg) > 0
        for bw, bg in zip(lw, lg):
            assert bg[0].is_cuda and torch.equal(bg[0].cpu(), bw[0])
            assert all(torch.equal(a, b) for a, b in zip(bg[1:], bw[1:]))
    # worker processes: same batches for the un-shuffled validation loader
    args.num_workers = 2
    assert all(torch.equal(a[0].cpu(), b[0]) for a, b in zip(dl.get_dataloader(args, split, split, split, None, val_tf)[1], want[1]))
