"""GPU: baseline-JPEG decode (host Huffman + HIP dequantise / inverse DCT / upsample / colour: csrc/jpeg.hip) against Pillow — every
byte — and through the loaders (--gpu_decode) against the reference-shaped loader whose workers run Pillow.
ref: utils/datasets/dataset_utils.py:130-180, utils/dataloader.py:107-135."""
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image

from tests.test_oracle_jpeg import CASES, jpeg_bytes, synth

pytestmark = pytest.mark.gpu


def test_decode_batch_equals_pillow_byte_for_byte():
    from oracle import jpeg_oracle as jo
    from swat_amd import jpeg
    rng = np.random.default_rng(5)
    files, want = [], []
    cases = CASES + [(375, 500, 2, 90, 0), (500, 375, 1, 75, 5), (224, 224, 0, 100, 0), (257, 1000, 2, 60, 0), (9, 3, 2, 85, 0), (6, 4, 1, 85, 0)]
    # a chroma row of <= 2 samples (W <= 4 with horizontal subsampling): libjpeg replicates instead of interpolating — the parser hands
    # such files to Pillow, like the two files of other formats below
    narrow = sum(1 for h, w, sub, q, rst in cases if sub and (w + 1) // 2 <= 2)
    for h, w, sub, q, rst in cases:
        files.append(jpeg_bytes(synth(rng, h, w), q, sub, rst))
        want.append(np.asarray(Image.open(io.BytesIO(files[-1])).convert("RGB")))
    files.append(jpeg_bytes(synth(rng, 40, 30)[..., 0], 80))                      # grayscale
    want.append(np.asarray(Image.open(io.BytesIO(files[-1])).convert("RGB")))
    b = io.BytesIO()
    Image.fromarray(synth(rng, 33, 44)).save(b, "JPEG", progressive=True)         # not baseline: decoded by Pillow, same buffer
    files.append(b.getvalue())
    want.append(np.asarray(Image.open(io.BytesIO(files[-1])).convert("RGB")))
    b = io.BytesIO()
    Image.fromarray(synth(rng, 21, 19)).save(b, "PNG")                            # not a JPEG at all
    files.append(b.getvalue())
    want.append(np.asarray(Image.open(io.BytesIO(files[-1])).convert("RGB")))
    assert np.array_equal(jo.decode(files[3]), want[3])                            # (the oracle is the checker, spot-checked here too)
    n_dev0, n_host0 = jpeg.stats["device"], jpeg.stats["host_other_format"]
    for threads in (1, 4):
        pixels, offsets, shapes = jpeg.decode_batch(files, "cuda:0", threads=threads)
        host = pixels.cpu().numpy()
        for i, w_ in enumerate(want):
            h, w = shapes[i]
            assert (h, w) == w_.shape[:2]
            got = host[offsets[i]:offsets[i] + 3 * h * w].reshape(h, w, 3)
            assert np.array_equal(got, w_), (i, threads, np.abs(got.astype(int) - w_.astype(int)).max())
    # the two files of other formats went to Pillow — visibly: counted, and refused in strict mode; every baseline file ran on the device
    assert jpeg.stats["host_other_format"] - n_host0 == 2 * (2 + narrow) and jpeg.stats["device"] - n_dev0 == 2 * (len(files) - 2 - narrow)
    import os
    from swat_amd import _lib
    os.environ["SWAT_JPEG_STRICT"] = "1"
    try:
        with pytest.raises(_lib.SwatHipError):
            jpeg.decode_batch(files, "cuda:0")
        jpeg.decode_batch([f for f, (h, w, sub, q, rst) in zip(files, cases) if not (sub and (w + 1) // 2 <= 2)] + [files[len(cases)]], "cuda:0")
    finally:
        del os.environ["SWAT_JPEG_STRICT"]
    one = jpeg.decode(files[10], "cuda:0")
    assert np.array_equal(one.cpu().numpy(), want[10])


def test_decode_feeds_the_device_transform_like_pillow_does():
    """DeviceBatchTransform.from_jpeg (files -> decode on the GPU -> crop / bicubic resize / flip / normalise where the bytes already
    are) gives the batch that decoding with Pillow and transforming with __call__ gives, bit for bit."""
    from swat_amd import transforms as T
    rng = np.random.default_rng(6)
    files = [jpeg_bytes(synth(rng, int(rng.integers(100, 420)), int(rng.integers(100, 420))), int(rng.integers(50, 96)), int(rng.integers(0, 3)))
             for _ in range(9)]
    imgs = [Image.open(io.BytesIO(f)).convert("RGB") for f in files]
    for mode in ("train", "val"):
        tf = T.DeviceBatchTransform(224, mode, "cuda:0")
        torch.manual_seed(8)
        geo = [tf.geometry(*im.size) for im in imgs]
        want = tf(imgs, geometry=geo)
        got = tf.from_jpeg(files, geometry=geo)
        assert torch.equal(got, want)
        torch.manual_seed(8)
        assert torch.equal(tf.from_jpeg(files), want)    # geometry drawn inside, from the header sizes: the same draws


def test_loader_with_gpu_decode_matches_host_loader(tmp_path):
    """real JPEG files (+ one PNG, one progressive JPEG), workers that only read them, decode + transform on the GPU: the batches of
    the reference-shaped loader whose workers decode and transform with Pillow"""
    from types import SimpleNamespace
    from swat_amd import dataloader as dl, transforms as T
    rng = np.random.default_rng(22)
    os.makedirs(tmp_path / "img")
    lines = []
    for i in range(12):
        h, w = int(rng.integers(120, 400)), int(rng.integers(120, 400))
        name = f"img/{i}.jpg" if i != 7 else f"img/{i}.png"
        if i == 7:
            Image.fromarray(synth(rng, h, w)).save(tmp_path / name)
        else:
            Image.fromarray(synth(rng, h, w)).save(tmp_path / name, "JPEG", quality=int(rng.integers(60, 96)), subsampling=int(rng.integers(0, 3)),
                                                   progressive=(i == 3))
        lines.append(f"{name} {i % 3} {i % 2}\n")
    (tmp_path / "split.txt").write_text("".join(lines))
    split = [["split.txt"], [str(tmp_path)]]
    args = SimpleNamespace(dataset_root=str(tmp_path), bsz=4, num_workers=0, device="cuda:0", host_transform=True)
    val_tf = T.build_transform(224, "val")
    torch.manual_seed(3)
    want = [list(l) for l in dl.get_dataloader(args, split, split, split, None, val_tf)]
    args.host_transform = False   # the default pipeline: workers read files, decode + transform on host threads / the GPU
    torch.manual_seed(3)
    loaders = dl.get_dataloader(args, split, split, split, None, val_tf)
    assert all(isinstance(l, dl.DeviceTransformLoader) and l.decode for l in loaders)
    got = [list(l) for l in loaders]
    for lw, lg in zip(want, got):
        assert len(lw) == len(lg) > 0
        for bw, bg in zip(lw, lg):
            assert bg[0].is_cuda and torch.equal(bg[0].cpu(), bw[0])
            assert all(torch.equal(a, b) for a, b in zip(bg[1:], bw[1:]))
    args.num_workers = 2
    assert all(torch.equal(a[0].cpu(), b[0]) for a, b in zip(dl.get_dataloader(args, split, split, split, None, val_tf)[1], want[1]))
    # the training loop's host RNG order with worker processes and the decode running one batch AHEAD on a producer
    # thread: per epoch the loader's iterator draws its seeds from torch's global generator when the loop creates it, per iteration
    # the loop draws the CutMix scalars and permutation — shuffled batches, pixels and draws must be those of the reference-shaped
    # loader, two epochs from the same seeds (a DataLoader iterator created on the producer thread would race with the loop's draws)
    import types
    from swat_amd import training
    cm = types.SimpleNamespace(cutmix_beta=1.0, mix_prob=0.5)

    def run(host):
        args.host_transform = host
        torch.manual_seed(9)
        np.random.seed(9)
        tl = dl.get_dataloader(args, split, split, split, None, val_tf)[0]
        out = []
        for epoch in range(2):
            for x, y, tok, src in tl:
                lam, perm, box = training.cutmix_draw(cm, x.shape)
                out.append((x.cpu(), y.clone(), lam, None if perm is None else perm.clone(), box))
        return out
    a, b = run(True), run(False)
    assert len(a) == len(b) == 2 * (12 // 4)
    for (xa, ya, la, pa, ba), (xb, yb, lb, pb, bb) in zip(a, b):
        assert torch.equal(xa, xb) and torch.equal(ya, yb) and la == lb and ba == bb and (pa is None) == (pb is None)
        assert pa is None or torch.equal(pa, pb)


def test_stage_two_feature_extraction_takes_the_default_pipeline_and_equals_the_host_pipeline(tmp_path):
    """extract_dataloader / extract_train_dataloader (ref utils/dataloader.py:11-31,278-296: the cached features stage 2 probes on): with
    the default input pipeline the features are those of the reference-shaped loaders (Pillow in the workers), bit for bit (the training form's RNG order
    is covered by test_loader_with_gpu_decode_matches_host_loader)."""
    from types import SimpleNamespace
    from swat_amd import dataloader as dl, jpeg, transforms as T
    from swat_amd.clip_model import CLIP
    rng = np.random.default_rng(4)
    os.makedirs(tmp_path / "img")
    lines = []
    for i in range(10):
        h, w = int(rng.integers(70, 160)), int(rng.integers(70, 160))
        Image.fromarray(synth(rng, h, w)).save(tmp_path / f"img/{i}.jpg", "JPEG", quality=88, subsampling=int(rng.integers(0, 3)))
        lines.append(f"img/{i}.jpg {i % 3} 1\n")
    (tmp_path / "split.txt").write_text("".join(lines))
    split = [["split.txt"], [str(tmp_path)]]
    torch.manual_seed(1)
    model = CLIP(vision_cfg=(64, 16, 128, 2, 2, 64, "quick_gelu"), text_cfg=(8, 64, 64, 1, 1)).to("cuda:0").eval()
    pre = T.build_transform(64, "val")

    def run(host):
        args = SimpleNamespace(dataset_root=str(tmp_path), num_workers=2, device="cuda:0", host_transform=host)
        torch.manual_seed(5)
        n0 = jpeg.stats["device"]
        val = dl.extract_dataloader(args, model, split, str(tmp_path / f"val{int(host)}.pth"), pre, None)
        return torch.load(tmp_path / f"val{int(host)}.pth", weights_only=False), jpeg.stats["device"] - n0, len(val)
    (fh, nh, lh), (fg, ng, lg) = run(True), run(False)
    assert nh == 0 and ng == 10 and lh == lg
    assert torch.equal(fh["image_features"], fg["image_features"]) and torch.equal(fh["labels"], fg["labels"])
