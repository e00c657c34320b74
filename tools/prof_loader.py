"""Dev tool: where a batch's time goes in the decode loader's producer (files on disk -> device batch)."""
import os, sys, io, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from types import SimpleNamespace
from swat_amd import dataloader as dl, transforms as T, jpeg
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
h, w = 375, 500
yy, xx = np.mgrid[0:h, 0:w]
root = tempfile.mkdtemp(prefix="swat_loader_")
os.makedirs(root + "/img")
lines = []
for i in range(2048):
    if i < 32:
        a = np.stack([128 + 100 * np.sin(xx / (7.0 + i) + yy / 11.0), 128 + 90 * np.cos(xx / 5.0 - yy / (3.0 + i)), (xx * 3 + yy * 5 + 17 * i) % 256], -1)
        a = np.clip(a + rng.normal(0, 10, a.shape), 0, 255).astype(np.uint8)
        Image.fromarray(a).save(f"{root}/img/{i}.jpg", "JPEG", quality=90, subsampling=2)
    else:
        shutil.copy(f"{root}/img/{i % 32}.jpg", f"{root}/img/{i}.jpg")
    lines.append(f"img/{i}.jpg {i % 200} 1\n")
open(root + "/split.txt", "w").write("".join(lines))
split = [["split.txt"], [root]]
args = SimpleNamespace(dataset_root=root, bsz=32, num_workers=8, device=dev)
tl, _, _ = dl.get_dataloader(args, split, split, split, None, T.build_transform(224, "val"))
t = {"next": 0.0, "slice": 0.0, "decode": 0.0, "transform": 0.0}
for ep in range(2):
    it = iter(tl.loader)
    n = 0
    while True:
        t0 = time.perf_counter()
        try:
            (blob, lens, geo, sizes), *rest = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        a, e, files = blob.numpy(), 0, []
        for k in lens:
            files.append(a[e:e + k]); e += k
        t2 = time.perf_counter()
        pixels, offsets, shapes = jpeg.decode_batch(files, dev, sizes=sizes, threads=0)
        t3 = time.perf_counter()
        out = tl.tf.from_jpeg(files, geometry=geo, sizes=sizes)
        t4 = time.perf_counter()
        if ep == 1 and n > 2:
            t["next"] += t1 - t0; t["slice"] += t2 - t1; t["decode"] += t3 - t2; t["transform"] += (t4 - t3) - (t3 - t2)
        n += 1
print({k: round(1e3 * v / (n - 3), 3) for k, v in t.items()}, "ms per batch;", n, "batches")
shutil.rmtree(root)
