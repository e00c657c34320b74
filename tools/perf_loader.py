"""Dev tool: the loader leg of bench.py alone (files on disk -> normalised batches through the CLI's loaders)."""
import os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
import bench
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
h, w = 375, 500
yy, xx = np.mgrid[0:h, 0:w]
files = []
for i in range(32):
    a = np.stack([128 + 100 * np.sin(xx / (7.0 + i) + yy / 11.0), 128 + 90 * np.cos(xx / 5.0 - yy / (3.0 + i)), (xx * 3 + yy * 5 + 17 * i) % 256], -1)
    a = np.clip(a + rng.normal(0, 10, a.shape), 0, 255).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(a).save(b, "JPEG", quality=90, subsampling=2); files.append(b.getvalue())
for workers in (8, 16):
    print(workers, bench.loader_throughput(dev, files, 32, w, h, workers=workers))
