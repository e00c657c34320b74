"""Dev tool: mined-feature extraction END TO END (path B1: files on disk -> mined.pth) with the default input pipeline (workers read
files, GPU decode + transform, caption pass without pixels) against the reference-shaped pipeline (--host_transform: Pillow decode +
host transform in the workers, both passes).  ViT-B/32 + text tower, random weights, 8 192 copies of 32 synthetic 500 x 375 JPEGs.
  python3 tools/perf_extract_e2e.py [n_files] [workers]"""
import io, os, pickle, shutil, sys, tempfile, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from swat_amd.clip_model import CLIP
from swat_amd.retrieval import extract_mined_feature as ex
from swat_amd.transforms import build_transform

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(0)
h, w = 375, 500
yy, xx = np.mgrid[0:h, 0:w]
blobs = []
for i in range(32):
    a = np.stack([128 + 100 * np.sin(xx / (7.0 + i) + yy / 11.0), 128 + 90 * np.cos(xx / 5.0 - yy / (3.0 + i)), (xx * 3 + yy * 5 + 17 * i) % 256], -1)
    b = io.BytesIO(); Image.fromarray(np.clip(a + rng.normal(0, 10, a.shape), 0, 255).astype(np.uint8)).save(b, "JPEG", quality=90, subsampling=2)
    blobs.append(b.getvalue())
tmp = tempfile.mkdtemp(prefix="swat_extract_")
root = os.path.join(tmp, "toy", "toy_retrieved")
cap = {}
for c in range(16):
    os.makedirs(os.path.join(root, str(c)))
    cap[str(c)] = {}
for i in range(n_files):
    c = i % 16
    with open(os.path.join(root, str(c), f"{i}.jpg"), "wb") as f:
        f.write(blobs[i % 32])
    cap[str(c)][str(i)] = "a photo of a small bird on a branch near the water"
cap_path = os.path.join(tmp, "toy.map")
pickle.dump(cap, open(cap_path, "wb"))
torch.manual_seed(0)
model = CLIP("ViT-B-32-quickgelu").to("cuda").eval()
vocab = 49408


def tokenizer(texts, ctx=77):   # (a hash tokenizer: the BPE vocabulary file is not in the image; same tensor shapes and cost class)
    out = torch.zeros(len(texts), ctx, dtype=torch.long)
    for r, t in enumerate(texts):
        ids = [49406] + [hash(wd) % 40000 + 1000 for wd in t.split()] + [49407]
        out[r, :len(ids)] = torch.tensor(ids)
    return out


pre = build_transform(224, "val")
try:
    for name, host in (("default pipeline (GPU decode + transform, caption pass without pixels)", False), ("--host_transform (the reference's pipeline)", True)):
        args = types.SimpleNamespace(root=tmp, bsz=1024, num_workers=workers, host_transform=host)
        ex.pre_extract_mined_fea(args, "toy", root, cap_path, "cfg", model, pre, tokenizer)   # warm-up (allocations, worker start)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = ex.pre_extract_mined_fea(args, "toy", root, cap_path, "cfg", model, pre, tokenizer)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name}: {n_files} image + caption pairs in {dt:.2f} s = {n_files / dt:,.0f} pairs/s ({workers} workers)")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
