#!/usr/bin/env python3
"""tools/micro/fwd_ab.py <libA.so> <libB.so> — the ViT-B/32 forward / forward + backward at B = 32 under two builds of the library, each in
its own child process, alternating (box-to-box and run-to-run differences are 1-3 %: an A/B has to alternate on ONE box)."""
import os
import subprocess
import sys

CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from swat_amd.clip_model import CLIP
torch.manual_seed(0)
dev = torch.device("cuda:0")
model = CLIP("ViT-B-32-quickgelu").to(dev)
v = model.visual
x = torch.randn(32, 3, 224, 224, device=dev)
df = torch.randn(32, v.output_dim, device=dev) * 0.05
for _ in range(10):
    v.forward_raw(x, save=True); v.backward_raw(df)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    v.forward_raw(x, save=True)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(n):
    v.forward_raw(x, save=True); v.backward_raw(df)
torch.cuda.synchronize()
t2 = time.perf_counter()
print("forward %%.4f ms   forward+backward %%.4f ms" %% (1e3 * (t1 - t0) / n, 1e3 * (t2 - t1) / n))
'''

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = sys.argv[1:3]
for rep in range(3):
    for lib in libs:
        env = dict(os.environ)
        if lib != "-":
            env["SWAT_HIP_LIB"] = os.path.abspath(lib)
        else:
            env.pop("SWAT_HIP_LIB", None)
        out = subprocess.run([sys.executable, "-c", CHILD % root], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.strip()
        print(f"{os.path.basename(lib):24s} {out}", flush=True)
