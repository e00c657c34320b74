#!/usr/bin/env python3
"""bench.py — headline benchmark of swat_amd on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): ViT-B/32 (open_clip ViT-B-32-quickgelu, random init) finetuning,
B = 32 images / GPU, 200 classes, synthetic N(0,1) 224x224 images resident in HBM, one "step" = CutMix + encode_image +
normalise + head + scaled mixed CE + backward + (gradient all-reduce over RCCL) + fused AdamW + LR step.
bf16 MFMA operands, fp32 accumulation / residual stream / master weights.

Prints ONE JSON line (rank 0): whole-job img/s, the MFMA roofline of the GEMM kernels (HIP start/stop events attached to
every GEMM dispatch in an instrumented replay of the same steps, lanes serialised), a CPU baseline (the fp32 oracle on
the host cores, N = 1 only) and, as secondary records, the retrieval scan and the dense exact top-k (path B).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time
import types

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMG_STEP = 26.453  # ViT-B/32 fwd+bwd, SURVEY §8(d) / BASELINE.md §2
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def vit_gflop_per_img(d, L, T, patch_cols, embed, mlp_ratio=4):
    """fwd + bwd matmul flops of one image (3 x forward; the patch embedding has no data gradient), as SURVEY 8(d) counts."""
    per_tok = 2 * d * (3 * d) + 2 * d * d + 2 * 2 * d * (mlp_ratio * d)         # qkv, out, fc, proj
    attn = 2 * 2 * T * d                                                        # QK^T and PV per token
    fwd = L * T * (per_tok + attn) + 2 * (T - 1) * patch_cols * d + 2 * d * embed
    return round(3 * fwd / 1e9 - 2 * (T - 1) * patch_cols * d / 1e9, 3)


class _Lg:
    handlers = []

    def info(self, *a):
        pass


def build_job(dev, world, bsz=32, n_cls=200, arch="ViT-B-32-quickgelu", total_iter=10000, trail=True):
    from swat_amd import optimizers
    from swat_amd.clip_model import CLIP
    from swat_amd.models import MyLinear
    from swat_amd.step import FusedTrainStep
    torch.manual_seed(0)
    model = CLIP(arch).to(dev)
    W = torch.randn(n_cls, model.visual.output_dim)
    W /= W.norm(dim=-1, keepdim=True)
    head = MyLinear(weights=W).to(dev)
    args = types.SimpleNamespace(method="cutmix", device=dev, lr_classifier=1e-4, lr_backbone=1e-6, temperature=0.07,
                                 optim="AdamW", wd=1e-2, epochs=1, cutmix_beta=1.0, mix_prob=0.5, loss_name="CE")
    params, logit_scale = optimizers.set_params(args, model, head, _Lg())
    opt, sched, _ = optimizers.set_optimizer(args, params, [None] * total_iter)
    step = FusedTrainStep(model, head, logit_scale, opt, world_size=world, trail=trail)
    return args, model, head, step, sched


def run_steps(args, step, sched, images, labels, n):
    from swat_amd.training import cutmix_draw
    for _ in range(n):
        lam, perm, box = cutmix_draw(args, images.size())
        if perm is not None:
            perm = perm.to(images.device, non_blocking=True)
            step(images, labels, labels[perm], lam, perm=perm, box=box)
        else:
            step(images, labels, labels, 1.0)
        sched.step()


def gemm_roofline(args, step, sched, images, labels, n_steps, bsz, overlap=1):
    from swat_amd import _lib
    lib = _lib.lib()
    trail, step.trail = step.trail, False
    lib.swat_vit_set_overlap(0)  # per-launch durations of kernels running alone (the timed region overlaps streams)
    lib.swat_gemm_prof_begin(400 * n_steps)
    run_steps(args, step, sched, images, labels, n_steps)
    torch.cuda.synchronize()
    lib.swat_vit_set_overlap(min(overlap, 1))
    step.trail = trail
    ms = (ctypes.c_double * 9)()
    cnt = (ctypes.c_int64 * 9)()
    fl = (ctypes.c_double * 9)()
    n = lib.swat_gemm_prof_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p))
    names = ["NT_64", "NT_128", "NT_pc", "NN_64", "NN_128", "NN_pc", "TN_64", "TN_128", "TN_pc"]
    total_ms = sum(ms) / n_steps
    # the T = 50 attention products run in the one-workgroup-per-head kernels, not in gemm_*: take their share out
    attn_flops = 3 * 2 * (2.0 * 50 * 50 * 64) * bsz * 12 * 12
    algo_flops = GFLOP_PER_IMG_STEP * 1e9 * bsz - attn_flops
    achieved = algo_flops / (total_ms * 1e-3) / 1e12
    per_kind = {names[i]: {"launches_per_step": cnt[i] / n_steps, "avg_us": (1e3 * ms[i] / cnt[i]) if cnt[i] else 0.0,
                           "tflops": (fl[i] / (ms[i] * 1e-3) / 1e12) if ms[i] else 0.0} for i in range(9) if cnt[i]}
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "r01_gemm_hbm_traffic.json")
    if os.path.exists(tpath):  # PMC passes cannot run inside this process: the committed rocprofv3 --pmc result of this command
        traffic = round(json.load(open(tpath))["hbm_bytes_per_gemm_launch"])
        traffic_src = "profiles/r01_gemm_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per GEMM launch)"
    return {"bound": "mfma", "kernel": "gemm_pc_kernel / gemm_dma_kernel / gemm_ring2_kernel / gemm_ring_kernel (all GEMM instantiations)",
            "achieved": round(achieved, 2),
            "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "traffic_source": traffic_src, "algorithmic_flops_per_launch": round(algo_flops / (n / n_steps)),
            "gemm_ms_per_step": round(total_ms, 4), "gemm_launches_per_step": n / n_steps,
            "algorithmic_gflop_per_step": round(algo_flops / 1e9, 1), "note": "846.5 GFLOP per step minus 8.8 GFLOP of attention products that run outside gemm_*", "per_instantiation": per_kind}


def cpu_baseline(bsz=32, n_cls=200, budget_s=20.0):
    """The fp32 oracle (plain PyTorch on the host cores): same step, B = 32; 1 warm-up + as many steps as fit ~20 s."""
    from oracle import vit_oracle as vo
    torch.manual_seed(0)
    clip = vo.OracleCLIP("ViT-B-32-quickgelu", with_text=False)
    head = torch.nn.Linear(512, n_cls, bias=False)
    ls = torch.nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
    opt = torch.optim.AdamW([{"params": head.parameters(), "lr": 1e-4}, {"params": clip.visual.parameters(), "lr": 1e-6},
                             {"params": [ls], "lr": 1e-4}], weight_decay=1e-2)
    x = torch.randn(bsz, 3, 224, 224)
    y = torch.randint(0, n_cls, (bsz,))
    perm = torch.randperm(bsz)

    def one():
        xm = vo.cutmix_paste(x, perm, (30, 40, 130, 160))
        loss, _ = vo.head_loss(clip.visual(xm), head.weight, ls, y, y[perm], 0.7)
        opt.zero_grad()
        loss.backward()
        opt.step()

    one()
    t0 = time.time()
    n = 0
    while n < 2 or time.time() - t0 < budget_s:
        one()
        n += 1
        if n >= 30:
            break
    dt = time.time() - t0
    return {"value": round(bsz * n / dt, 2), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} steps of the fp32 oracle (oracle/vit_oracle.py, torch CPU) at B={bsz}, ViT-B/32, after 1 warm-up"}


def retrieval_secondary(dev, steps=5):
    """Path B on one shard (1.25 M x 512 fp32 = the per-GPU share of 10 M rows): T2T-rank + T2I-threshold sampler,
    1000 class queries, k = 500.  HBM roofline of the similarity scan (2048 B/row/matrix)."""
    from swat_amd.retrieval import ops
    n, d, nseg, k = 1_250_000, 512, 1000, 500
    g = torch.Generator(device=dev).manual_seed(7)
    cap = torch.randn(n, d, device=dev, generator=g)
    cap /= cap.norm(dim=-1, keepdim=True)
    img = torch.randn(n, d, device=dev, generator=g)
    img /= img.norm(dim=-1, keepdim=True)
    q = torch.randn(nseg, d, device=dev, generator=g)
    q /= q.norm(dim=-1, keepdim=True)
    per = n // nseg
    off = torch.arange(0, n + 1, per, device=dev, dtype=torch.int64)
    row_seg = torch.repeat_interleave(torch.arange(nseg, device=dev, dtype=torch.int32), per)
    s1 = torch.empty(n, device=dev)
    s2 = torch.empty(n, device=dev)

    def once():
        ops.similarity_segmented(cap, q, row_seg, out=s1)
        ops.similarity_segmented(img, q, row_seg, out=s2)
        return ops.segment_topk(s1, off, k, thr=0.0, t2i=s2, t2i_thr=0.0)

    once()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    for _ in range(steps):
        once()
    e[1].record()
    e[2].record()
    for _ in range(steps):
        ops.similarity_segmented(cap, q, row_seg, out=s1)
    e[3].record()
    torch.cuda.synchronize()
    ms_all = e[0].elapsed_time(e[1]) / steps
    ms_sim = e[2].elapsed_time(e[3]) / steps
    gbs = n * d * 4 / 1e9 / (ms_sim * 1e-3)
    return {"metric": "retrieval top-k queries/s (T2T-rank-T2I-tshd, per-class queries)", "value": round(nseg / (ms_all * 1e-3), 1),
            "unit": "queries/s", "config": {"workload": f"{n} x {d} fp32 rows x 2 matrices, {nseg} class segments, k={k}"},
            "ms_per_pass": round(ms_all, 3),
            "roofline": {"bound": "hbm", "kernel": "sim_segmented_kernel", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "avg_ms": round(ms_sim, 4)}}


def finetune_secondary(dev, arch, gflop_per_img, bsz=32, steps=10, warmup=4):
    """The same fused step on another BASELINE config (configs[2] ViT-B/16, configs[4] ViT-L/14), one GPU, B = 32."""
    args, model, head, step, sched = build_job(dev, 1, bsz, arch=arch)
    g = torch.Generator(device=dev).manual_seed(4321)
    images = torch.randn(bsz, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 200, (bsz,), device=dev, generator=g)
    run_steps(args, step, sched, images, labels, warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args, step, sched, images, labels, steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert math.isfinite(step.loss.item()), "training diverged"
    del step, model, head
    return {"metric": f"finetune img/s ({arch} bsz={bsz})", "value": round(bsz / dt, 1), "unit": "img/s",
            "ms_per_step": round(dt * 1e3, 3), "steps": steps,
            "config": {"workload": f"{arch} (random init) --method cutmix finetune step, 200 classes, one GPU"},
            "step_mfma_frac": round(gflop_per_img * 1e9 * bsz / dt / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "algorithmic_gflop_per_img": gflop_per_img}


_ARCH_GFLOP = {"ViT-B-16": (768, 12, 197, 3 * 16 * 16, 512), "ViT-L-14": (1024, 24, 257, 3 * 14 * 14, 768)}


def finetune_secondary_subprocess(arch):
    """finetune_secondary in a child process of its own, so every configuration is measured the way a training run of
    that configuration would start (fresh allocator, fresh streams)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--finetune-only", arch], capture_output=True, text=True,
                       timeout=1200)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"metric": f"finetune img/s ({arch} bsz=32)", "value": None, "error": (r.stderr or r.stdout)[-400:]}
    return json.loads(lines[-1])


def retrieval_dense_secondary(dev, steps=3):
    """BASELINE config 4 on one shard: N_query = 200 random unit queries against 1.25 M x 512 fp32 rows (the per-GPU share
    of 10 M), k = 500, exact (value desc, row asc) top-k on the fp32 matrix cores (157.3 TF peak)."""
    from swat_amd.retrieval import dense
    n, d, nq, k = 1_250_000, 512, 200, 500
    g = torch.Generator(device=dev).manual_seed(7)
    db = torch.randn(n, d, device=dev, generator=g)
    db /= db.norm(dim=-1, keepdim=True)
    q = torch.randn(nq, d, device=dev, generator=g)
    q /= q.norm(dim=-1, keepdim=True)
    dense.topk_dense(db, q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        idx, val, cnt = dense.topk_dense(db, q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tf = 2.0 * nq * n * d / dt / 1e12
    return {"metric": "retrieval top-k queries/s (dense N_query x N_db, exact fp32)", "value": round(nq / dt, 1), "unit": "queries/s",
            "config": {"workload": f"{nq} queries x {n} x {d} fp32 rows, k={k}, threshold-pruned exact top-k"},
            "ms_per_pass": round(dt * 1e3, 3),
            "roofline": {"bound": "mfma", "kernel": "sim_topk_candidates_wide_kernel (v_mfma_f32_32x32x2_f32; 128 DB rows x 224 queries per workgroup)", "achieved": round(tf, 1),
                         "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4), "traffic": None}}


def preprocess_secondary(dev, bsz=32, h=375, w=500, steps=50):
    """§8(f).2: the loaders' train transform (random-resized crop to 224 with Pillow's antialiased bicubic filter, flip,
    ToTensor, Normalize) for one batch of decoded 500 x 375 RGB images resident in HBM — three launches, byte / integer
    work.  Algorithmic bytes = cropped source bytes read + fp32 output written."""
    import numpy as np
    from swat_amd import _lib, transforms as T
    lib = _lib.lib()
    tf = T.DeviceBatchTransform(224, "train", dev)
    torch.manual_seed(0)
    geo = [tf.geometry(w, h) for _ in range(bsz)]
    desc = torch.tensor([(w, h) + g[:8] + (g[8], 0) for g in geo], dtype=torch.int32, device=dev)
    offs = torch.arange(bsz, dtype=torch.int64, device=dev) * (h * w * 3)
    rows = torch.tensor(np.concatenate([[0], np.cumsum([g[3] for g in geo])]), dtype=torch.int64, device=dev)
    pix = torch.randint(0, 256, (bsz * h * w * 3 + 4,), dtype=torch.uint8, device=dev)
    ks = max(max(T._axis_ksize(g[2], g[4]), T._axis_ksize(g[3], g[5])) for g in geo)
    ws = torch.empty(lib.swat_preprocess_workspace_bytes(bsz, 224, ks, int(rows[-1])), dtype=torch.uint8, device=dev)
    out = torch.empty(bsz, 3, 224, 224, device=dev)
    s = torch.cuda.current_stream(dev)

    def run():
        _lib.check(lib.swat_preprocess_batch(pix.data_ptr(), offs.data_ptr(), desc.data_ptr(), rows.data_ptr(), tf._lut.data_ptr(),
                                             out.data_ptr(), bsz, 224, max(g[2] for g in geo), max(g[3] for g in geo), ks,
                                             ws.data_ptr(), ws.numel(), s.cuda_stream), "swat_preprocess_batch")
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / steps * 1e-3
    nbytes = sum(3 * g[2] * g[3] for g in geo) + bsz * 12 * 224 * 224
    return {"metric": "image transform img/s (crop + bicubic resize + flip + normalise, decoded bytes -> fp32 batch)",
            "value": round(bsz / dt, 1), "unit": "img/s", "config": {"workload": f"{bsz} x {w}x{h} RGB uint8 -> [{bsz},3,224,224] fp32, train transform"},
            "ms_per_pass": round(dt * 1e3, 4), "dtype": "u8/int32",
            "roofline": {"bound": "hbm", "kernel": "resample_table_kernel + resample_rows_fixed_kernel + resample_cols_fixed_kernel (3 launches)",
                         "achieved": round(nbytes / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / dt / 8e12, 4), "traffic": None}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--bsz", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--overlap", type=int, default=2, help="0: one stream; 1: weight-gradient lane in the backward; "
                    "2 (default, what swat_amd.training runs): lane + per-stage all-reduce / AdamW trailing the backward")
    ap.add_argument("--finetune-only", default=None, help="internal: print one finetune_secondary record for this arch and exit")
    a = ap.parse_args()
    if a.finetune_only:
        torch.cuda.set_device(0)
        print(json.dumps(finetune_secondary(torch.device("cuda", 0), a.finetune_only, vit_gflop_per_img(*_ARCH_GFLOP[a.finetune_only]))),
              flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: swat_amd has no CPU fallback")
    # test hooks (tests/test_gpu_ddp.py): two ranks on the single GPU of a test box over gloo — RCCL refuses that
    backend = os.environ.get("SWAT_BENCH_TEST_BACKEND", "nccl")
    local = int(os.environ.get("SWAT_BENCH_TEST_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    args, model, head, step, sched = build_job(dev, world, a.bsz, trail=a.overlap == 2)
    from swat_amd import _lib
    _lib.lib().swat_vit_set_overlap(min(a.overlap, 1))
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn(a.bsz, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 200, (a.bsz,), device=dev, generator=g)
    np.random.seed(1)      # same CutMix scalars (r, lam, box) on every rank; the permutation is over the local shard
    torch.manual_seed(1)

    run_steps(args, step, sched, images, labels, a.warmup)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    run_steps(args, step, sched, images, labels, a.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = step.loss.item()
    assert math.isfinite(loss), "training diverged"

    out = None
    if rank == 0:
        ms = dt / a.steps * 1e3
        out = {"metric": "finetune img/s (ViT-B/32 bsz=32)", "value": round(world * a.bsz * a.steps / dt, 1), "unit": "img/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "ViT-B/32 (open_clip ViT-B-32-quickgelu, random init) --method cutmix finetune step, "
                                      "200 classes, AdamW 3 LR groups, fp32 master + bf16 MFMA", "per_gpu_batch": a.bsz,
                          "global_batch": a.bsz * world, "image": "3x224x224 fp32 resident in HBM",
                          "parallelism": f"dp{world}" if world > 1 else "single"},
               "final_loss": round(loss, 4),
               "step_mfma_frac": round(GFLOP_PER_IMG_STEP * 1e9 * a.bsz / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
    # the instrumented replay runs on every rank (its steps contain the gradient all-reduces); rank 0 reports its own
    roof = gemm_roofline(args, step, sched, images, labels, min(a.steps, 10), a.bsz, a.overlap)
    if rank == 0:
        out["roofline"] = roof
        out["cpu_baseline"] = None
    if world == 1:
        if not a.no_cpu_baseline:
            del step, model
            out["cpu_baseline"] = cpu_baseline(a.bsz)
        if not a.no_secondary:
            ft = [finetune_secondary_subprocess(arch) for arch in ("ViT-B-16", "ViT-L-14")]
            out["secondary"] = [retrieval_secondary(dev), retrieval_dense_secondary(dev), preprocess_secondary(dev)] + ft
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
