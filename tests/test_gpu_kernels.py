"""GPU: each path A HIP kernel (through the C ABI) against a plain PyTorch fp32 reference of the same op."""
import ctypes
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF, F32 = torch.bfloat16, torch.float32


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0, dtype=F32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev()).to(dtype)


# bf16 inputs are exact in fp32, products exact, fp32 accumulation: the only differences vs an fp32 matmul of
# the same bf16 inputs are summation order (~1e-6 rel) and the bf16 rounding of a bf16 output (2^-9 rel).
def close(got, ref, bf16_out, k):
    got, ref = got.float().cpu(), ref.float().cpu()
    tol = (2.0 ** -8 if bf16_out else 1e-5) * ref.abs().max().item() + 1e-5 * math.sqrt(k)
    err = (got - ref).abs().max().item()
    assert err <= tol, f"max err {err} > {tol}"


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (1600, 2304, 768), (1600, 768, 3072), (100, 72, 96), (1568, 768, 3072),
                                   (33, 200, 64), (257, 1024, 1024), (50, 50, 64)])
@pytest.mark.parametrize("out_dtype", [BF, F32])
def test_gemm_nt(M, N, K, out_dtype):
    from swat_amd import ops
    x, w = rnd(M, K, seed=1, dtype=BF), rnd(N, K, seed=2, scale=0.05, dtype=BF)
    b = rnd(N, seed=3) if N % 4 == 0 else None
    if N % 4:
        pytest.skip("N%4 handled via padded ldc in the attention tests")
    y = ops.linear_fwd(x, w, bias=b, out_dtype=out_dtype)
    ref = x.float() @ w.float().t() + b
    close(y, ref, out_dtype == BF, K)


@pytest.mark.parametrize("M,N,K", [(1600, 768, 2304), (1600, 3072, 768), (100, 72, 96), (257, 1024, 4096), (50, 64, 50)])
def test_gemm_nn_dgrad(M, N, K):
    """dx[M,K'] = dy[M,N'] @ w[N',K'] — here (M, N, K) = (rows, reduction, out cols)."""
    from swat_amd import ops
    if N % 8:
        pytest.skip("reduction ld must be a multiple of 8; ragged reductions are covered by the attention test")
    dy, w = rnd(M, N, seed=4, dtype=BF), rnd(N, K, seed=5, scale=0.05, dtype=BF)
    if K % 4:
        pytest.skip("ragged")
    dx = ops.linear_dgrad(dy, w, out_dtype=F32)
    close(dx, dy.float() @ w.float(), False, N)
    dxb = ops.linear_dgrad(dy, w, out_dtype=BF)
    close(dxb, dy.float() @ w.float(), True, N)


@pytest.mark.parametrize("M,N,K", [(1600, 768, 768), (1600, 2304, 768), (1568, 768, 3072), (100, 72, 96), (8224, 1024, 1024),
                                   (96, 64, 64)])
def test_gemm_tn_wgrad(M, N, K):
    from swat_amd import ops
    dy, x = rnd(M, N, seed=6, dtype=BF), rnd(M, K, seed=7, dtype=BF)
    dw = ops.linear_wgrad(dy, x)
    close(dw, dy.float().t() @ x.float(), False, M)


@pytest.mark.parametrize("M,N,K", [(1600, 3072, 768), (1600, 2304, 768), (1600, 768, 768), (448, 192, 128), (100, 72, 96),
                                   (8224, 1024, 1024)])
def test_linear_wgrad_bias(M, N, K):
    """dW and the bias gradient of one linear layer: fused into the TN GEMM (rows % 64 == 0) or GEMM + column sums."""
    from swat_amd import _lib
    lib = _lib.lib()
    dy, x = rnd(M, N, seed=8, dtype=BF), rnd(M, K, seed=9, dtype=BF)
    dw = torch.empty(N, K, device=dy.device)
    db = torch.full((N,), float("nan"), device=dy.device)
    ws = torch.empty(lib.swat_colsum_bf16_workspace_bytes(N), dtype=torch.uint8, device=dy.device)
    _lib.check(lib.swat_linear_wgrad_bias(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), _lib.ptr(db), M, N, K, N, K, _lib.ptr(ws),
                                          _lib.stream()), "swat_linear_wgrad_bias")
    close(dw, dy.float().t() @ x.float(), False, M)
    ref = dy.float().sum(0)
    assert (db - ref).abs().max().item() <= 1e-5 * math.sqrt(M) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("cfg", [50, 51, 53, 55, 56, 58, 70, 71, 72])
def test_gemm_producer_consumer_epilogues(cfg):
    """The producer / consumer kernel's LDS-staged epilogue: bias, activation + second output, in-place fp32 residual,
    dgrad through the activation, fp32 and bf16 outputs, ragged M edge (M = 200 is not a multiple of any tile height)."""
    from swat_amd import ops, _lib
    M, N, K = 200, 3072, 768
    x, w, b = rnd(M, K, seed=1, dtype=BF), rnd(N, K, seed=2, scale=0.05, dtype=BF), rnd(N, seed=3)
    _lib.lib().swat_gemm_set_tile(cfg)
    try:
        pre, post = ops.linear_fwd(x, w, bias=b, act="quick_gelu", want_act=True)
        ref_pre = x.float() @ w.float().t() + b
        close(pre, ref_pre, True, K)
        f = pre.float()
        close(post, f * torch.sigmoid(1.702 * f), True, 1)
        res = rnd(M, K, seed=9)
        w2 = rnd(K, N, seed=10, scale=0.02, dtype=BF)
        expect = res + post.float() @ w2.float().t() + b[:K]
        out = res.clone()
        ops.gemm(ops.NT, post, w2, M, K, N, N, N, out, K, bias=b[:K].contiguous(), residual=out)
        close(out, expect, False, N)
        dg = rnd(M, K, seed=11, dtype=BF)
        df = ops.linear_dgrad(dg, w2, actin=pre, act="quick_gelu")
        fr = pre.float().requires_grad_(True)
        (fr * torch.sigmoid(1.702 * fr)).backward(dg.float() @ w2.float())
        close(df, fr.grad, True, K)
        y32 = ops.linear_fwd(x, w, out_dtype=F32)
        close(y32, x.float() @ w.float().t(), False, K)
    finally:
        _lib.lib().swat_gemm_set_tile(-1)


@pytest.mark.parametrize("cfg", [50, 53, 54, 55, 56])
def test_gemm_producer_consumer_wgrad_bias(cfg):
    """Weight gradient + fused bias gradient (one extra MFMA against a ones fragment) on the producer / consumer kernel."""
    from swat_amd import _lib
    lib = _lib.lib()
    M, N, K = 1600, 2304, 768
    dy, x = rnd(M, N, seed=8, dtype=BF), rnd(M, K, seed=9, dtype=BF)
    dw = torch.empty(N, K, device=dy.device)
    db = torch.full((N,), float("nan"), device=dy.device)
    ws = torch.empty(lib.swat_colsum_bf16_workspace_bytes(N), dtype=torch.uint8, device=dy.device)
    lib.swat_gemm_set_tile(cfg)
    try:
        _lib.check(lib.swat_linear_wgrad_bias(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), _lib.ptr(db), M, N, K, N, K, _lib.ptr(ws),
                                              _lib.stream()), "swat_linear_wgrad_bias")
    finally:
        lib.swat_gemm_set_tile(-1)
    close(dw, dy.float().t() @ x.float(), False, M)
    ref = dy.float().sum(0)
    assert (db - ref).abs().max().item() <= 1e-5 * math.sqrt(M) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shapes", [[(1600, 768, 3072, False), (1600, 3072, 768, True)], [(1600, 768, 768, False), (1600, 2304, 768, True)],
                                    [(1600, 768, 3072, False), (1600, 3072, 768, True), (1600, 768, 768, False), (1600, 2304, 768, True)],
                                    [(448, 192, 128, True), (448, 96, 104, True), (448, 200, 64, False)], [(100, 72, 96, True), (100, 64, 64, False)]])
def test_linear_wgrad_bias_group(shapes):
    """Several layers' weight + bias gradients in one persistent launch over the joint tile list (ragged tile edges,
    tiles of several products per workgroup); rows % 64 != 0 takes the layer-by-layer path."""
    import ctypes
    from swat_amd import _lib
    lib = _lib.lib()
    n = len(shapes)
    dys, xs, dws, dbs = [], [], [], []
    for i, (M, N, K, bias) in enumerate(shapes):
        dys.append(rnd(M, N, seed=20 + i, dtype=BF)); xs.append(rnd(M, K, seed=30 + i, dtype=BF))
        dws.append(torch.full((N, K), float("nan"), device=dev()))
        dbs.append(torch.full((N,), float("nan"), device=dev()) if bias else None)
    VP = ctypes.c_void_p * n
    IA = ctypes.c_int * n
    ws = torch.empty(lib.swat_colsum_bf16_workspace_bytes(max(s[1] for s in shapes)), dtype=torch.uint8, device=dev())
    rc = lib.swat_linear_wgrad_bias_group(n, VP(*[t.data_ptr() for t in dys]), VP(*[t.data_ptr() for t in xs]), VP(*[t.data_ptr() for t in dws]),
                                          VP(*[(t.data_ptr() if t is not None else None) for t in dbs]), IA(*[s[0] for s in shapes]),
                                          IA(*[s[1] for s in shapes]), IA(*[s[2] for s in shapes]), IA(*[s[1] for s in shapes]),
                                          IA(*[s[2] for s in shapes]), _lib.ptr(ws), _lib.stream())
    _lib.check(rc, "swat_linear_wgrad_bias_group")
    for (M, N, K, bias), dy, x, dw, db in zip(shapes, dys, xs, dws, dbs):
        close(dw, dy.float().t() @ x.float(), False, M)
        if bias:
            ref = dy.float().sum(0)
            assert (db - ref).abs().max().item() <= 1e-5 * math.sqrt(M) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shapes", [[(1664, 768, 3072, False), (1664, 3072, 768, True), (1664, 768, 768, False), (1664, 2304, 768, True)],
                                    [(448, 384, 200, True), (448, 264, 136, True), (448, 512, 128, False)], [(64, 256, 128, True)],
                                    [(128, 8, 8, True), (192, 1024, 4096, False)]])
def test_block_weight_gradients_as_one_launch_of_256x128_tiles(shapes):
    """csrc/gemm_wgrad.hip: the dW = dY^T X products of a block (and the column sums of dY that are the bias gradients) over ONE
    joint list of 256 x 128 tiles — the ViT-B/32 block at B = 32 (216 tiles: the default route there), ragged tile edges, one
    k-tile, a one-fragment product — against fp32 torch, and bit-identical from run to run (fixed summation order)."""
    import ctypes
    from swat_amd import _lib
    lib = _lib.lib()
    n = len(shapes)
    dys, xs, dws, dbs = [], [], [], []
    for i, (M, N, K, bias) in enumerate(shapes):
        dys.append(rnd(M, N, seed=20 + i, dtype=BF)); xs.append(rnd(M, K, seed=30 + i, dtype=BF))
        dws.append(torch.full((N, K), float("nan"), device=dev()))
        dbs.append(torch.full((N,), float("nan"), device=dev()) if bias else None)
    VP = ctypes.c_void_p * n
    IA = ctypes.c_int * n
    ws = torch.empty(lib.swat_colsum_bf16_workspace_bytes(max(s[1] for s in shapes)), dtype=torch.uint8, device=dev())

    def run():
        rc = lib.swat_linear_wgrad_bias_group(n, VP(*[t.data_ptr() for t in dys]), VP(*[t.data_ptr() for t in xs]), VP(*[t.data_ptr() for t in dws]),
                                              VP(*[(t.data_ptr() if t is not None else None) for t in dbs]), IA(*[s[0] for s in shapes]),
                                              IA(*[s[1] for s in shapes]), IA(*[s[2] for s in shapes]), IA(*[s[1] for s in shapes]),
                                              IA(*[s[2] for s in shapes]), _lib.ptr(ws), _lib.stream())
        _lib.check(rc, "swat_linear_wgrad_bias_group")
    lib.swat_gemm_prof_begin(8)
    try:
        lib.swat_gemm_set_wgrad_big(2)
        run()
        first = [t.clone() for t in dws]
        run()
    finally:
        lib.swat_gemm_set_wgrad_big(1)
        ms, cnt, fl = (ctypes.c_double * 16)(), (ctypes.c_int64 * 16)(), (ctypes.c_double * 16)()
        lib.swat_gemm_prof_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p))
    assert cnt[10] == 2 and sum(cnt) == 2, "both calls must have been the grouped 256 x 128 launch"
    for (M, N, K, bias), dy, x, dw, db, f in zip(shapes, dys, xs, dws, dbs, first):
        close(dw, dy.float().t() @ x.float(), False, M)
        assert torch.equal(dw, f)
        if bias:
            ref = dy.float().sum(0)
            assert (db - ref).abs().max().item() <= 1e-5 * math.sqrt(M) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_gemm_epilogues(act):
    from swat_amd import ops
    M, N, K = 200, 3072, 768
    x, w, b = rnd(M, K, seed=1, dtype=BF), rnd(N, K, seed=2, scale=0.05, dtype=BF), rnd(N, seed=3)
    pre, post = ops.linear_fwd(x, w, bias=b, act=act, want_act=True)
    ref_pre = x.float() @ w.float().t() + b
    close(pre, ref_pre, True, K)
    f = pre.float()
    ref_post = f * torch.sigmoid(1.702 * f) if act == "quick_gelu" else torch.nn.functional.gelu(f)
    close(post, ref_post, True, 1)
    # residual (in place, fp32)
    res = rnd(M, K, seed=9)
    w2 = rnd(K, N, seed=10, scale=0.02, dtype=BF)
    expect = res + post.float() @ w2.float().t() + b[:K]
    out = res.clone()
    ops.gemm(ops.NT, post, w2, M, K, N, N, N, out, K, bias=b[:K].contiguous(), residual=out)
    close(out, expect, False, N)
    # dgrad through the activation: dF = (dG @ W) * act'(F)
    dg = rnd(M, K, seed=11, dtype=BF)
    df = ops.linear_dgrad(dg, w2, actin=pre, act=act)
    fr = pre.float().requires_grad_(True)
    (fr * torch.sigmoid(1.702 * fr) if act == "quick_gelu" else torch.nn.functional.gelu(fr)).backward(dg.float() @ w2.float())
    close(df, fr.grad, True, K)


@pytest.mark.parametrize("B,H,T,dh", [(2, 12, 50, 64), (1, 16, 257, 64), (3, 8, 77, 64), (2, 12, 197, 64)])
def test_attention_as_batched_gemms(B, H, T, dh):
    """softmax(QK^T/sqrt(dh)) V from the packed qkv buffer through batched NT / NN GEMMs + the softmax kernel,
    and its backward through TN / NT / NN GEMMs, vs torch autograd in fp32."""
    from swat_amd import attention
    d = H * dh
    qkv = rnd(B * T, 3 * d, seed=12, dtype=BF)
    do = rnd(B * T, d, seed=13, dtype=BF)
    out, P = attention.attn_fwd(qkv, B, T, H, dh)
    q, k, v = [t.reshape(B, T, H, dh).permute(0, 2, 1, 3).float().requires_grad_(True) for t in qkv.float().split(d, dim=1)]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * T, d)
    assert (out.float() - o).abs().max().item() <= 2.0 ** -7 * o.abs().max().item() + 1e-3
    o.backward(do.float())
    dqkv = attention.attn_bwd(qkv, P, do, B, T, H, dh)
    ref = torch.cat([g.grad.permute(0, 2, 1, 3).reshape(B * T, d) for g in (q, k, v)], dim=1)
    got, ref = dqkv.float().cpu(), ref.cpu()
    assert (got - ref).abs().max().item() <= 0.03 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("B,H,T", [(2, 12, 50), (3, 4, 64), (2, 3, 17), (1, 2, 1), (32, 12, 50)])
def test_attention_one_workgroup_per_head(B, H, T):
    """T <= 64, head width 64: QK^T / softmax / PV and the whole backward in one workgroup per (image, head), vs torch
    autograd in fp32 (same tolerances as the batched-GEMM form) — and within bf16 rounding of that form."""
    from swat_amd import _lib, attention
    lib = _lib.lib()
    dh, d = 64, H * 64
    qkv = rnd(B * T, 3 * d, seed=21, dtype=BF)
    do = rnd(B * T, d, seed=22, dtype=BF)
    out = torch.full((B * T, d), float("nan"), device=qkv.device, dtype=BF)
    dqkv = torch.full((B * T, 3 * d), float("nan"), device=qkv.device, dtype=BF)
    _lib.check(lib.swat_attn_fwd_small(_lib.ptr(qkv), _lib.ptr(out), B, T, d, H, _lib.stream()), "swat_attn_fwd_small")
    _lib.check(lib.swat_attn_bwd_small(_lib.ptr(qkv), _lib.ptr(do), _lib.ptr(dqkv), B, T, d, H, _lib.stream()), "swat_attn_bwd_small")
    q, k, v = [t.reshape(B, T, H, dh).permute(0, 2, 1, 3).float().requires_grad_(True) for t in qkv.float().split(d, dim=1)]
    p = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * T, d)
    assert (out.float() - o).abs().max().item() <= 2.0 ** -7 * o.abs().max().item() + 1e-3
    o.backward(do.float())
    ref = torch.cat([g.grad.permute(0, 2, 1, 3).reshape(B * T, d) for g in (q, k, v)], dim=1)
    assert (dqkv.float() - ref).abs().max().item() <= 0.03 * ref.abs().max().item() + 1e-3
    out2, P = attention.attn_fwd(qkv, B, T, H, dh)
    dqkv2 = attention.attn_bwd(qkv, P, do, B, T, H, dh)
    assert (out.float() - out2.float()).abs().max().item() <= 2.0 ** -7 * o.abs().max().item()
    assert (dqkv.float() - dqkv2.float()).abs().max().item() <= 0.02 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("B,H,T,causal", [(2, 3, 65, 0), (2, 12, 197, 0), (1, 16, 257, 0), (3, 8, 77, 1), (2, 2, 320, 0), (1, 2, 128, 1),
                                              (2, 4, 50, 0), (1, 1, 64, 1), (2, 2, 129, 0), (1, 3, 144, 0), (2, 2, 241, 0), (1, 2, 256, 1)])
def test_attention_tiled(B, H, T, causal):
    """64 < T <= 320 (ViT-B/16, ViT-L/14, the causal text tower): row-tile workgroups of 4..9 waves over LDS-staged key /
    query tiles, running-maximum softmax forward (+ row log-sum-exp), dQ per query tile and dK / dV per key tile backward;
    vs torch autograd.  The T values cover one workgroup per head (<= 144 rows), 8-wave tiles with idle waves (257),
    exact tiles (256) and a 1-row last fragment (129, 241, 257)."""
    from swat_amd import _lib
    lib = _lib.lib()
    d = 64 * H
    qkv = rnd(B * T, 3 * d, seed=31, dtype=BF)
    do = rnd(B * T, d, seed=32, dtype=BF)
    out = torch.full((B * T, d), float("nan"), device=qkv.device, dtype=BF)
    dqkv = torch.full((B * T, 3 * d), float("nan"), device=qkv.device, dtype=BF)
    lse = torch.full((B * H, T), float("nan"), device=qkv.device)
    _lib.check(lib.swat_attn_fwd_tiled(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(lse), B, T, d, H, causal, _lib.stream()), "swat_attn_fwd_tiled")
    q, k, v = [t.reshape(B, T, H, 64).permute(0, 2, 1, 3).float().requires_grad_(True) for t in qkv.float().split(d, dim=1)]
    s = (q @ k.transpose(-1, -2)) / 8.0
    if causal:
        s = s.masked_fill(torch.ones(T, T, device=qkv.device).triu(1).bool(), float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, d)
    assert (out.float() - o).abs().max().item() <= 2.0 ** -7 * o.abs().max().item() + 1e-3
    assert (lse - torch.logsumexp(s, -1).reshape(B * H, T)).abs().max().item() < 5e-3
    if causal:
        return  # the text tower is frozen: no backward
    ws = torch.empty(lib.swat_attn_bwd_tiled_workspace_bytes(B, T, H), device=qkv.device, dtype=torch.uint8)
    _lib.check(lib.swat_attn_bwd_tiled(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(do), _lib.ptr(lse), _lib.ptr(dqkv), B, T, d, H,
                                       _lib.ptr(ws), _lib.stream()), "swat_attn_bwd_tiled")
    o.backward(do.float())
    ref = torch.cat([g.grad.permute(0, 2, 1, 3).reshape(B * T, d) for g in (q, k, v)], dim=1)
    assert (dqkv.float() - ref).abs().max().item() <= 0.03 * ref.abs().max().item() + 1e-3
    # bit-identical when repeated (and without the lse output): every element is summed by one wave in a fixed order
    out2, dqkv2 = torch.empty_like(out), torch.empty_like(dqkv)
    _lib.check(lib.swat_attn_fwd_tiled(_lib.ptr(qkv), _lib.ptr(out2), None, B, T, d, H, causal, _lib.stream()), "swat_attn_fwd_tiled")
    _lib.check(lib.swat_attn_bwd_tiled(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(do), _lib.ptr(lse), _lib.ptr(dqkv2), B, T, d, H,
                                       _lib.ptr(ws), _lib.stream()), "swat_attn_bwd_tiled")
    assert torch.equal(out, out2) and torch.equal(dqkv, dqkv2)


@pytest.mark.parametrize("causal", [False, True])
def test_softmax_kernels(causal):
    from swat_amd import ops
    BH, T, ld = 5, 50, 64
    S = rnd(BH, T, ld, seed=14, scale=3.0)
    P = ops.softmax_fwd(S, T, T, 0.125, causal=causal)
    s = S[..., :T] * 0.125
    if causal:
        s = s.masked_fill(torch.ones(T, T, device=dev()).triu(1).bool(), float("-inf"))
    ref = torch.softmax(s, -1)
    assert (P[..., :T].float() - ref).abs().max().item() < 4e-3
    assert (P[..., T:] == 0).all()
    dP = rnd(BH, T, ld, seed=15)
    dS = ops.softmax_bwd(P, dP, T, 0.125)
    p = P[..., :T].float()
    refd = 0.125 * p * (dP[..., :T] - (p * dP[..., :T]).sum(-1, keepdim=True))
    assert (dS[..., :T].float() - refd).abs().max().item() < 2e-3
    assert (dS[..., T:] == 0).all()


@pytest.mark.parametrize("rows,d", [(1600, 768), (7, 1024), (8224, 1024), (33, 512)])
def test_layernorm_fwd_bwd(rows, d):
    from swat_amd import ops
    x = rnd(rows, d, seed=16, scale=2.0) + 0.5
    g, b = rnd(d, seed=17) * 0.1 + 1.0, rnd(d, seed=18) * 0.1
    y_bf, y_f, mean, rstd = ops.layernorm_fwd(x, g, b, out_f32=True)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-5)
    assert (y_f - ref).abs().max().item() < 2e-5
    assert (y_bf.float() - ref).abs().max().item() < 2.0 ** -7 * ref.abs().max().item()
    dy = rnd(rows, d, seed=19)
    add = rnd(rows, d, seed=20)
    ref.backward(dy)
    dx, dx_bf, dg, db = ops.layernorm_bwd(dy, x, g, mean, rstd, dx_add=add, want_bf16=True)
    assert (dx - (xr.grad + add)).abs().max().item() < 1e-4
    assert (dx_bf.float() - dx).abs().max().item() <= 2.0 ** -8 * dx.abs().max().item()
    assert (dg - gr.grad).abs().max().item() < 2e-3 * max(1.0, gr.grad.abs().max().item())
    assert (db - br.grad).abs().max().item() < 2e-3 * max(1.0, br.grad.abs().max().item())


def test_layernorm_strided_cls_rows():
    from swat_amd import ops
    B, T, d = 6, 50, 768
    x = rnd(B, T, d, seed=21)
    g, b = rnd(d, seed=22) + 1.0, rnd(d, seed=23)
    _, y, mean, rstd = ops.layernorm_fwd(x, g, b, rows=B, row_stride=T * d, out_bf16=False, out_f32=True)
    ref = torch.nn.functional.layer_norm(x[:, 0], (d,), g, b, 1e-5)
    assert (y - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("ps,H", [(32, 224), (16, 224), (14, 224)])
def test_patchify_cutmix(ps, H):
    from swat_amd import ops
    B = 5
    img = rnd(B, 3, H, H, seed=24)
    perm = torch.tensor([3, 0, 4, 1, 2], device=dev())
    box = (17, 40, 130, 201)  # bbx1, bby1, bbx2, bby2 as rand_bbox returns them
    mixed = img.clone()
    mixed[:, :, box[0]:box[2], box[1]:box[3]] = img[perm][:, :, box[0]:box[2], box[1]:box[3]]  # ref utils/training.py:1466
    g = H // ps
    K = 3 * ps * ps
    for use_mix, src in ((True, mixed), (False, img)):
        out = ops.patchify_cutmix(img, ps, perm if use_mix else None, box if use_mix else None)
        ref = src.reshape(B, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, K)  # conv1 im2col
        assert torch.equal(out[:, :K], ref.to(BF))
        assert (out[:, K:] == 0).all()
    assert torch.equal(ops.cutmix_paste(img, perm, box), mixed)


def test_colsum_and_cast():
    from swat_amd import ops
    X = rnd(1600, 2304, seed=25, dtype=BF)
    got = ops.colsum(X)
    assert (got - X.float().sum(0)).abs().max().item() < 2e-3
    Xf = rnd(6 * 50, 768, seed=26)
    got = ops.colsum(Xf, rows=6 * 49, n=768, ld=768, group=(49, 50, 1))
    ref = Xf.reshape(6, 50, 768)[:, 1:].reshape(-1, 768).sum(0)
    assert (got - ref).abs().max().item() < 1e-4
    assert torch.equal(ops.cast_bf16(Xf), Xf.to(BF))


@pytest.mark.parametrize("B,E,C", [(32, 512, 200), (7, 768, 1000), (64, 512, 37)])
def test_head_loss_one_launch_per_sample_equals_the_separate_launches_and_torch(B, E, C):
    """swat_head_loss with gradients (ref utils/training.py:1475-1481, utils/losses.py:10-11): the sample-local path as ONE launch with
    a workgroup per sample (csrc/tail.hip head_rows_kernel) against the five separate launches and against torch in float64 —
    loss, d feats, d W_head, d logit_scale; and the two-part form (want_grad 2 then 3) against the one-part form, bit for bit."""
    import ctypes
    from swat_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(B + C)
    feats = torch.randn(B, E, generator=g)
    W = torch.randn(C, E, generator=g)
    W /= W.norm(dim=1, keepdim=True)
    ls = torch.tensor([2.6593])
    ya, yb = torch.randint(0, C, (B,), generator=g), torch.randint(0, C, (B,), generator=g)
    lam = 0.37
    f64, W64 = feats.double().requires_grad_(True), W.double().requires_grad_(True)
    ls64 = ls.double().requires_grad_(True)
    z = (f64 / f64.norm(dim=1, keepdim=True)) @ W64.t() * ls64.exp()
    ref = lam * torch.nn.functional.cross_entropy(z, ya) + (1 - lam) * torch.nn.functional.cross_entropy(z, yb)
    ref.backward()

    def run(fused, parts):
        L.swat_head_set_fused(fused)
        d = dev()
        ws = torch.zeros(L.swat_head_loss_workspace_bytes(B, E, C), dtype=torch.uint8, device=d)
        loss = torch.zeros(1, device=d); dfe = torch.zeros(B, E, device=d); dW = torch.zeros(C, E, device=d); dls = torch.zeros(1, device=d)
        acc = torch.zeros(1, dtype=torch.float64, device=d)
        keep = [feats.to(d), W.to(d), ls.to(d), ya.to(d), yb.to(d)]   # (alive until the synchronise below)
        args = [_lib.ptr(keep[0]), _lib.ptr(keep[1]), _lib.ptr(keep[2]), _lib.ptr(keep[3]), _lib.ptr(keep[4]), lam, _lib.ptr(ws),
                _lib.ptr(loss), _lib.ptr(dfe), _lib.ptr(dW), _lib.ptr(dls), None, None, _lib.ptr(acc), B, E, C]
        for wg in parts:
            _lib.check(L.swat_head_loss(*args, wg, 1, _lib.stream()), "swat_head_loss")
        torch.cuda.synchronize()
        return loss.cpu(), dfe.cpu(), dW.cpu(), dls.cpu(), acc.cpu()

    def rel(a, b_):
        return ((a - b_).norm() / (b_.norm() + 1e-30)).item()

    try:
        sep = run(0, (1,))
        one = run(1, (1,))
        two = run(1, (2, 3))
    finally:
        L.swat_head_set_fused(1)
    for got in (sep, one):
        assert abs(got[0].item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
        assert rel(got[1].double(), f64.grad) < 2e-5 and rel(got[2].double(), W64.grad) < 2e-5
        assert abs(got[3].item() - ls64.grad.item()) < 2e-5 * max(1.0, abs(ls64.grad.item()))
        assert abs(got[4].item() - got[0].item()) < 1e-6
    for a, b_ in zip(one, two):
        assert torch.equal(a, b_)


def test_argmax_confusion_follows_torch_argmax_on_ties_and_nans():
    """swat_argmax_confusion (validate's device tail, ref testing.py:449-458): first index of the row maximum, NaN greater than
    everything (several NaNs: the first), like torch.argmax — also on diverged logits."""
    from swat_amd import _lib
    L = _lib.lib()
    B, C = 37, 203
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, C, generator=g)
    z[1, 5] = z[1, 150] = z[1].max() + 1.0          # a tie across lanes: the lower index
    z[2, 77] = float("nan")                          # one NaN
    z[3, 190] = z[3, 66] = float("nan")              # two NaNs: the first
    z[4, :] = float("nan")
    z[5, 0] = float("inf"); z[5, 100] = float("nan")
    y = torch.randint(0, C, (B,), generator=g)
    zd, yd = z.to(dev()), y.to(dev())
    pred = torch.empty(B, dtype=torch.int64, device=dev())
    conf = torch.zeros(C, C, dtype=torch.int64, device=dev())
    correct = torch.zeros(1, dtype=torch.int64, device=dev())
    _lib.check(L.swat_argmax_confusion(_lib.ptr(zd), _lib.ptr(yd), B, C, _lib.ptr(pred), None, _lib.ptr(conf), _lib.ptr(correct), _lib.stream()),
               "swat_argmax_confusion")
    want = torch.argmax(z, dim=1)
    assert torch.equal(pred.cpu(), want)
    ref = torch.zeros(C, C, dtype=torch.int64)
    for a, b in zip(y.tolist(), want.tolist()):
        ref[a, b] += 1
    assert torch.equal(conf.cpu(), ref) and correct.item() == int((want == y).sum())


def test_tail_kernels():
    from swat_amd import ops
    B, E, C = 32, 512, 200
    f = rnd(B, E, seed=27)
    W = rnd(C, E, seed=28, scale=0.05)
    ls = torch.tensor([math.log(1 / 0.07)], device=dev())
    ya = torch.randint(0, C, (B,), device=dev())
    yb = ya[torch.randperm(B, device=dev())]
    lam = 0.37
    fr, Wr, lsr = f.clone().requires_grad_(True), W.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    fn_ref = fr / fr.norm(dim=-1, keepdim=True)
    logits_ref = (fn_ref @ Wr.t()) * lsr.exp()
    ce = torch.nn.CrossEntropyLoss()
    loss_ref = lam * ce(logits_ref, ya) + (1.0 - lam) * ce(logits_ref, yb)  # mixup_criterion
    loss_ref.backward()
    fn, nrm = ops.l2norm_fwd(f)
    assert (fn - fn_ref).abs().max().item() < 1e-6
    z = ops.sgemm_small(fn, W, B, C, E, E, 1, 1, E)
    loss, dz, dls, logits = ops.ce_mixed(z, ls, ya, yb, lam, want_logits=True)
    assert abs(loss.item() - loss_ref.item()) < 1e-5
    assert (logits - logits_ref).abs().max().item() < 1e-4
    assert abs(dls.item() - lsr.grad.item()) < 1e-5
    dW = ops.sgemm_small(dz, fn, C, E, B, 1, C, E, 1)  # dW[c,e] = sum_b dz[b,c] fn[b,e]
    assert (dW - Wr.grad).abs().max().item() < 1e-6
    dfn = ops.sgemm_small(dz, W, B, E, C, C, 1, E, 1)  # dfn[b,e] = sum_c dz[b,c] W[c,e]
    df = ops.l2norm_bwd(fn, nrm, dfn)
    assert (df - fr.grad).abs().max().item() < 1e-6
    # plain CE = lam 1, yb = ya
    loss1, _, _, _ = ops.ce_mixed(z, ls, ya, ya, 1.0)
    assert abs(loss1.item() - ce(logits_ref.detach(), ya).item()) < 1e-5


def test_adamw_matches_torch():
    from swat_amd import ops
    n = 4096 * 3 + 4
    segs = [0, 4096, 4096 * 3]
    lrs = [1e-4, 1e-6, 1e-4]
    p0 = rnd(n, seed=29)
    tp = [p0[segs[i]:(segs + [n])[i + 1]].clone().requires_grad_(True) for i in range(3)]
    opt = torch.optim.AdamW([{"params": [t], "lr": lr, "betas": (0.9, 0.999), "weight_decay": 1e-2} for t, lr in zip(tp, lrs)])
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    shadow = torch.empty(n, dtype=BF, device=dev())
    lrs_dev = torch.tensor(lrs, device=dev())
    for step in range(1, 6):
        g = rnd(n, seed=100 + step, scale=0.1)
        for i, t in enumerate(tp):
            t.grad = g[segs[i]:(segs + [n])[i + 1]].clone()
        opt.step()
        bc = torch.tensor([1 - 0.9 ** step, 1 - 0.999 ** step], device=dev())
        ops.adamw_step(p, g, m, v, shadow, segs, lrs_dev, bc)
        ref = torch.cat([t.detach() for t in tp])
        assert (p - ref).abs().max().item() < 3e-6
    assert torch.equal(shadow, p.to(BF))


PC_LAYOUTS = {50: ("nt", "nn", "tn"), 51: ("nt", "nn"), 53: ("nt", "nn", "tn"), 54: ("tn",), 55: ("nt", "nn", "tn"),
              56: ("nt", "nn", "tn"), 58: ("nt", "nn"),
              70: ("nt", "nn", "tn"), 71: ("nt", "nn", "tn"), 72: ("nt", "nn", "tn"),  # 70-72: the persistent tile loop
              80: ("nt", "nn")}                                                        # 80: 256 x 256 (gemm_big.hip)


@pytest.mark.parametrize("cfg", [14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 30, 31, 32, 33, 34, 43,
                                 50, 51, 53, 54, 55, 56, 58, 70, 71, 72, 80])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_every_tile_configuration(cfg, layout):
    """All register-staged and LDS-DMA pipelined tile configurations give the same product (ragged M/N edges)."""
    from swat_amd import ops, _lib
    M, N, K = 1600, 840, 768  # 1600 = 12.5 x 128 rows, 840 = 6.56 x 128 cols: edge tiles on both sides
    a, b = rnd(M, K, seed=31, dtype=BF), rnd(N, K, seed=32, scale=0.05, dtype=BF)
    ref = a.float() @ b.float().t()
    if cfg >= 50 and layout not in PC_LAYOUTS[cfg]:
        pytest.skip("producer / consumer tile not built for this layout")
    _lib.lib().swat_gemm_set_tile(cfg)
    try:
        C = torch.empty(M, N, device=dev(), dtype=F32)
        if layout == "nt":
            ops.gemm(ops.NT, a, b, M, N, K, K, K, C, N)
        elif layout == "nn":
            bt = b.t().contiguous()  # [K, N]
            ops.gemm(ops.NN, a, bt, M, N, K, K, N, C, N)
        else:
            at, bt = a.t().contiguous(), b.t().contiguous()  # [K, M], [K, N]
            ops.gemm(ops.TN, at, bt, M, N, K, M, N, C, N)
    finally:
        _lib.lib().swat_gemm_set_tile(-1)
    close(C, ref, False, K)


@pytest.mark.parametrize("cfg", [14, 20, 43, 33, 51, 53])
def test_gemm_activated_output_only(cfg):
    """C = NULL with a C2 (the c_fc product of an inference forward): every kernel family stores act(round(pre-activation)) and
    nothing else — the bits of the two-output form."""
    from swat_amd import ops, _lib
    M, N, K = 1600, 3072, 768
    a = rnd(M, K, seed=51, dtype=BF)
    w = rnd(N, K, seed=52, scale=0.05, dtype=BF)
    bias = rnd(N, seed=53)
    _lib.lib().swat_gemm_set_tile(cfg)
    try:
        C = torch.empty(M, N, device=dev(), dtype=BF)
        C2 = torch.empty(M, N, device=dev(), dtype=BF)
        ops.gemm(ops.NT, a, w, M, N, K, K, K, C, N, bias=bias, act="quick_gelu", C2=C2)
        only = torch.full((M, N), float("nan"), device=dev(), dtype=BF)
        ops.gemm(ops.NT, a, w, M, N, K, K, K, None, N, bias=bias, act="quick_gelu", C2=only)
        assert torch.equal(only, C2)
    finally:
        _lib.lib().swat_gemm_set_tile(-1)


@pytest.mark.parametrize("layout", ["nt", "nn"])
@pytest.mark.parametrize("epi", ["bias_act_c2", "residual_f32", "actin"])
def test_gemm_256_tile_epilogues_on_many_rows(layout, epi):
    """gemm_big.hip (256 x 256, the many-row tile pick_cfg takes for ViT-B/16 / ViT-L/14 products) with the fused epilogues
    of the step, ragged edges on both sides (M = 8 224 = 32.1 tiles, N = 1 000 = 3.9 tiles), against fp32 torch."""
    from swat_amd import ops, _lib
    M, N, K = 8224, 1000, 1024
    a = rnd(M, K, seed=41, dtype=BF)
    w = rnd(N, K, seed=42, scale=0.05, dtype=BF)
    ref = a.float() @ w.float().t()
    B = w if layout == "nt" else w.t().contiguous()
    lay, ldb = (ops.NT, K) if layout == "nt" else (ops.NN, N)
    bias = rnd(N, seed=43)
    _lib.lib().swat_gemm_set_tile(80)
    try:
        if epi == "bias_act_c2":
            C = torch.empty(M, N, device=dev(), dtype=BF)
            C2 = torch.empty(M, N, device=dev(), dtype=BF)
            ops.gemm(lay, a, B, M, N, K, K, ldb, C, N, bias=bias, act="quick_gelu", C2=C2)
            want = (ref + bias).to(BF)
            close(C.float(), want.float(), True, K)
            wf = want.float()
            close(C2.float(), (C.float() * torch.sigmoid(1.702 * C.float())), True, 8)
            only = torch.empty(M, N, device=dev(), dtype=BF)   # inference form: C = NULL, only the activated output, same bits
            ops.gemm(lay, a, B, M, N, K, K, ldb, None, N, bias=bias, act="quick_gelu", C2=only)
            assert torch.equal(only, C2)
        elif epi == "residual_f32":
            C = rnd(M, N, seed=44)
            res = C.clone()
            ops.gemm(lay, a, B, M, N, K, K, ldb, C, N, bias=bias, residual=C)
            close(C, ref + bias + res, False, K)
        else:
            f = rnd(M, N, seed=45, dtype=BF)
            C = torch.empty(M, N, device=dev(), dtype=BF)
            ops.gemm(lay, a, B, M, N, K, K, ldb, C, N, actin=f, ldact=N, act="quick_gelu")
            x = f.float()
            sg = torch.sigmoid(1.702 * x)
            close(C.float(), ref * (sg * (1 + 1.702 * x * (1 - sg))), True, K)
    finally:
        _lib.lib().swat_gemm_set_tile(-1)


@pytest.mark.parametrize("layout", ["nt", "nn"])
@pytest.mark.parametrize("epi", ["bias_act_c2", "residual_f32", "actin", "plain_f32"])
@pytest.mark.parametrize("M", [8224, 1024])
def test_gemm_256x128_loader_consumer_tile_epilogues(layout, epi, M):
    """csrc/gemm_xl.hip (configuration 90: 256 x 128 tiles, 8 consumer + 4 loader waves, persistent over the tile list) with the
    fused epilogues of the step — M = 8 224 = 32 whole tile rows + 32 rows that leave through a second, small launch (ViT-L/14 at
    B = 32) and M = 1 024 (fewer tiles than workgroups) — against fp32 torch, bit-identical from run to run."""
    from swat_amd import ops, _lib
    N, K = 1024, 1024
    a = rnd(M, K, seed=41, dtype=BF)
    w = rnd(N, K, seed=42, scale=0.05, dtype=BF)
    ref = a.float() @ w.float().t()
    B = w if layout == "nt" else w.t().contiguous()
    lay, ldb = (ops.NT, K) if layout == "nt" else (ops.NN, N)
    bias = rnd(N, seed=43)
    lib = _lib.lib()
    lib.swat_gemm_set_tile(90)
    lib.swat_gemm_prof_begin(16)
    try:
        if epi == "bias_act_c2":
            C = torch.empty(M, N, device=dev(), dtype=BF)
            C2 = torch.empty(M, N, device=dev(), dtype=BF)
            ops.gemm(lay, a, B, M, N, K, K, ldb, C, N, bias=bias, act="quick_gelu", C2=C2)
            close(C.float(), (ref + bias).to(BF).float(), True, K)
            close(C2.float(), (C.float() * torch.sigmoid(1.702 * C.float())), True, 8)
            only = torch.empty(M, N, device=dev(), dtype=BF)
            ops.gemm(lay, a, B, M, N, K, K, ldb, None, N, bias=bias, act="quick_gelu", C2=only)
            assert torch.equal(only, C2)
        elif epi == "residual_f32":
            C = rnd(M, N, seed=44)
            res = C.clone()
            ops.gemm(lay, a, B, M, N, K, K, ldb, C, N, bias=bias, residual=C)
            close(C, ref + bias + res, False, K)
        elif epi == "actin":
            f = rnd(M, N, seed=45, dtype=BF)
            C = torch.empty(M, N, device=dev(), dtype=BF)
            ops.gemm(lay, a, B, M, N, K, K, ldb, C, N, actin=f, ldact=N, act="quick_gelu")
            x = f.float()
            sg = torch.sigmoid(1.702 * x)
            close(C.float(), ref * (sg * (1 + 1.702 * x * (1 - sg))), True, K)
        else:
            C = torch.full((M, N), float("nan"), device=dev())
            ops.gemm(lay, a, B, M, N, K, K, ldb, C, N)
            close(C, ref, False, K)
            again = torch.full((M, N), float("nan"), device=dev())
            ops.gemm(lay, a, B, M, N, K, K, ldb, again, N)
            assert torch.equal(C, again)
    finally:
        lib.swat_gemm_set_tile(-1)
        ms, cnt, fl = (ctypes.c_double * 16)(), (ctypes.c_int64 * 16)(), (ctypes.c_double * 16)()
        lib.swat_gemm_prof_end(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p))
    assert cnt[11] >= 1, "the product must have gone through the 256 x 128 loader / consumer kernel"


def test_gemm_256x128_loader_consumer_tile_weight_gradient_many_rows():
    """The same kernel on one deep weight gradient (ViT-L/14 c_fc at B = 32: 4 096 x 1 024 outputs, 8 256 padded rows): 128 tiles on
    128 workgroups, 129 k-tiles each, with the bias gradient's column sums."""
    from swat_amd import _lib
    lib = _lib.lib()
    R, Nout, Nin = 8256, 4096, 1024
    dy, x = rnd(R, Nout, seed=61, dtype=BF), rnd(R, Nin, seed=62, dtype=BF)
    dy[8224:].zero_(); x[8224:].zero_()
    dw = torch.full((Nout, Nin), float("nan"), device=dev())
    db = torch.full((Nout,), float("nan"), device=dev())
    ws = torch.empty(lib.swat_colsum_bf16_workspace_bytes(Nout), dtype=torch.uint8, device=dev())
    lib.swat_gemm_set_tile(90)
    try:
        _lib.check(lib.swat_linear_wgrad_bias(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), _lib.ptr(db), R, Nout, Nin, Nout, Nin, _lib.ptr(ws),
                                              _lib.stream()), "swat_linear_wgrad_bias")
    finally:
        lib.swat_gemm_set_tile(-1)
    close(dw, dy.float().t() @ x.float(), False, R)
    ref = dy.float().sum(0)
    assert (db - ref).abs().max().item() <= 1e-5 * math.sqrt(R) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(32, 200, 512), (32, 768, 512), (64, 1000, 768), (5, 37, 64), (1, 1, 1024), (33, 129, 192)])
def test_small_k_ordered_product_is_the_matrix_core_chain(M, N, K):
    """The head's two k-contiguous products (logits z = fn W^T, du = df proj^T: swat_sgemm_small, M <= 64, K % 64 == 0) run as one
    k-ordered fp32 fma chain per output on the vector ALU — bit for bit the f32-matrix-core product (swat_sim_topk_candidates, itself
    pinned to the C oracle's canonical order in tests/test_gpu_retrieval.py) that computed them until round 3."""
    from swat_amd import _lib
    L = _lib.lib()
    a = rnd(M, K, seed=71)
    b = rnd(N, K, seed=72)
    ldc = (N + 3) // 4 * 4
    c1 = torch.full((M, ldc), float("nan"), device=dev())
    c2 = torch.full((M, ldc), float("nan"), device=dev())
    _lib.check(L.swat_sgemm_small(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c1), M, N, K, K, 1, 1, K, ldc, 1.0, 0, _lib.stream()), "swat_sgemm_small")
    _lib.check(L.swat_sim_topk_candidates(_lib.ptr(a), M, _lib.ptr(b), N, K, None, 0, None, 0, None, _lib.ptr(c2), ldc, _lib.stream()),
               "swat_sim_topk_candidates")
    assert torch.equal(c1[:, :N], c2[:, :N])
    ref = a.double() @ b.double().t()
    assert (c1[:, :N].double() - ref).abs().max().item() <= 1e-4 * (K ** 0.5)
