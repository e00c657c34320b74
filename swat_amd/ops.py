"""Path A device ops: thin tensor-level wrappers over the C ABI (include/swat_hip.h).

Every function launches hand-written gfx950 kernels on torch's current stream and allocates only
outputs; there is no PyTorch fallback.  bf16 tensors are torch.bfloat16, everything else fp32/int64.
"""
import torch

from . import _lib
from ._lib import ptr as _p

NT, NN, TN = 0, 1, 2
ACT = {None: 0, "none": 0, "quick_gelu": 1, "gelu": 2}
_BF, _F32 = torch.bfloat16, torch.float32


def _chk(rc, what):
    _lib.check(rc, what)


def gemm(layout, A, B, M, N, K, lda, ldb, C, ldc, batch=1, inner=1, sA=(0, 0), sB=(0, 0), sC=(0, 0),
         bias=None, residual=None, actin=None, ldact=0, act=None, C2=None, alpha=1.0):
    """Raw GEMM entry on explicit pointers/strides; A, B bf16 tensors (any shape), C bf16 or fp32 (None with a C2: only the
    activated output is stored)."""
    rc = _lib.lib().swat_gemm_bf16(layout, _p(A), _p(B), _p(C), M, N, K, lda, ldb, ldc, batch, inner,
                                   sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], _p(bias), _p(residual), _p(actin),
                                   ldact, ACT[act], _p(C2), int(C is not None and C.dtype == _F32), float(alpha), _lib.stream())
    _chk(rc, "swat_gemm_bf16")
    return C


def linear_fwd(x, w, bias=None, residual=None, act=None, out_dtype=_BF, want_act=False):
    """y = x @ w.T (+bias) (+residual); x [M,K] bf16, w [N,K] bf16.  want_act -> (pre, act(pre))."""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=out_dtype, device=x.device)
    y2 = torch.empty((M, N), dtype=_BF, device=x.device) if want_act else None
    gemm(NT, x, w, M, N, K, K, K, y, N, bias=bias, residual=residual, act=act, C2=y2)
    return (y, y2) if want_act else y


def linear_dgrad(dy, w, actin=None, act=None, out_dtype=_BF):
    """dx = dy @ w (* act'(actin)); dy [M,N] bf16, w [N,K] bf16."""
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty((M, K), dtype=out_dtype, device=dy.device)
    gemm(NN, dy, w, M, K, N, N, K, dx, K, actin=actin, ldact=K if actin is not None else 0, act=act)
    return dx


def linear_wgrad(dy, x, out=None):
    """dw = dy.T @ x  -> [N,K] fp32; dy [M,N] bf16, x [M,K] bf16."""
    M, N = dy.shape
    K = x.shape[1]
    if out is None:
        out = torch.empty((N, K), dtype=_F32, device=dy.device)
    gemm(TN, dy, x, N, K, M, N, K, out, K)
    return out


def layernorm_fwd(x, gamma, beta, eps=1e-5, rows=None, row_stride=None, out_bf16=True, out_f32=False):
    d = gamma.numel()
    if rows is None:
        rows = x.numel() // d
        row_stride = d
    dev = x.device
    y_bf = torch.empty((rows, d), dtype=_BF, device=dev) if out_bf16 else None
    y_f = torch.empty((rows, d), dtype=_F32, device=dev) if out_f32 else None
    mean = torch.empty(rows, dtype=_F32, device=dev)
    rstd = torch.empty(rows, dtype=_F32, device=dev)
    rc = _lib.lib().swat_layernorm_fwd(_p(x), row_stride, _p(gamma), _p(beta), _p(y_bf), _p(y_f), _p(mean), _p(rstd),
                                       rows, d, float(eps), _lib.stream())
    _chk(rc, "swat_layernorm_fwd")
    return y_bf, y_f, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dx_add=None, want_bf16=False, rows=None, dy_stride=None, x_stride=None,
                  dx_out=None, dxo_stride=None, dxa_stride=None, want_colsum=False):
    d = gamma.numel()
    if rows is None:
        rows = dy.numel() // d
    dy_stride = dy_stride or d
    x_stride = x_stride or d
    dev = dy.device
    if dx_out is None:
        dx_out = torch.empty((rows, d), dtype=_F32, device=dev)
        dxo_stride = d
    dx_bf = torch.empty((rows, d), dtype=_BF, device=dev) if want_bf16 else None
    dg = torch.empty(d, dtype=_F32, device=dev)
    db = torch.empty(d, dtype=_F32, device=dev)
    dcs = torch.empty(d, dtype=_F32, device=dev) if want_bf16 else None
    ws = torch.empty(_lib.lib().swat_layernorm_bwd_workspace_bytes(rows, d), dtype=torch.uint8, device=dev)
    rc = _lib.lib().swat_layernorm_bwd(_p(dy), dy_stride, _p(x), x_stride, _p(gamma), _p(mean), _p(rstd), _p(dx_add),
                                       dxa_stride or d, _p(dx_out), dxo_stride, _p(dx_bf), d, _p(dg), _p(db), _p(dcs), _p(ws),
                                       rows, d, _lib.stream())
    _chk(rc, "swat_layernorm_bwd")
    if want_colsum:
        return dx_out, dx_bf, dg, db, dcs
    return dx_out, dx_bf, dg, db


def softmax_fwd(S, Tq, T, scale, causal=False):
    """S fp32 [..., Tq, ld] -> P bf16 same shape (pad columns zero)."""
    ld = S.shape[-1]
    rows = S.numel() // ld
    P = torch.empty(S.shape, dtype=_BF, device=S.device)
    rc = _lib.lib().swat_softmax_fwd(_p(S), _p(P), rows, Tq, T, ld, float(scale), int(causal), _lib.stream())
    _chk(rc, "swat_softmax_fwd")
    return P


def softmax_bwd(P, dP, T, scale):
    ld = P.shape[-1]
    rows = P.numel() // ld
    dS = torch.empty(P.shape, dtype=_BF, device=P.device)
    rc = _lib.lib().swat_softmax_bwd(_p(P), _p(dP), _p(dS), rows, T, ld, float(scale), _lib.stream())
    _chk(rc, "swat_softmax_bwd")
    return dS


def patchify_cutmix(images, patch, perm=None, box=None, ldp=None):
    """images fp32 [B,3,H,W] -> patches bf16 [B*g*g, ldp]; box = (bbx1, bby1, bbx2, bby2) as rand_bbox returns."""
    B, _, H, W = images.shape
    K = 3 * patch * patch
    ldp = ldp or (K + 7) // 8 * 8
    out = torch.empty((B * (H // patch) * (W // patch), ldp), dtype=_BF, device=images.device)
    x1 = x2 = y1 = y2 = 0
    if perm is not None and box is not None:
        x1, y1, x2, y2 = [int(v) for v in box]
    rc = _lib.lib().swat_patchify_cutmix(_p(images), _p(perm) if perm is not None else None, _p(out), B, H, W, patch,
                                         ldp, x1, x2, y1, y2, _lib.stream())
    _chk(rc, "swat_patchify_cutmix")
    return out


def cutmix_paste(images, perm, box):
    """Out-of-place images[:, :, x1:x2, y1:y2] = images[perm, :, x1:x2, y1:y2] (ref utils/training.py:1466)."""
    B, _, H, W = images.shape
    x1, y1, x2, y2 = [int(v) for v in box]
    out = torch.empty_like(images)
    rc = _lib.lib().swat_cutmix_paste(_p(images), _p(perm), _p(out), B, H, W, x1, x2, y1, y2, _lib.stream())
    _chk(rc, "swat_cutmix_paste")
    return out


def colsum(X, rows=None, n=None, ld=None, group=(0, 0, 0)):
    if rows is None:
        rows, n = X.shape
        ld = n
    out = torch.empty(n, dtype=_F32, device=X.device)
    rc = _lib.lib().swat_colsum(_p(X), 1 if X.dtype == _BF else 0, rows, n, ld, _p(out), group[0], group[1], group[2],
                                _lib.stream())
    _chk(rc, "swat_colsum")
    return out


def cast_bf16(x, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=_BF, device=x.device)
    rc = _lib.lib().swat_cast_bf16(_p(x), _p(out), x.numel(), _lib.stream())
    _chk(rc, "swat_cast_bf16")
    return out


def widen_bf16(x, out):
    """bf16 -> fp32 into `out` (same number of elements, a multiple of 4)."""
    rc = _lib.lib().swat_widen_bf16(_p(x), _p(out), x.numel(), _lib.stream())
    _chk(rc, "swat_widen_bf16")
    return out


def l2norm_fwd(x):
    rows, d = x.shape
    y = torch.empty_like(x)
    nrm = torch.empty(rows, dtype=_F32, device=x.device)
    _chk(_lib.lib().swat_l2norm_fwd(_p(x), _p(y), _p(nrm), rows, d, _lib.stream()), "swat_l2norm_fwd")
    return y, nrm


def l2norm_bwd(y, nrm, dy):
    rows, d = y.shape
    dx = torch.empty_like(y)
    _chk(_lib.lib().swat_l2norm_bwd(_p(y), _p(nrm), _p(dy), _p(dx), rows, d, _lib.stream()), "swat_l2norm_bwd")
    return dx


def sgemm_small(A, B, M, N, K, sam, sak, sbk, sbn, out=None, alpha=1.0, accumulate=False):
    if out is None:
        out = torch.empty((M, N), dtype=_F32, device=A.device)
    rc = _lib.lib().swat_sgemm_small(_p(A), _p(B), _p(out), M, N, K, sam, sak, sbk, sbn, out.shape[-1], float(alpha),
                                     int(accumulate), _lib.stream())
    _chk(rc, "swat_sgemm_small")
    return out


def ce_mixed(z, logit_scale, ya, yb, lam, want_grad=True, want_logits=False, loss_accum=None):
    """loss / dz / dlogit_scale of lam*CE(z*e^ls, ya) + (1-lam)*CE(z*e^ls, yb) (mean over the batch)."""
    B, C = z.shape
    dev = z.device
    ws = torch.empty(2 * B, dtype=_F32, device=dev)
    loss = torch.empty(1, dtype=_F32, device=dev)
    dls = torch.empty(1, dtype=_F32, device=dev)
    dz = torch.empty_like(z) if want_grad else None
    logits = torch.empty_like(z) if want_logits else None
    rc = _lib.lib().swat_ce_mixed(_p(z), _p(logit_scale), _p(ya), _p(yb), float(lam), _p(ws), _p(loss), _p(dls), _p(dz),
                                  _p(logits), _p(loss_accum), B, C, _lib.stream())
    _chk(rc, "swat_ce_mixed")
    return loss, dz, dls, logits


def adamw_step(params, grads, exp_avg, exp_avg_sq, shadow, seg_begin, lrs_dev, bias_corr_dev, beta1=0.9, beta2=0.999,
               eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
    import ctypes
    n = params.numel()
    arr = (ctypes.c_int64 * len(seg_begin))(*seg_begin)
    rc = _lib.lib().swat_adamw(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), _p(shadow), n,
                               ctypes.cast(arr, ctypes.c_void_p), len(seg_begin), _p(lrs_dev), float(beta1), float(beta2),
                               float(eps), float(weight_decay), _p(bias_corr_dev), float(grad_scale), _lib.stream())
    _chk(rc, "swat_adamw")
