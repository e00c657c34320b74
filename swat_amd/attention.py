"""Multi-head attention of the ViT block as batched MFMA GEMMs + a row-softmax kernel.

ref: nn.MultiheadAttention inside open_clip's ResidualAttentionBlock (packed in_proj, scores scaled by
1/sqrt(64), softmax over keys, no mask/dropout for the vision tower; causal for the text tower).
T <= 257 here, so the [T,T] score tile per (batch, head) is materialised (fp32 scores, bf16 probabilities,
leading dimension padded to a multiple of 8 so every GEMM operand row is 16-B aligned); the probabilities
are kept for the backward pass.  The C++ composite (csrc/vit.hip) issues exactly this sequence.
"""
import math

import torch

from . import ops

_BF, _F32 = torch.bfloat16, torch.float32


def _ldp(T):
    return (T + 7) // 8 * 8


def attn_fwd(qkv, B, T, H, dh, causal=False):
    """qkv bf16 [B*T, 3*H*dh] (q|k|v) -> (out bf16 [B*T, H*dh], P bf16 [B*H, T, ldp])."""
    d = H * dh
    ldp = _ldp(T)
    dev = qkv.device
    S = torch.empty((B * H, T, ldp), dtype=_F32, device=dev)
    q, k, v = qkv, qkv[:, d:], qkv[:, 2 * d:]
    ops.gemm(ops.NT, q, _view(k), T, T, dh, 3 * d, 3 * d, S, ldp, batch=B * H, inner=H,
             sA=(T * 3 * d, dh), sB=(T * 3 * d, dh), sC=(H * T * ldp, T * ldp))
    P = ops.softmax_fwd(S, T, T, 1.0 / math.sqrt(dh), causal=causal)
    out = torch.empty((B * T, d), dtype=_BF, device=dev)
    ops.gemm(ops.NN, P, _view(v), T, dh, T, ldp, 3 * d, out, d, batch=B * H, inner=H,
             sA=(H * T * ldp, T * ldp), sB=(T * 3 * d, dh), sC=(T * d, dh))
    return out, P


def attn_bwd(qkv, P, dout, B, T, H, dh):
    """-> dqkv bf16 [B*T, 3*H*dh]."""
    d = H * dh
    ldp = P.shape[-1]
    dev = qkv.device
    dqkv = torch.empty_like(qkv)
    q, k, v = qkv, _view(qkv[:, d:]), _view(qkv[:, 2 * d:])
    dq, dk, dv = dqkv, _view(dqkv[:, d:]), _view(dqkv[:, 2 * d:])
    sP = (H * T * ldp, T * ldp)
    sQ = (T * 3 * d, dh)
    sO = (T * d, dh)
    # dV = P^T dO
    ops.gemm(ops.TN, P, dout, T, dh, T, ldp, d, dv, 3 * d, batch=B * H, inner=H, sA=sP, sB=sO, sC=sQ)
    # dP = dO V^T
    dP = torch.empty((B * H, T, ldp), dtype=_F32, device=dev)
    ops.gemm(ops.NT, dout, v, T, T, dh, d, 3 * d, dP, ldp, batch=B * H, inner=H, sA=sO, sB=sQ, sC=sP)
    dS = ops.softmax_bwd(P, dP, T, 1.0 / math.sqrt(dh))
    # dQ = dS K ; dK = dS^T Q
    ops.gemm(ops.NN, dS, k, T, dh, T, ldp, 3 * d, dq, 3 * d, batch=B * H, inner=H, sA=sP, sB=sQ, sC=sQ)
    ops.gemm(ops.TN, dS, q, T, dh, T, ldp, 3 * d, dk, 3 * d, batch=B * H, inner=H, sA=sP, sB=sQ, sC=sQ)
    return dqkv


class _view:
    """A column-offset view of a contiguous 2-D tensor: only its data pointer is used."""

    def __init__(self, t):
        self.t = t
        self.is_cuda = t.is_cuda
        self.dtype = t.dtype

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return self.t.data_ptr()
