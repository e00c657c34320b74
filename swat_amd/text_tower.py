"""CLIP text tower on the gfx950 kernels: token-embedding gather + positional add -> L causal pre-LN
blocks (the vision tower's own GEMM / LayerNorm / softmax kernels) -> ln_final at the EOT token -> @ text_projection.

ref: model.encode_text — utils/features.py:27-36, retrieval/extract_mined_feature.py:113-125; architecture SURVEY App. A.
Forward only: SWAT always freezes the text tower (utils/optimizers.py:12-19).
"""
import ctypes

import torch

from . import _lib, ops
from .arena import Arena

_ACT = {"quick_gelu": 1, "gelu": 2}


def _text_arena(model):
    params = [model.token_embedding.weight, model.positional_embedding, model.ln_final.weight, model.ln_final.bias,
              model.text_projection]
    for blk in model.transformer.resblocks:
        params += blk.ordered()
    a = getattr(model, "_text_arena", None)
    ok = a is not None
    if ok:
        base = a.flat.data_ptr()
        ok = all(p.is_cuda and 0 <= p.data_ptr() - base < 4 * a.numel for p in params)
    if not ok:
        if not params[0].is_cuda:
            raise _lib.SwatHipError("swat_amd text tower needs its weights on the GPU (no CPU fallback)")
        a = Arena(params, params[0].device)
        object.__setattr__(model, "_text_arena", a)
    return a


@torch.no_grad()
def encode_text(model, text):
    """text int64 [B, ctx] -> [B, E] fp32 (un-normalised, like open_clip)."""
    if not text.is_cuda:
        raise _lib.SwatHipError("swat_amd text tower needs GPU tensors (no CPU fallback)")
    text = text.contiguous().long()
    B, T = text.shape
    d = model.positional_embedding.shape[1]
    L = len(model.transformer.resblocks)
    H = model.text_heads
    mlp = model.transformer.resblocks[0].mlp.c_fc.weight.shape[0]
    a = _text_arena(model)
    a.refresh_shadow()
    dev = text.device
    x = torch.empty((B * T, d), dtype=torch.float32, device=dev)
    lib = _lib.lib()
    rc = lib.swat_embed_tokens(_lib.ptr(text), _lib.ptr(model.token_embedding.weight), _lib.ptr(model.positional_embedding),
                               _lib.ptr(x), B * T, T, d, _lib.stream())
    _lib.check(rc, "swat_embed_tokens")
    dims = (ctypes.c_int32 * 8)(B, T, d, L, H, mlp, _ACT[model.text_act], 1)
    o = []
    for blk in model.transformer.resblocks:
        o += [a.offset_of(p) for p in blk.ordered()]
    offs = (ctypes.c_int64 * len(o))(*o)
    ws = torch.empty(lib.swat_transformer_workspace_bytes(dims), dtype=torch.uint8, device=dev)
    rc = lib.swat_transformer_forward(dims, offs, _lib.ptr(a.flat), _lib.ptr(a.shadow), _lib.ptr(x), _lib.ptr(ws), _lib.stream())
    _lib.check(rc, "swat_transformer_forward")
    eot = text.argmax(dim=-1)                      # EOT has the highest id (49407)
    rows = x.view(B, T, d)[torch.arange(B, device=dev), eot].contiguous()
    _, y, _, _ = ops.layernorm_fwd(rows, model.ln_final.weight, model.ln_final.bias, out_bf16=False, out_f32=True)
    E = model.text_projection.shape[1]
    return ops.sgemm_small(y, model.text_projection, B, E, d, d, 1, E, 1)
