"""The `model` object SWAT's loops drive: open_clip's CLIP surface (ref: utils/extras.py:138-171 returns it;
SURVEY §8(b)) with the vision tower executed by the gfx950 composite in csrc/vit.hip.

  model.encode_image(x) -> [B,E] fp32        (autograd-capable: one Function node for the whole tower)
  model.visual                               (nn.Module; .parameters() is the backbone LR group, .proj exists)
  model.transformer / token_embedding / positional_embedding / ln_final / text_projection / logit_scale
  state_dict() / load_state_dict() with open_clip key names; survives copy.deepcopy, .float(), .to()

The tower has no CPU / PyTorch fallback: calling it with CPU tensors or without libswat_hip.so raises.
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _lib
from .arena import Arena

# --model_cfg -> (open_clip arch, pretrained tag) (ref: utils/extras.py:97-114) and arch -> vision config
# (image, patch, width, layers, heads, embed, act) / text config (ctx, vocab, width, heads, layers)  
VISION_CFGS = {
    "ViT-B-32-quickgelu": (224, 32, 768, 12, 12, 512, "quick_gelu"),
    "ViT-B-32": (224, 32, 768, 12, 12, 512, "gelu"),
    "ViT-B-16": (224, 16, 768, 12, 12, 512, "gelu"),
    "ViT-L-14": (224, 14, 1024, 24, 16, 768, "gelu"),
}
TEXT_CFGS = {
    "ViT-B-32-quickgelu": (77, 49408, 512, 8, 12),
    "ViT-B-32": (77, 49408, 512, 8, 12),
    "ViT-B-16": (77, 49408, 512, 8, 12),
    "ViT-L-14": (77, 49408, 768, 12, 12),
}
_ACT = {"quick_gelu": 1, "gelu": 2}


class _Params(nn.Module):
    """A bag of named Parameters (stands in for nn.Linear / nn.LayerNorm / nn.Conv2d: same state-dict keys)."""

    def __init__(self, **shapes):
        super().__init__()
        for k, shp in shapes.items():
            self.register_parameter(k, nn.Parameter(torch.empty(shp)))


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = _Params(weight=(d, d), bias=(d,))


class _Mlp(nn.Module):
    def __init__(self, d, m):
        super().__init__()
        self.c_fc = _Params(weight=(m, d), bias=(m,))
        self.c_proj = _Params(weight=(d, m), bias=(d,))


class _Block(nn.Module):
    def __init__(self, d, m):
        super().__init__()
        self.ln_1 = _Params(weight=(d,), bias=(d,))
        self.attn = _Attn(d)
        self.ln_2 = _Params(weight=(d,), bias=(d,))
        self.mlp = _Mlp(d, m)

    def ordered(self):
        return [self.ln_1.weight, self.ln_1.bias, self.attn.in_proj_weight, self.attn.in_proj_bias,
                self.attn.out_proj.weight, self.attn.out_proj.bias, self.ln_2.weight, self.ln_2.bias,
                self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias]


class _Transformer(nn.Module):
    def __init__(self, d, m, L):
        super().__init__()
        self.resblocks = nn.ModuleList([_Block(d, m) for _ in range(L)])


def _init_block(blk, d, m):
    """open_clip / torch defaults: xavier in_proj, zero MHA biases, kaiming-uniform(a=sqrt 5) Linear, LN 1/0."""
    nn.init.xavier_uniform_(blk.attn.in_proj_weight)
    nn.init.zeros_(blk.attn.in_proj_bias)
    for lin, fan_in in ((blk.attn.out_proj, d), (blk.mlp.c_fc, d), (blk.mlp.c_proj, m)):
        nn.init.kaiming_uniform_(lin.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(fan_in)
        nn.init.uniform_(lin.bias, -bound, bound)
    nn.init.zeros_(blk.attn.out_proj.bias)
    for ln in (blk.ln_1, blk.ln_2):
        nn.init.ones_(ln.weight)
        nn.init.zeros_(ln.bias)


class _EncodeImageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, tower, *params):
        ctx.tower = tower
        ctx.n_params = len(params)
        need = any(ctx.needs_input_grad[2:])  # grad mode is off inside forward(); this is the reliable signal
        ctx.saved = need
        return tower.forward_raw(images, save=need)

    @staticmethod
    def backward(ctx, dfeats):
        tower = ctx.tower
        if not ctx.saved:
            raise RuntimeError("encode_image ran without saving activations")
        tower.backward_raw(dfeats.contiguous())
        arena = tower.arena()
        grads = []
        for p in tower.ordered_params():
            o = arena.offset_of(p)
            grads.append(arena.grad[o:o + p.numel()].view(p.shape) if p.requires_grad else None)
        return (None, None, *grads)


class VisionTransformer(nn.Module):
    """open_clip VisionTransformer parameters + the HIP composite that runs them."""

    def __init__(self, image_size=224, patch=32, width=768, layers=12, heads=12, embed=512, act="quick_gelu", mlp_ratio=4):
        super().__init__()
        self.image_size, self.patch_size, self.width, self.layers, self.heads = image_size, patch, width, layers, heads
        self.output_dim, self.act, self.mlp_width = embed, act, width * mlp_ratio
        g = image_size // patch
        self.grid, self.tokens = g, g * g + 1
        scale = width ** -0.5
        self.conv1 = _Params(weight=(width, 3, patch, patch))
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.tokens, width))
        self.ln_pre = _Params(weight=(width,), bias=(width,))
        self.transformer = _Transformer(width, self.mlp_width, layers)
        self.ln_post = _Params(weight=(width,), bias=(width,))
        self.proj = nn.Parameter(scale * torch.randn(width, embed))
        nn.init.kaiming_uniform_(self.conv1.weight, a=math.sqrt(5))
        for ln in (self.ln_pre, self.ln_post):
            nn.init.ones_(ln.weight)
            nn.init.zeros_(ln.bias)
        for blk in self.transformer.resblocks:
            _init_block(blk, width, self.mlp_width)
        self._arena = None
        self._ws = {}
        self._last_ws = None

    # ---- parameter layout -------------------------------------------------------------------
    def stage_params(self, stage):
        """Parameters touched by backward stage `stage` (0 stem, l+1 block l, L+1 ln_post+proj)."""
        if stage == 0:
            return [self.conv1.weight, self.class_embedding, self.positional_embedding, self.ln_pre.weight, self.ln_pre.bias]
        if stage == self.layers + 1:
            return [self.ln_post.weight, self.ln_post.bias, self.proj]
        return self.transformer.resblocks[stage - 1].ordered()

    def ordered_params(self):
        out = []
        for s in range(self.layers + 2):
            out += self.stage_params(s)
        return out

    def arena(self):
        """The arena holding this tower's weights (a private one unless an optimizer arena adopted them)."""
        params = self.ordered_params()
        a = self._arena
        ok = a is not None
        if ok:
            try:
                base = a.flat.data_ptr()
                ok = all(p.is_cuda and 0 <= p.data_ptr() - base < 4 * a.numel for p in params)
            except RuntimeError:
                ok = False
        if not ok:
            dev = params[0].device
            if dev.type != "cuda":
                raise _lib.SwatHipError("swat_amd vision tower needs its weights on the GPU (no CPU fallback): "
                                        "call model.to('cuda')")
            self._arena = a = Arena(params, dev)
            self._ws = {}
        return a

    def adopt_arena(self, arena):
        self._arena = arena
        self._ws = {}

    def _tables(self, B, save):
        # one pass over the parameters' addresses both validates the arena and keys the cached tables
        key = (B, bool(save), tuple(p.data_ptr() for p in self.ordered_params()))
        cached = self.__dict__.get("_tab")
        if cached is not None and cached[0] == key and cached[1][2] is self._arena:
            return cached[1]
        dims = (ctypes.c_int32 * 10)(B, self.image_size, self.patch_size, self.width, self.layers, self.heads, self.mlp_width,
                                     self.output_dim, _ACT[self.act], int(save))
        a = self.arena()
        o = [a.offset_of(p) for p in (self.conv1.weight, self.class_embedding, self.positional_embedding, self.ln_pre.weight,
                                      self.ln_pre.bias, self.ln_post.weight, self.ln_post.bias, self.proj)]
        for blk in self.transformer.resblocks:
            o += [a.offset_of(p) for p in blk.ordered()]
        offs = (ctypes.c_int64 * len(o))(*o)
        self._tab = (key, (dims, offs, a))
        return dims, offs, a

    def _workspace(self, B, save, dims, dev):
        key = (B, bool(save))
        ws = self._ws.get(key)
        if ws is None or ws.device != dev:
            nbytes = _lib.lib().swat_vit_workspace_bytes(dims)
            if nbytes < 0:
                raise _lib.SwatHipError("swat_vit_workspace_bytes: unsupported model dimensions")
            ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)  # zero once: the row padding of the wgrad operands must read as 0
            self._ws[key] = ws
        return ws

    # ---- raw passes (no autograd) -----------------------------------------------------------
    def forward_raw(self, images, save=False, perm=None, box=None, batch=None):
        """encode_image on the current stream.  perm/box: fuse the CutMix paste of ref utils/training.py:1466 into the
        patch extraction (box = (bbx1, bby1, bbx2, bby2) as rand_bbox returns it).  batch: number of samples when
        `images` holds more rows than that (rows >= batch are only read through perm: a data-parallel rank's CutMix
        partners, swat_amd/dp.py)."""
        if not images.is_cuda:
            raise _lib.SwatHipError("swat_amd vision tower needs GPU tensors (no CPU fallback)")
        images = images.contiguous().float()
        B = images.shape[0] if batch is None else int(batch)
        assert B <= images.shape[0]
        assert images.shape[1:] == (3, self.image_size, self.image_size), images.shape
        dims, offs, a = self._tables(B, save)
        a.refresh_shadow()
        ws = self._workspace(B, save, dims, images.device)
        feats = torch.empty((B, self.output_dim), dtype=torch.float32, device=images.device)
        x1 = x2 = y1 = y2 = 0
        if perm is not None and box is not None:
            x1, y1, x2, y2 = [int(v) for v in box]
        rc = _lib.lib().swat_vit_forward(dims, offs, _lib.ptr(a.flat), _lib.ptr(a.shadow), _lib.ptr(images),
                                         _lib.ptr(perm) if perm is not None else None, x1, x2, y1, y2, _lib.ptr(ws),
                                         _lib.ptr(feats), _lib.stream())
        _lib.check(rc, "swat_vit_forward")
        if save:
            self._last_ws = (B, dims, offs, a)
        return feats

    def backward_raw(self, dfeats, stage_hi=None, stage_lo=0):
        """Backward of the last forward_raw(save=True); gradients are written into arena.grad."""
        B, dims, offs, a = self._last_ws
        if a.grad is None:
            a.grad = torch.zeros_like(a.flat)
        ws = self._ws[(B, True)]
        hi = self.layers + 1 if stage_hi is None else stage_hi
        rc = _lib.lib().swat_vit_backward(dims, offs, _lib.ptr(a.flat), _lib.ptr(a.shadow), _lib.ptr(ws), _lib.ptr(dfeats),
                                          _lib.ptr(a.grad), hi, stage_lo, _lib.stream())
        _lib.check(rc, "swat_vit_backward")

    def forward(self, images):
        params = self.ordered_params()
        return _EncodeImageFn.apply(images, self, *params)

    def __deepcopy__(self, memo):
        # the arena / workspaces are caches: the copy rebuilds its own (ref: copy.deepcopy(model), main.py:555)
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        import copy
        for k, v in self.__dict__.items():
            if k in ("_arena", "_ws", "_last_ws", "_tab"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._arena, new._ws, new._last_ws = None, {}, None
        with torch.no_grad():
            for p in new.parameters():
                p.data = p.data.clone()
        return new


class CLIP(nn.Module):
    """open_clip.CLIP-shaped container: `visual` + the text-tower attributes SWAT touches
    (ref: utils/optimizers.py:12-19 lock_text_tower; utils/features.py:27-36 encode_text)."""

    def __init__(self, arch="ViT-B-32-quickgelu", vision_cfg=None, text_cfg=None):
        super().__init__()
        v = vision_cfg or VISION_CFGS[arch]
        ctx, vocab, tw, th, tl = text_cfg or TEXT_CFGS.get(arch, TEXT_CFGS["ViT-B-32"])
        self.arch = arch
        self.visual = VisionTransformer(*v)
        self.context_length, self.vocab_size, self.text_heads, self.text_act = ctx, vocab, th, v[6]
        self.transformer = _Transformer(tw, 4 * tw, tl)
        self.token_embedding = _Params(weight=(vocab, tw))
        self.positional_embedding = nn.Parameter(0.01 * torch.randn(ctx, tw))
        self.ln_final = _Params(weight=(tw,), bias=(tw,))
        self.text_projection = nn.Parameter(tw ** -0.5 * torch.randn(tw, v[5]))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.ones_(self.ln_final.weight)
        nn.init.zeros_(self.ln_final.bias)
        for blk in self.transformer.resblocks:
            _init_block(blk, tw, 4 * tw)

    def __deepcopy__(self, memo):
        # the text tower's arena is a cache (swat_amd/text_tower.py keeps it in __dict__): the copy rebuilds its own on
        # first use instead of dragging a second fp32 + bf16 image of the tower along (ref: copy.deepcopy(model) per
        # improving epoch, utils/training.py:1534; zeroshot_model / WiSE-FT models, main.py:76-98,555)
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_text_arena":
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        with torch.no_grad():
            for n_, p in new.named_parameters():
                if not n_.startswith("visual."):  # the vision tower re-owned its storage in its own __deepcopy__
                    p.data = p.data.clone()
        return new

    def encode_image(self, images):
        return self.visual(images)

    def encode_text(self, text):
        from .text_tower import encode_text
        return encode_text(self, text)

    def forward(self, images, text):
        i = self.encode_image(images)
        t = self.encode_text(text)
        i = i / i.norm(dim=-1, keepdim=True)
        t = t / t.norm(dim=-1, keepdim=True)
        return self.logit_scale.exp() * i @ t.t(), self.logit_scale.exp() * t @ i.t()
