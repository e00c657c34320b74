"""Flat parameter arena: every trainable tensor of a run lives in ONE fp32 buffer (+ a bf16 shadow the GEMMs
read, + optional grad / Adam-moment buffers at the same offsets), so the ViT composite addresses weights by
offset, the optimizer is one kernel launch and data-parallel gradient reduction works on contiguous slices.

nn.Parameters stay ordinary Parameters whose `.data` are views into the arena, so state_dict() / load_state_dict()
/ copy.deepcopy() / WiSE-FT interpolation (ref: main.py:76-98, utils/models.py:99,133, testing.py:88) keep working.
Anything that re-allocates a parameter (`.to()`, `.float()`, deepcopy) is detected through data_ptr() and the
arena is rebuilt lazily.
"""
import torch

from . import ops

ALIGN = 64  # elements; keeps every tensor 256-B aligned in fp32 and 128-B in bf16


class Arena:
    def __init__(self, params, device, with_grad=False, with_adam=False):
        """params: list of nn.Parameter (order = layout order)."""
        self.params = list(params)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.device = torch.device(device)
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device) if with_grad else None
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=self.device) if with_adam else None
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=self.device) if with_adam else None
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data.to(self.device, torch.float32))
                p.data = view
                if with_grad:
                    p.grad = None
        self._shadow_version = -1
        self.refresh_shadow()

    def grad_view(self, i):
        p, o = self.params[i], self.offsets[i]
        return self.grad[o:o + p.numel()].view(p.shape)

    def owns(self, p, i):
        return p.data_ptr() == self.flat.data_ptr() + 4 * self.offsets[i] and p.device == self.flat.device

    def valid(self):
        return all(self.owns(p, i) for i, p in enumerate(self.params))

    def offset_of(self, p):
        off = (p.data_ptr() - self.flat.data_ptr()) // 4
        if not (0 <= off < self.numel) or p.device != self.flat.device:
            raise RuntimeError("parameter is not inside this arena")
        return off

    def _version_key(self):
        """Parameters are attached with `p.data = view`, so each keeps a version counter of its own: an in-place torch
        write through a Parameter (load_state_dict, p.copy_, WiSE-FT interpolation) bumps p._version and NOT
        flat._version.  The key covers both routes; kernels that write through raw pointers (AdamW, cast) bump neither
        and call mark_shadow_fresh() themselves."""
        return (self.flat._version,) + tuple(p._version for p in self.params)

    def refresh_shadow(self, force=False):
        """bf16 shadow <- fp32 master when the master was modified through torch since the shadow was last written."""
        key = self._version_key()
        if force or key != self._shadow_version:
            ops.cast_bf16(self.flat, self.shadow)
            self._shadow_version = self._version_key()

    def mark_shadow_fresh(self):
        self._shadow_version = self._version_key()
