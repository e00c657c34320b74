"""Data-parallel gradient reduction over contiguous slices of the flat gradient arena.

The tower is laid out backward-stage by stage, so "the gradients stage s produced" is ONE slice: the step issues an
asynchronous all-reduce (RCCL over xGMI through torch.distributed's `nccl` backend; gloo on CPU in the tests) per
stage right after that stage's kernels were enqueued, and waits for all of them before the optimizer.  The 1/world
average is folded into the AdamW kernel's grad_scale.
"""
import torch.distributed as dist


def stage_slices(arena, stages):
    """stages: list of parameter lists -> [(lo, hi)] element ranges (64-aligned, as the arena lays tensors out)."""
    out = []
    for ps in stages:
        lo = arena.offset_of(ps[0])
        hi = arena.offset_of(ps[-1]) + (ps[-1].numel() + 63) // 64 * 64
        out.append((lo, hi))
    return out


class GradWire:
    """How a contiguous gradient range crosses the links (SURVEY 8(e) row 1 / section 5 "Distributed communication backend"):

    * collective `all_reduce` (default) or `rs_ag`: reduce_scatter_tensor + all_gather_into_tensor — on 8 fully connected
      GPUs every rank then exchanges 1/8 of the range with each of its 7 peers over all 7 xGMI links at once instead of
      pushing the whole range round a ring (SWAT_DP_COLLECTIVE=rs_ag);
    * wire format `fp32` (this class's default; the CLI's --dp_wire auto picks fp32 for the batch-splitting mode and bf16 for --dp_weak,
      bench.py --gpus N picks bf16: swat_amd/dp.py) or `bf16` (SWAT_DP_WIRE=bf16): the range is rounded to bf16 (swat_cast_bf16), SUMMED in
      bf16 by the collective and widened back into the fp32 gradient buffer the optimizer reads — half the bytes per link.
      Tolerance: every addend rounded to 8 significant bits and every partial sum rounded again, i.e.
      |error| <= (1 + log2(world)) * 2^-8 * sum_r |g_r| per element (tests/test_distributed_cpu.py holds it to that);
      AdamW's fp32 moments average it over steps.

    Everything is issued on the caller's current stream (torch.cuda.stream(comm) in the step)."""

    def __init__(self, numel, device, group=None, wire=None, collective=None):
        import os
        import torch
        self.group = group
        self.wire = wire or os.environ.get("SWAT_DP_WIRE", "fp32")
        self.collective = collective or os.environ.get("SWAT_DP_COLLECTIVE", "all_reduce")
        if self.wire not in ("fp32", "bf16") or self.collective not in ("all_reduce", "rs_ag"):
            raise ValueError(f"SWAT_DP_WIRE={self.wire} / SWAT_DP_COLLECTIVE={self.collective}: expected fp32|bf16 and all_reduce|rs_ag")
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._wire_buf = torch.empty(numel, dtype=torch.bfloat16, device=device) if self.wire == "bf16" else None
        self._shard = None

    def reduce(self, flat_grad, lo, hi):
        import torch
        from . import ops
        g = flat_grad[lo:hi]
        t = g
        if self.wire == "bf16":
            t = self._wire_buf[lo:hi]
            if g.is_cuda:
                ops.cast_bf16(g, t)
            else:
                t.copy_(g)   # CPU tensors (gloo tests of the bookkeeping): torch's own round-to-nearest-even
        n = hi - lo
        if self.collective == "rs_ag" and n % self.world == 0:
            per = n // self.world
            if self._shard is None or self._shard.numel() < per or self._shard.dtype != t.dtype:
                self._shard = torch.empty(per, dtype=t.dtype, device=t.device)
            sh = self._shard[:per]
            dist.reduce_scatter_tensor(sh, t, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(t, sh, group=self.group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        if self.wire == "bf16":
            if g.is_cuda:
                ops.widen_bf16(t, g)
            else:
                g.copy_(t)


class SliceReducer:
    def __init__(self, group=None):
        self.group = group
        self.pending = []

    def reduce(self, flat_grad, lo, hi):
        self.pending.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
