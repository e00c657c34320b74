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
