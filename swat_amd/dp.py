"""Data-parallel finetuning from the command line: one process per GPU under `torch.distributed.run`,
`torch.distributed` over RCCL (backend "nccl" on ROCm; gloo for the one-GPU tests), model / optimizer / head replicated.

The reference trains on one GPU (utils/training.py:1452-1489, loaders utils/dataloader.py:107-135).  Two ways to spread it:

* global batch (default): the job's batch stays `--bsz`; every rank draws the SAME shuffles and CutMix decisions (same seed, RNG
  states re-synchronised from rank 0 at the start of every training loop) and works on rows [rank*b, (rank+1)*b) of each batch.
  Its loader yields those rows and, in a CutMix iteration, the b partner rows they are pasted from — nothing else is read or
  decoded (RankBatchSampler below; `--dp_full_load` restores the round-3 form in which every rank loaded the whole batch and
  sliced it, shard_batch).  The CutMix scalars (r, lambda, box) and the permutation are the single-process draws.  Per-rank mean
  losses averaged over ranks = the whole-batch mean, so the optimizer trajectory is the single-process one up to the summation
  order of the gradient reduction.
* `--dp_weak`: every rank loads its own `--bsz` rows per step (global batch = world x bsz): the sampler deals the
  shuffled index list out by rank (same permutation on every rank), CutMix permutes inside the rank's batch.

Validation, logging, checkpoints, WiSE-FT / tau-normalisation and stage-2 probing run on rank 0 only; the metrics that
steer the loop (val / test accuracy) are broadcast so every rank takes the same branches.  The per-stage gradient
all-reduce itself lives in swat_amd/step.py (FusedTrainStep, trailing mode)."""
import os
import pickle
import random

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import Sampler


def init_from_env(args, split_batch=True):
    """Call FIRST in main(): creates the process group before anything touches the GPU (a launcher must not start other
    programs from a process that has initialised the device, and RCCL wants its device bound at creation).
    split_batch=False: the caller does not split --bsz over the ranks (feature extraction), so any world size goes."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    args.rank, args.world, args.local_rank = 0, 1, 0
    if world <= 1:
        return args
    rank, local = int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    ngpu = torch.cuda.device_count()           # counting devices does not initialise the GPU
    index = local % max(ngpu, 1)               # several ranks per GPU only in the one-GPU tests (gloo)
    backend = getattr(args, 'dist_backend', 'nccl')
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', index))
    else:
        dist.init_process_group(backend)
    if ngpu:
        torch.cuda.set_device(index)
    args.rank, args.world, args.local_rank = rank, world, local
    # what crosses the links (ddp.GradWire reads SWAT_DP_WIRE): an explicit --dp_wire fp32 | bf16 wins; with `auto` an exported
    # SWAT_DP_WIRE wins over the mode's default
    wire = getattr(args, 'dp_wire', 'auto')
    if wire == 'auto':
        os.environ.setdefault('SWAT_DP_WIRE', 'bf16' if getattr(args, 'dp_weak', False) else 'fp32')
    else:
        env = os.environ.get('SWAT_DP_WIRE')
        if env and env != wire and rank == 0:
            import warnings
            warnings.warn(f"--dp_wire {wire} overrides the exported SWAT_DP_WIRE={env}")
        os.environ['SWAT_DP_WIRE'] = wire
    if split_batch and not getattr(args, 'dp_weak', False) and getattr(args, 'bsz', world) % world:
        raise ValueError(f"--bsz {args.bsz} is split over {world} ranks: it must be a multiple of the world size "
                         "(or use --dp_weak for --bsz rows per rank)")
    per = getattr(args, 'bsz', 0) // world
    if split_batch and not getattr(args, 'dp_weak', False) and 0 < per < 16 and rank == 0:
        import warnings
        warnings.warn(f"--bsz {args.bsz} over {world} ranks leaves {per} images per GPU and step: this mode reproduces the 1-GPU "
                      f"trajectory of the reference's batch size; a step that small is launch-bound on every rank.  Use --dp_weak "
                      f"(every rank its own --bsz images, global batch {world} x {args.bsz}) to scale throughput.")
    return args


def world(args):
    return getattr(args, 'world', 1)


def is_main(args):
    return getattr(args, 'rank', 0) == 0


def barrier(args):
    if world(args) > 1:
        dist.barrier()


def bcast_object(args, obj):
    """rank 0's (picklable) object on every rank."""
    if world(args) == 1:
        return obj
    box = [obj if is_main(args) else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def sync_rng(args):
    """Every rank continues from rank 0's torch / numpy / python RNG states: the loaders' shuffles and the CutMix draws
    stay identical across ranks even after rank 0 alone ran RNG-consuming code (stage-2 probing's shuffled extraction)."""
    if world(args) == 1:
        return
    state = (torch.get_rng_state(), np.random.get_state(), random.getstate(),
             torch.cuda.get_rng_state() if torch.cuda.is_available() else None)
    state = bcast_object(args, state)
    if not is_main(args):
        torch.set_rng_state(state[0])
        np.random.set_state(state[1])
        random.setstate(state[2])
        if state[3] is not None:
            torch.cuda.set_rng_state(state[3])


def mean_over_ranks(args, value):
    """Average of a host scalar over ranks (the epoch's training loss for loss.csv: one collective per epoch)."""
    if world(args) == 1:
        return value
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item() / world(args)


def shard_batch(images, labels, perm, rank, nranks):
    """Rows [rank*b, (rank+1)*b) of a global batch -> (images_local, y_a, y_b, perm_local, b).

    With a CutMix permutation over the GLOBAL batch the partners' images are appended behind the rank's own rows and
    perm_local points at them, so the fused paste (csrc: swat_patchify_cutmix) reads exactly the pixels the
    single-process step reads; `b` is the number of samples of the step (images_local may hold 2b rows)."""
    B = images.shape[0]
    if B % nranks:
        raise ValueError(f"global batch {B} is not a multiple of the world size {nranks}")
    b = B // nranks
    lo, hi = rank * b, (rank + 1) * b
    if perm is None:
        return images[lo:hi].contiguous(), labels[lo:hi], labels[lo:hi], None, b
    partner = perm[lo:hi]
    local = torch.cat([images[lo:hi], images.index_select(0, partner.to(images.device))]).contiguous()
    perm_local = torch.arange(b, 2 * b, device=perm.device, dtype=perm.dtype)
    return local, labels[lo:hi], labels.index_select(0, partner.to(labels.device)), perm_local, b


class RankStridedSampler(Sampler):
    """--dp_weak: the reference's shuffle (torch.randperm from the global RNG, drawn on EVERY rank so the streams stay in
    step), dealt out by rank: rank r gets elements r, r + W, r + 2W, ... of the permutation (a partition of the epoch)."""

    def __init__(self, n, rank, nranks, shuffle=True):
        self.n, self.rank, self.nranks, self.shuffle = n, rank, nranks, shuffle

    def __len__(self):
        return self.n // self.nranks

    def __iter__(self):
        order = torch.randperm(self.n).tolist() if self.shuffle else list(range(self.n))
        return iter(order[self.rank:self.n // self.nranks * self.nranks:self.nranks])


class RankBatchSampler(Sampler):
    """Global-batch mode without W-fold loading: rank r loads ONLY rows [r b, (r + 1) b) of every global batch, plus — in a CutMix
    iteration — the b partner rows its own rows are pasted from.  (Round 3: every rank iterated the whole loader and decoded all
    W b images of every batch to keep the single-process trajectory; at 8 ranks that is 8x the decode work for the same job.)

    How the single-process trajectory survives.  The batch composition is the reference loader's: `RandomSampler`'s permutation
    (drawn here exactly as torch draws it: a seed from the global generator, randperm on a private generator), cut into batches of
    `bsz`, tail dropped.  The CutMix decisions of iteration i (np.random.rand, np.random.beta, torch.randperm(bsz), rand_bbox's two
    np.random.randint: ref utils/training.py:1458-1472) are drawn HERE, when batch i's indices are handed to the DataLoader, in the
    same per-stream order as the training loop draws them in a single process, and queued for the loop (`pop_draw`).  The draws thus
    happen earlier in wall time, not in another order — the streams are the single-process streams as long as nothing else in the MAIN
    process draws from torch's / numpy's global generators during an epoch: loader workers (the reference's default, 8) have their own
    generators, `--synthetic` items are seeded per item.  With `--num_workers 0` on real files the dataset's crop / flip draws share
    torch's global generator with randperm(bsz): the job is then still a correct run of the same recipe, but not stream-identical to a
    1-process run.
    Yields index lists of length b (no CutMix in this iteration) or 2 b (own rows, then partner rows)."""

    def __init__(self, n, bsz, rank, nranks, img_hw=(224, 224)):
        if bsz % nranks:
            raise ValueError(f"--bsz {bsz} must be a multiple of the world size {nranks}")
        self.n, self.bsz, self.rank, self.nranks, self.img_hw = n, bsz, rank, nranks, img_hw
        self.b = bsz // nranks
        self.draw_fn = None          # training.cutmix_draw bound to args (set by the loop: configure)
        self.draws = []

    def configure(self, draw_fn):
        """draw_fn(size) -> (lam, rand_index | None, box | None): the loop's per-iteration CutMix draws (None: no CutMix at all)"""
        self.draw_fn = draw_fn
        self.draws = []

    def pop_draw(self):
        return self.draws.pop(0)

    def __len__(self):
        return self.n // self.bsz

    def __iter__(self):
        seed = int(torch.empty((), dtype=torch.int64).random_().item())      # torch.utils.data.RandomSampler.__iter__, verbatim order
        g = torch.Generator()
        g.manual_seed(seed)
        order = torch.randperm(self.n, generator=g).tolist()
        lo, hi = self.rank * self.b, (self.rank + 1) * self.b
        for i in range(self.n // self.bsz):
            batch = order[i * self.bsz:(i + 1) * self.bsz]
            lam, perm, box = self.draw_fn((self.bsz, 3) + tuple(self.img_hw)) if self.draw_fn is not None else (1.0, None, None)
            self.draws.append((lam, perm, box))
            own = batch[lo:hi]
            if perm is None:
                yield own
            else:
                yield own + [batch[j] for j in perm[lo:hi].tolist()]


class _NullLog:
    """loss.csv stand-in on ranks > 0."""

    def write(self, *_):
        pass

    def flush(self):
        pass

    def close(self):
        pass
