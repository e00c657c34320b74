"""One fused finetuning step on gfx950: CutMix paste -> encode_image -> L2-norm -> head -> scaled mixed CE ->
backward -> (data-parallel gradient reduction) -> AdamW, with no autograd graph and four host calls.

ref: the body of the hot loops utils/training.py:326-347 (train_ce) and :1452-1489 (train_cutmix).
"""
import collections
import os

import torch
import torch.distributed as dist

from . import _lib
from .ddp import GradWire, SliceReducer


class FusedTrainStep:
    def __init__(self, model, classifier_head, logit_scale, optimizer, world_size=None, staged=None, trail=True):
        self.visual = model.visual
        self.head = classifier_head
        self.logit_scale = logit_scale
        self.opt = optimizer
        self.arena = optimizer.arena
        W = classifier_head.linear.weight
        self.C, self.E = W.shape
        self.off_head = self.arena.offset_of(W)
        try:
            self.off_ls = self.arena.offset_of(logit_scale)
        except RuntimeError:
            self.off_ls = None  # fixed temperature (probing / zero-shot: ref utils/optimizers.py:51,58), not trained
        self.train_visual = any(g.get('swat_tower') is self.visual for g in optimizer.param_groups)
        dev = self.arena.flat.device
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.loss_accum = torch.zeros(1, dtype=torch.float64, device=dev)
        self._dls_scratch = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ls_fixed = None if self.off_ls is not None else logit_scale.detach().reshape(-1)[:1].to(dev, torch.float32).contiguous()  # per-epoch running sum: ONE host sync per epoch
        self._ws = {}
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.opt.grad_scale = 1.0 / self.world
        # staged backward (one call per stage + a slice all-reduce after each) is the N > 1 path; `staged=True` forces it
        # at world 1 so single-GPU tests exercise exactly the code the 8-GPU job runs
        self.staged = (self.world > 1) if staged is None else bool(staged)
        # one GPU: AdamW of the blocks' matrices inside their weight-gradient launch (csrc/gemm_xl.hip) when the blocks' backward runs
        # as the persistent launch; SWAT_FUSE_ADAMW=0 keeps the separate optimizer launch
        import os
        self.fuse_adamw = os.environ.get("SWAT_FUSE_ADAMW", "1") != "0"
        self.reduce = self.staged and dist.is_initialized()
        # contiguous gradient slices per backward stage (the tower is laid out stage by stage in the arena)
        self._stage_slices = []
        if self.train_visual:
            for s in range(self.visual.layers + 2):
                ps = self.visual.stage_params(s)
                lo = self.arena.offset_of(ps[0])
                hi = self.arena.offset_of(ps[-1]) + (ps[-1].numel() + 63) // 64 * 64
                self._stage_slices.append((lo, hi))
        # trailing mode: per backward stage, the slice's (all-reduce and) AdamW trail the weight-gradient lane
        # (swat_hip.h: swat_vit_set_overlap mode 2), so the HBM-bound optimizer overlaps the GEMM-bound backward of the
        # stages below.  Needs the slices to tile the whole arena.
        # N > 1: transformer blocks per gradient collective, from the top of the tower down.  Default: graduated — two buckets of L/3
        # blocks while most of the backward is still ahead to hide them, then halving, so that the LAST collective (block 1, the stem
        # and the head) is small: the next step's forward (one persistent launch that reads every weight) cannot start before that
        # bucket's all-reduce and AdamW are through, so its size is exposed in full.  ViT-B/32: 4 + 4 + 2 + 1 + (1 + stem + head) =
        # 113 / 113 / 57 / 28 / 31 MB instead of 4 x 85 MB.  SWAT_DP_BUCKETS=a,b,c overrides the list, SWAT_DP_BUCKET_BLOCKS=n makes
        # every bucket n blocks (the round-2 behaviour with n = 3).
        L_ = self.visual.layers if self.train_visual else 0
        if os.environ.get("SWAT_DP_BUCKETS"):
            self.bucket_sched = [max(1, int(x)) for x in os.environ["SWAT_DP_BUCKETS"].split(",")]
        elif os.environ.get("SWAT_DP_BUCKET_BLOCKS"):
            self.bucket_sched = [max(1, int(os.environ["SWAT_DP_BUCKET_BLOCKS"]))] * 64
        else:
            a_ = max(1, L_ // 3)
            self.bucket_sched, r_ = [a_, a_], L_ - 2 * a_
            while r_ > 1:
                t_ = max(1, r_ // 2)
                self.bucket_sched.append(t_)
                r_ -= t_
        self.trail = False
        if trail and self.train_visual:
            sl = sorted(self._stage_slices + [(self.off_head, self.off_head + (self.C * self.E + 63) // 64 * 64)]
                        + ([(self.off_ls, self.off_ls + 64)] if self.off_ls is not None else []))
            self.trail = sl[0][0] == 0 and sl[-1][1] == self.arena.numel and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        # the last steps' arguments: a persistent forward that gives up on a hand-off (csrc/vit_mega.hip; only when a team member is
        # not resident) voids its step — the device then skips every AdamW launch until the host has noticed (swat_vit_mega_poll at
        # the next call, no synchronisation), and the host re-runs the steps from the failed launch on, op by op (_recover)
        self._recent = collections.deque(maxlen=16)
        self._replaying = False
        if self.trail:
            # process-wide streams owned by the library (created once, together with the weight-gradient lane)
            self._comm_stream = torch.cuda.ExternalStream(_lib.lib().swat_vit_aux_stream(0), device=dev)  # all-reduces
            self._opt_stream = torch.cuda.ExternalStream(_lib.lib().swat_vit_aux_stream(1), device=dev)   # AdamW slices at N > 1

    def _wire(self):
        w = getattr(self, "_grad_wire", None)
        if w is None:
            w = self._grad_wire = GradWire(self.arena.numel, self.arena.flat.device)
        return w

    def _head_ws(self, B, dev):
        ws = self._ws.get(B)
        if ws is None:
            n = _lib.lib().swat_head_loss_workspace_bytes(B, self.E, self.C)
            ws = (torch.empty(n, dtype=torch.uint8, device=dev), torch.empty((B, self.E), dtype=torch.float32, device=dev))
            self._ws[B] = ws
        return ws

    def __call__(self, images, target_a, target_b=None, lam=1.0, perm=None, box=None, batch=None):
        """images fp32 [B,3,H,W] (GPU), targets int64 [B].  Returns the device scalar holding this step's loss.
        batch: number of samples when `images` carries CutMix partner rows behind them (swat_amd/dp.py:shard_batch)."""
        if self.train_visual and not self._replaying:
            lib = _lib.lib()
            if lib.swat_vit_mega_poll():
                self._recover()
            self._recent.append([0, self.opt.step_count, [g['lr'] for g in self.opt.param_groups],
                                 (images, target_a, target_b, lam, perm, box, batch)])
        a = self.arena
        B = images.shape[0] if batch is None else int(batch)
        dev = images.device
        if target_b is None:
            target_b = target_a
        ws, dfeats = self._head_ws(B, dev)
        red = SliceReducer()
        if self.train_visual:
            n0 = _lib.lib().swat_vit_mega_launches()
            feats = self.visual.forward_raw(images, save=True, perm=perm, box=box, batch=B)
            if not self._replaying and _lib.lib().swat_vit_mega_launches() > n0:
                self._recent[-1][0] = n0 + 1  # this step's forward was persistent launch number n0 + 1 of the device
        else:
            feats = images.float().contiguous()  # pre-extracted, already normalised features (train_probing)
        g = a.grad
        # a step whose blocks run backward as ONE persistent launch (one GPU, trailing mode) can still be voided by that launch: its
        # loss joins the running sum behind it (on the lane, _trailing_backward), not in the loss kernel
        self._late_loss = bool(self.train_visual and self.trail and not self.reduce and
                               _lib.lib().swat_vit_backward_blocks_fused(self.visual._last_ws[1]))
        ls_ptr = a.flat.data_ptr() + 4 * self.off_ls if self.off_ls is not None else _lib.ptr(self._ls_fixed)
        dls_ptr = g.data_ptr() + 4 * self.off_ls if self.off_ls is not None else _lib.ptr(self._dls_scratch)
        # (_late_loss: only what the backward waits for now; the batch means and d W_head behind the blocks' backward, _trailing_backward)
        head_args = [_lib.ptr(feats), a.flat.data_ptr() + 4 * self.off_head, ls_ptr, _lib.ptr(target_a), _lib.ptr(target_b), float(lam),
                     _lib.ptr(ws), _lib.ptr(self.loss), _lib.ptr(dfeats), g.data_ptr() + 4 * self.off_head, dls_ptr, None, None,
                     None if self._late_loss else _lib.ptr(self.loss_accum), B, self.E, self.C]
        self._head_rest = (head_args, int(self.train_visual)) if self._late_loss else None
        rc = _lib.lib().swat_head_loss(*head_args, 2 if self._late_loss else 1, int(self.train_visual), _lib.stream())
        _lib.check(rc, "swat_head_loss")
        if self.trail:
            return self._trailing_backward(g, dfeats)
        if self.reduce:
            red.reduce(g, self.off_head, self.off_head + self.C * self.E)  # head group (first in the arena)
            if self.off_ls is not None:
                red.reduce(g, self.off_ls, self.off_ls + 1)                  # logit_scale group (last)
        if self.train_visual:
            L = self.visual.layers
            if self.staged:
                # stage by stage: reduce a stage's slice (RCCL, its own stream) while the next stage's kernels run
                for s in range(L + 1, -1, -1):
                    self.visual.backward_raw(dfeats, stage_hi=s, stage_lo=s)
                    if self.reduce:
                        lo, hi = self._stage_slices[s]
                        red.reduce(g, lo, hi)
            else:
                self.visual.backward_raw(dfeats)
        red.wait()
        self.opt.step()
        return self.loss

    def _adamw_args(self, lo, hi):
        """Raw-pointer AdamW arguments of an arbitrary arena range inside one LR group (cached per range)."""
        sl = self._slice_args()
        key = ("range", lo, hi)
        if key not in sl:
            sl[key] = self._mk_adamw(lo, hi)
        return sl[key]

    def _slice_args(self):
        """Raw-pointer argument tuples of the per-slice AdamW launches (rebuilt when the arena moves)."""
        import ctypes
        a, opt = self.arena, self.opt
        key = (a.flat.data_ptr(), a.grad.data_ptr(), opt.grad_scale)
        if getattr(self, "_slice_key", None) != key:
            zero = (ctypes.c_int64 * 1)(0)
            self._zero_seg = zero

            def mk(lo, hi):
                gi = max(i for i, b in enumerate(opt.seg_begin) if b <= lo)
                assert gi + 1 == len(opt.seg_begin) or hi <= opt.seg_begin[gi + 1], "slice crosses an LR group"
                return (a.flat.data_ptr() + 4 * lo, a.grad.data_ptr() + 4 * lo, a.exp_avg.data_ptr() + 4 * lo,
                        a.exp_avg_sq.data_ptr() + 4 * lo, a.shadow.data_ptr() + 2 * lo, hi - lo,
                        ctypes.cast(zero, ctypes.c_void_p), 1, opt._dev.data_ptr() + 4 * gi, float(opt.betas[0]),
                        float(opt.betas[1]), float(opt.eps), float(opt.weight_decay), opt._dev.data_ptr() + 16,
                        float(opt.grad_scale))
            self._mk_adamw = mk
            self._slice_cache = {"head": mk(self.off_head, self.off_head + (self.C * self.E + 63) // 64 * 64),
                                 "ls": mk(self.off_ls, self.off_ls + 64) if self.off_ls is not None else None,
                                 "stages": [mk(lo, hi) for lo, hi in self._stage_slices]}
            self._slice_key = key
        return self._slice_cache

    def _trailing_backward(self, g, dfeats):
        lib = _lib.lib()
        main = torch.cuda.current_stream()
        side = self._opt_stream
        comm = self._comm_stream
        lane_h = lib.swat_vit_lane_stream()
        # One GPU: the AdamW slices go straight onto the weight-gradient lane (main + lane are the only streams: every
        # extra stream is another chance of sharing a hardware queue with one of them, which was measured to triple the
        # step time — tools/micro/sec_order32.py).  N > 1: all-reduces on `comm` (so the next slice's all-reduce does not
        # queue behind this slice's AdamW), AdamW on `side` behind them.
        side_h = side.cuda_stream if self.reduce else lane_h
        ready_h = comm.cuda_stream
        opt = self.opt
        sl = self._slice_args()

        def finish(args, lo, n, stream_h=None):  # one GPU: this slice's AdamW behind the lane (or on the stream given)
            _lib.check(lib.swat_adamw(*args, side_h if stream_h is None else stream_h), "swat_adamw")

        # N > 1: gradients leave in BUCKETS of `self.bucket_sched[i]` transformer blocks (one contiguous arena range each:
        # logit_scale rides in the first bucket, stage 0 and the head in the last), all-reduced on `comm` (RCCL, stream-ordered) with the
        # range's AdamW launches behind it on `side`.  ViT-B/32: 5 collectives per step instead of 16 of <= 28 MB — fewer launch
        # latencies on the host and on the links, still overlapped with the backward of the blocks below.
        pend = []  # (lo, hi, adamw args) of the open bucket, in arena order of issue

        def flush():
            lo, hi = min(p[0] for p in pend), max(p[1] for p in pend)
            assert sum(p[1] - p[0] for p in pend) == hi - lo, "bucket is not one contiguous arena range"
            lib.swat_vit_lane_order(ready_h)  # `comm` follows everything the bucket's stages ran on either stream
            with torch.cuda.stream(comm):
                self._wire().reduce(g, lo, hi)  # all_reduce | reduce_scatter + all_gather, fp32 | bf16 on the links (ddp.GradWire)
            side.wait_stream(comm)
            vis = [p for p in pend if p[2] is None]
            if vis:  # the tower's stages of the bucket: one AdamW launch over their joint range (one LR group)
                _lib.check(lib.swat_adamw(*self._adamw_args(min(p[0] for p in vis), max(p[1] for p in vis)), side_h), "swat_adamw")
            for p in pend:
                if p[2] is not None:
                    _lib.check(lib.swat_adamw(*p[2], side_h), "swat_adamw")
            pend.clear()

        # this step's optimizer scalars (learning rates, bias corrections) go up on the stream the AdamW launches run on, not on the
        # main stream between the loss and the backward (tools/step_timeline.py: the 4-byte copy and the bubble behind it cost the
        # data-gradient chain ~15 us per step)
        if self.reduce:
            opt_stream = side
        else:
            if getattr(self, "_lane_ext", None) is None:
                self._lane_ext = torch.cuda.ExternalStream(lane_h, device=main.device)
            opt_stream = self._lane_ext
        _lib.check(lib.swat_vit_set_overlap(2), "swat_vit_set_overlap")
        try:
            if not self.reduce:
                # the lane first follows the main stream: the previous step's small optimizer launches ran THERE (below) and still
                # read the scalars this upload overwrites
                _lib.check(lib.swat_vit_lane_follow(main.cuda_stream), "swat_vit_lane_follow")
            with torch.cuda.stream(opt_stream):
                opt.begin_step()
                scalars_up = None
                if not self.reduce:
                    if getattr(self, "_scalars_ev", None) is None:
                        self._scalars_ev = torch.cuda.Event()
                    scalars_up = self._scalars_ev
                    scalars_up.record()   # (the main stream's own optimizer launches below wait for the upload)
            if self.reduce:
                side.wait_stream(main)  # head gradients + this step's scalars
                comm.wait_stream(main)
                # arena order: [head | stage 0 | stage 1 | ... | stage L+1 | logit_scale] — the backward walks it from the top down
                if self.off_ls is not None:
                    pend.append((self.off_ls, self.off_ls + 64, sl["ls"]))
                blocks, bucket = 0, 0
                for s in range(self.visual.layers + 1, -1, -1):
                    self.visual.backward_raw(dfeats, stage_hi=s, stage_lo=s)
                    lo, hi = self._stage_slices[s]
                    pend.append((lo, hi, None))
                    blocks += 1 if 1 <= s <= self.visual.layers else 0
                    if s == 0:
                        pend.append((self.off_head, self.off_head + (self.C * self.E + 63) // 64 * 64, sl["head"]))
                    want = self.bucket_sched[bucket] if bucket < len(self.bucket_sched) else 1 << 30
                    if s == 0 or (blocks >= want and s > 1):  # (stage 1 always leaves with stage 0 and the head)
                        flush()
                        blocks, bucket = 0, bucket + 1
            else:
                L = self.visual.layers
                fused_blocks = self._late_loss

                def finish_head(stream_h=None):
                    finish(sl["head"], self.off_head, self.C * self.E, stream_h)
                    if self.off_ls is not None:
                        finish(sl["ls"], self.off_ls, 1, stream_h)
                if not fused_blocks:
                    finish_head()
                if fused_blocks:
                    # the blocks' data gradients run as ONE persistent launch (csrc/vit_mega.hip): one call for all of them, their
                    # weight gradients follow on the lane and one AdamW launch over the blocks' joint arena range behind those
                    # (the optimizer launches of the head, the logit scale and stage L+1 wait until the persistent backward is behind
                    # them on the lane: a hand-off it gives up on voids the step, and the device guard must be up before ANY update
                    # of the step — the host then re-runs the whole step, swat_amd/step.py:_recover)
                    for hi_s, lo_s in ((L + 1, L + 1), (L, 1), (0, 0)):
                        if hi_s == lo_s:
                            if hi_s == L + 1:
                                lib.swat_vit_backward_defer_top()   # (d proj behind the persistent backward: its AdamW runs there too)
                            self.visual.backward_raw(dfeats, stage_hi=hi_s, stage_lo=lo_s)
                            if hi_s == 0:
                                lo, hi = self._stage_slices[0]
                                finish(sl["stages"][0], lo, hi - lo, main.cuda_stream)   # (behind the stem backward, on its stream)
                        else:
                            lo, hi = self._stage_slices[lo_s][0], self._stage_slices[hi_s][1]
                            args = self._adamw_args(lo, hi)
                            if self.fuse_adamw:
                                # AdamW of the blocks inside their weight-gradient launch (include/swat_hip.h: swat_vit_arm_fused_adamw)
                                a = self.arena
                                _lib.check(lib.swat_vit_arm_fused_adamw(a.flat.data_ptr(), a.grad.data_ptr(), a.exp_avg.data_ptr(),
                                                                        a.exp_avg_sq.data_ptr(), a.shadow.data_ptr(), args[8], args[13],
                                                                        args[9], args[10], args[11], args[12], args[14]),
                                           "swat_vit_arm_fused_adamw")
                            self.visual.backward_raw(dfeats, stage_hi=hi_s, stage_lo=lo_s)
                            if not (self.fuse_adamw and lib.swat_vit_fused_adamw_consumed()):
                                finish(args, lo, hi - lo)
                            # the small optimizer launches of the step go on the MAIN stream, right behind the persistent backward and
                            # beside the lane's weight gradients — at the lane's tail they would stand between this step and the next
                            # (0.07 ms); the main stream first waits for this step's optimizer scalars (uploaded on the lane)
                            main.wait_event(scalars_up)
                            mh = main.cuda_stream
                            _lib.check(lib.swat_head_loss(*self._head_rest[0], 3, self._head_rest[1], mh), "swat_head_loss")
                            finish_head(mh)
                            lo, hi = self._stage_slices[L + 1]
                            finish(sl["stages"][L + 1], lo, hi - lo, mh)
                            _lib.check(lib.swat_loss_accumulate(_lib.ptr(self.loss), _lib.ptr(self.loss_accum), mh), "swat_loss_accumulate")
                else:
                    for s in range(L + 1, -1, -1):
                        self.visual.backward_raw(dfeats, stage_hi=s, stage_lo=s)
                        lo, hi = self._stage_slices[s]
                        finish(sl["stages"][s], lo, hi - lo)
        finally:
            lib.swat_vit_backward_join(main.cuda_stream)
            lib.swat_vit_set_overlap(1)
        if self.reduce:
            main.wait_stream(side)
        opt.end_step()
        return self.loss

    def _recover(self):
        """A persistent forward gave up on a hand-off (a team member was not resident).  Since that launch the device has skipped
        every AdamW launch and every addition to the running loss, so weights, moments and the loss sum are those of the last good
        step: switch to the op-by-op forward and run the voided steps again, with their own learning rates and step numbers."""
        torch.cuda.synchronize()
        failed = _lib.persistent_forward_failed()
        if not failed:
            return
        if self.reduce:
            raise _lib.PersistentForwardGaveUp(
                f"persistent forward launch {failed} gave up on a hand-off on this rank: its void gradients have been summed into the "
                "other ranks' updates; restart from the last checkpoint with SWAT_VIT_MEGA=0")
        first = next((i for i, e in enumerate(self._recent) if e[0] == failed), None)
        todo = list(self._recent)[first:] if first is not None else []   # every step since then had its update skipped
        if not todo:
            raise _lib.PersistentForwardGaveUp(
                f"persistent forward launch {failed} gave up on a hand-off and its step is no longer in the replay window "
                f"({len(self._recent)} steps): weights and optimizer state are those of the step before it; restart from them "
                "with SWAT_VIT_MEGA=0")
        opt = self.opt
        lrs_now, count_now = [g['lr'] for g in opt.param_groups], opt.step_count
        opt.step_count = todo[0][1]
        self._replaying = True
        try:
            for _, _, lrs, args in todo:
                for g, lr in zip(opt.param_groups, lrs):
                    g['lr'] = lr
                self(*args)
        finally:
            self._replaying = False
            for g, lr in zip(opt.param_groups, lrs_now):
                g['lr'] = lr
        assert opt.step_count == count_now, (opt.step_count, count_now)
        torch.cuda.synchronize()

    def pop_loss_sum(self):
        v = self.loss_accum.item()
        # the persistent forward's hand-offs are bounded spins: a give-up is recorded, not hung on.  This is a synchronisation
        # point: nothing is in flight, so a failure that the per-call poll has not seen yet is visible now
        if self.train_visual and _lib.lib().swat_vit_mega_poll():
            self._recover()
            v = self.loss_accum.item()
        self.loss_accum.zero_()
        return v
