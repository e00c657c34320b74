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
        # N > 1: a give-up on any rank voids the step on every rank (include/swat_hip.h: swat_vit_mega_void_flag / _absorb).  Steps are
        # counted alike on every rank; a voided step is acted on VOID_LAG steps later, when its absorb launch has certainly run on
        # every rank's device — so all ranks turn to the replay at the same step without any extra host collective
        self._seq = 0
        self._absorbed = collections.deque(maxlen=16)   # (step number, event behind its absorb launch)
        self._void_flag = None
        # N > 1: the blocks' backward as ONE persistent launch with the buckets' all-reduce + AdamW trailing the weight gradients
        # (csrc/vit.hip: swat_vit_backward_set_buckets); SWAT_DP_PERSISTENT=0 keeps the stage-by-stage backward of rounds 2-4
        # (auto, the default: decided once, at the first data-parallel step, from a timed collective — _decide_dp_path)
        self._dp_mode = os.environ.get("SWAT_DP_PERSISTENT", "auto")
        self.dp_persistent = self._dp_mode != "0"
        self.dp_calibration = None
        self.lazy_join = os.environ.get("SWAT_LAZY_JOIN", "1") != "0"
        if self.trail:
            # process-wide streams owned by the library (created once, together with the weight-gradient lane)
            self._comm_stream = torch.cuda.ExternalStream(_lib.lib().swat_vit_aux_stream(0), device=dev)  # all-reduces
            self._opt_stream = torch.cuda.ExternalStream(_lib.lib().swat_vit_aux_stream(1), device=dev)   # AdamW slices at N > 1

    DP_COMM_BREAK_EVEN_MS = 1.35

    def _decide_dp_path(self):
        """SWAT_DP_PERSISTENT=auto, once per job, on every rank alike.  The persistent launches own every CU, so with them a bucket's
        collective can only run in the optimizer phase (~0.8 ms at the benchmark's shape): step ~ 1.9 ms + max(0.8, t_comm).  The
        stage-by-stage backward of rounds 2-4 is slower alone (3.4 ms) but its collectives run beside ordinary kernels through the whole
        backward: step ~ max(3.4, 0.9 + t_comm).  The two cross at t_comm ~ 1.35 ms — the gradient's time on the links, which is 4.6 /
        2.3 / 1.15 ms for fp32 and half that for bf16 on 2 / 4 / 8 GPUs by the link count alone.  So: time a
        32 MB collective of the wire's dtype on the communication backend, scale it to the arena, take the slowest rank's figure, and
        keep the persistent form when it is below the break-even.  Only on RCCL with more than one rank: the one-GPU tests' gloo
        groups (host round trips) and 1-rank groups keep the persistent form, as does any failure of the measurement."""
        self._dp_mode = "decided"
        try:
            force = os.environ.get("SWAT_DP_CALIBRATE_FORCE") == "1"   # tests: run the measurement in a 1-rank RCCL group
            if not (dist.is_initialized() and dist.get_backend() == "nccl" and (dist.get_world_size() > 1 or force)):
                return
            wire = self._wire()
            dt = torch.bfloat16 if wire.wire == "bf16" else torch.float32
            n = min(self.arena.numel, (32 << 20) // (2 if dt == torch.bfloat16 else 4))
            buf = torch.zeros(n, dtype=dt, device=self.arena.flat.device)
            for _ in range(2):
                dist.all_reduce(buf)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(4):
                dist.all_reduce(buf)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 4 * (self.arena.numel / n)
            t = torch.tensor([ms], dtype=torch.float32, device=buf.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
            self.dp_persistent = ms <= self.DP_COMM_BREAK_EVEN_MS
            self.dp_calibration = {"gradient_ms_on_the_links": round(ms, 3), "break_even_ms": self.DP_COMM_BREAK_EVEN_MS,
                                   "path": "persistent backward, buckets trailing" if self.dp_persistent else "stage-by-stage backward, collectives beside it"}
        except Exception as ex:  # noqa: a failed measurement must not stop a training job
            import warnings
            warnings.warn(f"swat_amd: could not time the gradient collective ({ex}); keeping the persistent data-parallel step")
            self.dp_persistent = True

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
            if self.reduce:
                self._check_void()
            elif lib.swat_vit_mega_poll():
                self._recover()
            self._seq += 1
            self._recent.append([0, self.opt.step_count, [g['lr'] for g in self.opt.param_groups],
                                 (images, target_a, target_b, lam, perm, box, batch), self._seq])
            self._cur_seq = self._seq
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
        blocks_fused = bool(self.train_visual and self.trail and _lib.lib().swat_vit_backward_blocks_fused(self.visual._last_ws[1]))
        self._late_loss = blocks_fused and not self.reduce
        # N > 1: the same one persistent launch for the blocks, its buckets reduced behind the weight gradients (_reduce_persistent)
        if self.reduce and self._dp_mode == "auto":
            self._decide_dp_path()
        self._dp_fused = blocks_fused and self.reduce and self.dp_persistent
        # N > 1: a step can be voided by ANOTHER rank's give-up, known only after the step's flag exchange: the running loss sum is
        # fed behind that exchange (swat_loss_accumulate), not by the loss kernel
        dp_loss_late = bool(self.reduce and self.trail and self.train_visual)
        ls_ptr = a.flat.data_ptr() + 4 * self.off_ls if self.off_ls is not None else _lib.ptr(self._ls_fixed)
        dls_ptr = g.data_ptr() + 4 * self.off_ls if self.off_ls is not None else _lib.ptr(self._dls_scratch)
        # (_late_loss: only what the backward waits for now; the batch means and d W_head behind the blocks' backward, _trailing_backward)
        head_args = [_lib.ptr(feats), a.flat.data_ptr() + 4 * self.off_head, ls_ptr, _lib.ptr(target_a), _lib.ptr(target_b), float(lam),
                     _lib.ptr(ws), _lib.ptr(self.loss), _lib.ptr(dfeats), g.data_ptr() + 4 * self.off_head, dls_ptr, None, None,
                     None if (self._late_loss or dp_loss_late) else _lib.ptr(self.loss_accum), B, self.E, self.C]
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

    def _bucket_plan(self):
        """Backward stages per gradient bucket, from the top of the tower down (stage L+1 = ln_post / proj, l+1 = block l, 0 = stem):
        bucket_sched[i] blocks per bucket; stage 1 always leaves with stage 0 (and the head)."""
        L = self.visual.layers
        plan, cur, blocks, bucket = [], [], 0, 0
        for s in range(L + 1, -1, -1):
            cur.append(s)
            blocks += 1 if 1 <= s <= L else 0
            want = self.bucket_sched[bucket] if bucket < len(self.bucket_sched) else 1 << 30
            if s == 0 or (blocks >= want and s > 1):
                plan.append(cur)
                cur, blocks, bucket = [], 0, bucket + 1
        return plan

    def _void_exchange(self, comm):
        """On `comm`, before the step's first gradient collective: one float per rank says whether a persistent launch of this step
        gave up here; its maximum over the ranks raises every rank's device guard (the step's optimizer launches and loss addition
        are then skipped everywhere) and publishes the step number for _check_void."""
        lib = _lib.lib()
        if self._void_flag is None:
            self._void_flag = torch.zeros(1, dtype=torch.float32, device=self.arena.flat.device)
        h = comm.cuda_stream
        _lib.check(lib.swat_vit_mega_void_flag(_lib.ptr(self._void_flag), h), "swat_vit_mega_void_flag")
        dist.all_reduce(self._void_flag, op=dist.ReduceOp.MAX)
        _lib.check(lib.swat_vit_mega_void_absorb(_lib.ptr(self._void_flag), int(self._cur_seq), h), "swat_vit_mega_void_absorb")
        if not self._replaying:
            ev = torch.cuda.Event()
            ev.record(comm)
            self._absorbed.append((self._cur_seq, ev))

    VOID_LAG = 4

    def _check_void(self, final=False):
        """N > 1.  Has some rank voided a step?  Decided from steps whose flag exchange has certainly completed on THIS device (the
        step VOID_LAG back; `final`: the caller has synchronised the device, every step counts), so that every rank answers alike
        and turns to the replay at the same step."""
        if not (self.reduce and self.trail and self.train_visual):
            return False
        lib = _lib.lib()
        if final:
            horizon = self._seq
        else:
            horizon = self._seq + 1 - self.VOID_LAG     # (called before step _seq + 1 is issued)
            for q, ev in self._absorbed:
                if q == horizon:
                    ev.synchronize()    # long done: the host runs a step or two ahead of the device, not four
        v = lib.swat_vit_mega_poll_void()
        if v and v <= horizon:
            self._recover_dp(v)
            return True
        return False

    def _recover_dp(self, v):
        """Step number v was voided on every rank (the guard has been up on every device since): switch every rank to the op-by-op
        path and run the steps from v on again — the same collectives in the same order on every rank."""
        import warnings
        lib = _lib.lib()
        torch.cuda.synchronize()
        st = lib.swat_vit_mega_status()   # clears the guard and both host words
        lib.swat_vit_set_mega(0)
        warnings.warn(f"data-parallel step {v}: a persistent launch gave up on a team hand-off on some rank (local status {st:#x}); "
                      "the step was skipped on every rank and is re-run op by op from here on")
        todo = [e for e in self._recent if e[4] >= v]
        if not todo or todo[0][4] != v:
            raise _lib.PersistentForwardGaveUp(
                f"step {v} was voided and is no longer in the replay window ({len(self._recent)} steps): weights and optimizer state "
                "are those of the step before it on every rank; restart from them with SWAT_VIT_MEGA=0")
        opt = self.opt
        lrs_now, count_now = [g['lr'] for g in opt.param_groups], opt.step_count
        opt.step_count = todo[0][1]
        self._absorbed.clear()
        self._replaying = True
        try:
            for e in todo:
                for g, lr in zip(opt.param_groups, e[2]):
                    g['lr'] = lr
                self._cur_seq = e[4]
                self(*e[3])
        finally:
            self._replaying = False
            for g, lr in zip(opt.param_groups, lrs_now):
                g['lr'] = lr
        assert opt.step_count == count_now, (opt.step_count, count_now)
        torch.cuda.synchronize()

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

        flushed = [0]

        def flush(bucket_event=None):
            lo, hi = min(p[0] for p in pend), max(p[1] for p in pend)
            assert sum(p[1] - p[0] for p in pend) == hi - lo, "bucket is not one contiguous arena range"
            if bucket_event is None:
                lib.swat_vit_lane_order(ready_h)  # `comm` follows everything the bucket's stages ran on either stream
            else:
                _lib.check(lib.swat_vit_backward_bucket_wait(bucket_event, ready_h), "swat_vit_backward_bucket_wait")
            with torch.cuda.stream(comm):
                if flushed[0] == 0:
                    self._void_exchange(comm)   # the step's persistent launches are behind `comm` now; no optimizer launch yet
                self._wire().reduce(g, lo, hi)  # all_reduce | reduce_scatter + all_gather, fp32 | bf16 on the links (ddp.GradWire)
            side.wait_stream(comm)
            if flushed[0] == 0:
                # (behind the flag exchange: a step voided by any rank adds nothing to the running loss on any rank)
                _lib.check(lib.swat_loss_accumulate(_lib.ptr(self.loss), _lib.ptr(self.loss_accum), side_h), "swat_loss_accumulate")
            flushed[0] += 1
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
                L = self.visual.layers
                plan = self._bucket_plan()
                if self._dp_fused:
                    # ONE persistent launch for the data gradients of all blocks (as on one GPU); on the lane behind it the buckets'
                    # weight gradients, each with an event — bucket i's all-reduce (`comm`) and AdamW (`side`) run beside the weight
                    # gradients of the buckets below it.  The collectives never share the chip with a persistent launch (every CU
                    # holds one of its workgroups: an RCCL kernel beside it would keep team members from becoming resident).
                    import ctypes
                    counts = [sum(1 for s_ in b if 1 <= s_ <= L) for b in plan]
                    self.visual.backward_raw(dfeats, stage_hi=L + 1, stage_lo=L + 1)
                    arr = (ctypes.c_int32 * len(counts))(*counts)
                    _lib.check(lib.swat_vit_backward_set_buckets(ctypes.cast(arr, ctypes.c_void_p), len(counts)), "swat_vit_backward_set_buckets")
                    self.visual.backward_raw(dfeats, stage_hi=L, stage_lo=1)
                    rec = lib.swat_vit_backward_buckets_recorded()   # 0: the blocks ran op by op after all (no persistent forward before it)
                    self.visual.backward_raw(dfeats, stage_hi=0, stage_lo=0)
                    for i, b in enumerate(plan):
                        for s_ in b:
                            lo, hi = self._stage_slices[s_]
                            pend.append((lo, hi, None))
                        if 0 in b:
                            pend.append((self.off_head, self.off_head + (self.C * self.E + 63) // 64 * 64, sl["head"]))
                        # the last bucket also holds the stem's gradients (main stream, behind the persistent launch): lane order
                        flush(i if (rec == len(plan) and 0 not in b) else None)
                else:
                    for b in plan:
                        for s in b:
                            self.visual.backward_raw(dfeats, stage_hi=s, stage_lo=s)
                            lo, hi = self._stage_slices[s]
                            pend.append((lo, hi, None))
                        if 0 in b:
                            pend.append((self.off_head, self.off_head + (self.C * self.E + 63) // 64 * 64, sl["head"]))
                        flush()
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
            # one GPU: the join with the lane / second stream (the blocks' weight gradients and AdamW) is OWED, not paid here — the next
            # step's stem reads stage-0 parameters only (stepped on the main stream above) and runs beside the optimizer's tail;
            # swat_vit_forward pays the join behind its stem (include/swat_hip.h: swat_vit_backward_join_later).  Whoever reads the
            # arena with torch ops before the next step calls join() (pop_loss_sum does)
            # Only the FUSED step (blocks as the persistent launch) steps the head, the logit scale and stage 0 on the main stream; the
            # op-by-op branch runs every AdamW on the lane, so the next stem (conv1 / cls / pos / ln_pre) would race with it: it joins
            if self.lazy_join and not self.reduce and self._late_loss:
                lib.swat_vit_backward_join_later(main.cuda_stream)
            else:
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
            for _, _, lrs, args, _seq in todo:
                for g, lr in zip(opt.param_groups, lrs):
                    g['lr'] = lr
                self(*args)
        finally:
            self._replaying = False
            for g, lr in zip(opt.param_groups, lrs_now):
                g['lr'] = lr
        assert opt.step_count == count_now, (opt.step_count, count_now)
        torch.cuda.synchronize()

    def join(self):
        """The current stream waits for everything the last step left on the library's side streams (the blocks' weight gradients and
        AdamW of a one-GPU step: see _trailing_backward).  Call before reading weights, moments or gradients with torch ops."""
        _lib.check(_lib.lib().swat_vit_join_pending(torch.cuda.current_stream().cuda_stream), "swat_vit_join_pending")

    def pop_loss_sum(self):
        self.join()
        v = self.loss_accum.item()
        # the persistent forward's hand-offs are bounded spins: a give-up is recorded, not hung on.  This is a synchronisation
        # point: nothing is in flight, so a failure that the per-call poll has not seen yet is visible now
        if self.train_visual and self.reduce:
            torch.cuda.synchronize()   # (every stream: the flag exchanges run on the collective stream)
            if self._check_void(final=True):
                v = self.loss_accum.item()
        elif self.train_visual and _lib.lib().swat_vit_mega_poll():
            self._recover()
            v = self.loss_accum.item()
        self.loss_accum.zero_()
        return v
