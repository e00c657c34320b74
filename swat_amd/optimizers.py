"""Parameter groups, fused AdamW and the warm-up + cosine LR schedule of SWAT's finetuning
(mirror of the reference's utils/optimizers.py: same function names, argument meaning and LR trace).

ref: lock_text_tower :12-19, set_optimizer :22-35, set_params :38-127, get_optimizer :131-146,
     LinearWarmupScheduler :158-206.
The optimizer is ONE gfx950 kernel launch over a flat parameter arena (csrc/elementwise.hip adamw_kernel).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .arena import Arena

FINETUNE_METHODS = ("finetune", "finetune-multitask", "finetune-mixed", "dataset-cls", "mixup", "mixup-fs", "cutmix",
                    "cutmix-fs", "resizemix", "saliencymix", "attentivemix", "CMO", "fixmatch")


def lock_text_tower(model):
    """ref :12-19"""
    for m in [model.transformer, model.token_embedding, model.positional_embedding, model.ln_final, model.text_projection]:
        if type(m) is nn.Parameter:
            m.requires_grad = False
        else:
            for p in m.parameters():
                p.requires_grad = False


def set_params(args, model, classifier_head, logger, dataset_classifier_head=None):
    """ref :38-127 (zeroshot / probing / finetune-family branches; FLYP and multitask are out of scope)."""
    params_classifier = [{'params': list(classifier_head.parameters()), 'lr': args.lr_classifier}]
    params_visual = [{'params': list(model.visual.parameters()), 'lr': args.lr_backbone, 'swat_visual': model.visual}]

    if args.method == "zeroshot":
        logger.info('zeroshot only.')
        for param in model.parameters():
            param.requires_grad = False
        params = params_classifier  # place holder
        logit_scale = torch.tensor([4.60517]).to(device=args.device)
    elif args.method in ("probing", "REAL-Linear", "CMLP"):
        logger.info('Freezing the visual encoder.')
        for param in model.parameters():
            param.requires_grad = False
        params = params_classifier
        logit_scale = torch.tensor([4.60517]).to(device=args.device)  # ln(100): temperature 0.01
    elif args.method in FINETUNE_METHODS:
        logger.info('Training the visual encoder and linear head.')
        for param in model.parameters():
            param.requires_grad = True
        lock_text_tower(model)
        params = params_classifier + params_visual
        logit_scale = nn.Parameter((torch.ones([]) * np.log(1 / args.temperature)).to(torch.float32).to(args.device))
        params.append({'params': [logit_scale], 'lr': args.lr_classifier})
    else:
        raise NotImplementedError(f'Method {args.method} not implemented.')
    args.logit_scale = logit_scale
    return params, logit_scale


class FusedAdamW:
    """torch.optim.AdamW semantics (betas (0.9, 0.999), eps 1e-8, decoupled weight decay on every tensor — ref :143-146)
    over a flat arena.  `param_groups` keeps the torch layout so schedulers / checkpoint code work unchanged."""

    def __init__(self, params, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, device=None):
        self.param_groups = []
        ordered = []
        self.seg_begin = []
        for g in params:
            g = dict(g)
            plist = list(g['params'])
            tower = g.pop('swat_visual', None)
            if tower is not None:  # lay the tower out stage by stage (contiguous slices per backward stage)
                want = tower.ordered_params()
                assert {id(p) for p in want} == {id(p) for p in plist}
                plist = want
            g['params'] = plist
            g.setdefault('betas', betas)
            g.setdefault('eps', eps)
            g.setdefault('weight_decay', weight_decay)
            g['initial_lr'] = g['lr']
            g['swat_tower'] = tower
            self.param_groups.append(g)
        assert 1 <= len(self.param_groups) <= 4, "fused AdamW supports up to 4 LR groups"
        dev = device or next(p.device for g in self.param_groups for p in g['params'])
        if torch.device(dev).type != "cuda":
            raise RuntimeError("swat_amd's fused AdamW needs GPU parameters (no CPU fallback)")
        for g in self.param_groups:
            self.seg_begin.append(sum((p.numel() + 63) // 64 * 64 for p in ordered))
            ordered += g['params']
        self.arena = Arena(ordered, dev, with_grad=True, with_adam=True)
        for g in self.param_groups:
            if g['swat_tower'] is not None:
                g['swat_tower'].adopt_arena(self.arena)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.step_count = 0
        # step scalars (group LRs, bias corrections) go up through a ring of pinned rows: the host may run several
        # steps ahead of the GPU, so a row is rewritten only after the copy that read it has executed
        self._host = torch.zeros(8, 6, dtype=torch.float32).pin_memory()
        self._host_ev = [None] * 8
        self._dev = torch.zeros(6, dtype=torch.float32, device=dev)
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none=True):
        pass  # every backward WRITES the arena gradient; nothing accumulates

    def begin_step(self):
        """Bump the step count and upload this step's scalars (current stream)."""
        self.step_count += 1
        t = self.step_count
        slot = t % len(self._host_ev)
        if self._host_ev[slot] is not None:
            self._host_ev[slot].synchronize()
        row = self._host[slot]
        for i, g in enumerate(self.param_groups):
            row[i] = g['lr']
        row[4] = 1.0 - self.betas[0] ** t
        row[5] = 1.0 - self.betas[1] ** t
        self._dev.copy_(row, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._host_ev[slot] = ev

    def step_slice(self, lo, hi):
        """AdamW on arena elements [lo, hi) (inside ONE LR group) on the current stream: the per-stage form the fused
        step trails behind the backward.  Call begin_step() first and end_step() after the last slice."""
        a = self.arena
        gi = max(i for i, b in enumerate(self.seg_begin) if b <= lo)
        assert gi + 1 == len(self.seg_begin) or hi <= self.seg_begin[gi + 1], "slice crosses an LR group"
        ops.adamw_step(a.flat[lo:hi], a.grad[lo:hi], a.exp_avg[lo:hi], a.exp_avg_sq[lo:hi], a.shadow[lo:hi], [0],
                       self._dev[gi:], self._dev[4:], self.betas[0], self.betas[1], self.eps, self.weight_decay,
                       self.grad_scale)

    def end_step(self):
        self.arena.mark_shadow_fresh()

    def step(self):
        self.begin_step()
        a = self.arena
        ops.adamw_step(a.flat, a.grad, a.exp_avg, a.exp_avg_sq, a.shadow, self.seg_begin, self._dev, self._dev[4:],
                       self.betas[0], self.betas[1], self.eps, self.weight_decay, self.grad_scale)
        self.end_step()

    def import_grads(self):
        """Copy autograd-produced p.grad tensors into the arena (compatibility path: loss.backward() by torch)."""
        a = self.arena
        for i, p in enumerate(a.params):
            if p.grad is not None:
                gv = a.grad_view(i)
                if p.grad.data_ptr() != gv.data_ptr():
                    gv.copy_(p.grad)
            else:
                a.grad_view(i).zero_()

    def state_dict(self):
        return {'step': self.step_count, 'exp_avg': self.arena.exp_avg, 'exp_avg_sq': self.arena.exp_avg_sq,
                'param_groups': [{k: v for k, v in g.items() if k not in ('params', 'swat_tower')} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.step_count = sd['step']
        self.arena.exp_avg.copy_(sd['exp_avg'])
        self.arena.exp_avg_sq.copy_(sd['exp_avg_sq'])
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update(s)


def get_optimizer(params, optim_type, wd):
    """ref :131-146"""
    if optim_type == 'AdamW':
        return FusedAdamW(params, betas=(0.9, 0.999), weight_decay=wd)
    raise NotImplementedError(f"optimizer {optim_type}: only AdamW (the reference default, utils/parser.py) has a HIP kernel")


class LinearWarmupScheduler:
    """Closed form of the reference's LinearWarmupScheduler(CosineAnnealingLR(T_max=total_iter, eta_min=1e-9),
    warmup 50 its, first LR = warmup_lr for every group) — ref :22-35, :158-206.  The LR set after k calls of step()
    is the one optimizer.step() number k (0-based) uses:
        k = 0          -> warmup_lr            (every group)
        1 <= k < 50    -> base * k / 50
        k >= 50        -> eta_min + (base - eta_min) * (1 + cos(pi * (k - 50) / T_max)) / 2
    """

    def __init__(self, optimizer, total_iter, warmup_iter=50, warmup_lr=1e-6, eta_min=1e-9):
        self.optimizer, self.total_iter, self.warmup_iter = optimizer, total_iter, warmup_iter
        self.warmup_lr, self.eta_min = warmup_lr, eta_min
        self.base_lrs = [g['initial_lr'] for g in optimizer.param_groups]
        self.last_epoch = 0
        self._apply()

    def lr_at(self, k, base):
        if k == 0:
            return self.warmup_lr
        if k < self.warmup_iter:
            return base * k / self.warmup_iter
        return self.eta_min + (base - self.eta_min) * (1 + math.cos(math.pi * (k - self.warmup_iter) / self.total_iter)) / 2

    def _apply(self):
        self._last_lr = [self.lr_at(self.last_epoch, b) for b in self.base_lrs]
        for g, lr in zip(self.optimizer.param_groups, self._last_lr):
            g['lr'] = lr

    def step(self, epoch=None):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return list(self._last_lr)

    def state_dict(self):
        return {'last_epoch': self.last_epoch, 'total_iter': self.total_iter, 'base_lrs': self.base_lrs}

    def load_state_dict(self, sd):
        self.last_epoch = sd['last_epoch']
        self._apply()


def set_optimizer(args, params, train_loader):
    """ref :22-35"""
    optimizer = get_optimizer(params, optim_type=args.optim, wd=args.wd)
    if isinstance(train_loader, tuple):
        total_iter = len(train_loader[0]) * args.epochs
    else:
        total_iter = len(train_loader) * args.epochs
    warmup_lr = 1e-5 if args.lr_backbone > 5e-5 else 1e-6
    scheduler = LinearWarmupScheduler(optimizer, total_iter, warmup_iter=50, warmup_lr=warmup_lr, eta_min=1e-9)
    return optimizer, scheduler, total_iter
