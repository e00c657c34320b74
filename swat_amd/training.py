"""`--method` plug-ins: the finetuning loops of SWAT on the fused gfx950 step
(mirror of the reference's utils/training.py: same names, signatures, return values and loss.csv rows).

ref: set_training_seed :10-20, run_zeroshot :23-44, train_probing :47-161, train_ce :296-412, mixup_criterion
     :1157-1158, rand_bbox :1404-1420, train_cutmix :1422-1556.

Differences from the reference that do not change results:
  * the whole step runs in swat_amd.step.FusedTrainStep (no autograd graph, no per-step .item()): the loss is
    accumulated on the device and read once per epoch for the loss.csv row;
  * the CutMix paste is fused into patch extraction (the input batch is not modified in place);
  * host RNG consumption order per iteration is the reference's: np.random.rand(1) ->
    [np.random.beta] -> torch.randperm(B) (global CPU generator) -> np.random.randint(W) -> np.random.randint(H).
"""
import copy
import random

import numpy as np
import torch

from . import dp
from .step import FusedTrainStep
from .testing import validate, calculate_scores, load_model
from .models import save_model_ckpt


def set_training_seed(args):
    random.seed(args.training_seed)
    torch.manual_seed(args.training_seed)
    np.random.seed(args.training_seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(args.training_seed)


def run_zeroshot(args, test_loader, model, logger, loss, logit_scale, classifier_head):
    zs_test_acc, zs_loss, zs_confusion_matrix = validate(args, data_loader=test_loader, model=model, logger=logger, loss=loss,
                                                         logit_scale=logit_scale, classifier_head=classifier_head,
                                                         show_confusion_matrix=True, dataset=args.dataset,
                                                         output_dir=args.output_dir, device=args.device,
                                                         pre_extracted=args.pre_extracted)
    logger.info(f"+++++ Zero-shot Test Acc: {round(zs_test_acc, 3)}")
    return zs_test_acc


def mixup_criterion(criterion, pred, y_a, y_b, lam):
    return lam * criterion(pred, y_a) + (1.0 - lam) * criterion(pred, y_b)


def rand_bbox(size, lam):
    """ref :1404-1420 (W = size[2], H = size[3]; both 224 here)."""
    W = size[2]
    H = size[3]
    cut_rat = np.sqrt(1. - lam)
    cut_w = int(W * cut_rat)
    cut_h = int(H * cut_rat)
    cx = np.random.randint(W)
    cy = np.random.randint(H)
    bbx1 = np.clip(cx - cut_w // 2, 0, W)
    bby1 = np.clip(cy - cut_h // 2, 0, H)
    bbx2 = np.clip(cx + cut_w // 2, 0, W)
    bby2 = np.clip(cy + cut_h // 2, 0, H)
    return bbx1, bby1, bbx2, bby2


class _PinnedRing:
    """Pinned host buffers for the per-iteration permutation.  `torch.randperm(B).cuda()` (ref :1462) copies from PAGEABLE memory:
    hipMemcpyAsync then stages the copy and returns only when the stream has reached it — the host, which otherwise runs a step or
    two ahead of the GPU, stood still once per CutMix iteration and every step began with an idle GPU (tools/step_timeline.py: the
    copy sits at the very end of the previous step's stream).  Drawn into a pinned buffer (same generator, same values) the copy is
    asynchronous.  64 slots; a slot's next draw waits for the event behind its last upload (to_device)."""

    def __init__(self, slots=64):
        self.slots, self.i, self.buf, self.ev = slots, 0, {}, {}

    def take(self, n):
        if not torch.cuda.is_available():
            return None
        self.i = (self.i + 1) % self.slots
        key = (self.i, n)
        if key not in self.buf:
            self.buf[key] = torch.empty(n, dtype=torch.int64).pin_memory()
        ev = self.ev.pop(key, None)
        if ev is not None:
            ev.synchronize()
        return self.buf[key]

    def uploaded(self, t):
        for key, b in self.buf.items():
            if b.data_ptr() == t.data_ptr():
                ev = torch.cuda.Event()
                ev.record()
                self.ev[key] = ev
                return


_perm_ring = _PinnedRing()


def to_device(perm, device):
    """The permutation of cutmix_draw on the device, without stalling the host (see _PinnedRing)."""
    out = perm.to(device, non_blocking=True)
    if perm.is_pinned():
        _perm_ring.uploaded(perm)
    return out


def cutmix_draw(args, size):
    """The host-side draws of one train_cutmix iteration (ref :1458-1472) -> (lam, rand_index|None, box|None).  rand_index is a
    pinned host tensor when a GPU is present (upload it with to_device)."""
    r = np.random.rand(1)
    if args.cutmix_beta > 0 and r < args.mix_prob:
        lam = np.random.beta(args.cutmix_beta, args.cutmix_beta)
        pin = _perm_ring.take(size[0])
        rand_index = torch.randperm(size[0], out=pin) if pin is not None else torch.randperm(size[0])
        bbx1, bby1, bbx2, bby2 = rand_bbox(size, lam)
        lam = 1 - ((bbx2 - bbx1) * (bby2 - bby1) / (size[-1] * size[-2]))
        return float(lam), rand_index, (int(bbx1), int(bby1), int(bbx2), int(bby2))
    return 1.0, None, None


def _train_loop(args, logger, loss_logger, model, classifier_head, train_loader, val_loader, test_loader, use_cutmix):
    model.eval() if args.freeze_visual else model.train()
    classifier_head.train()
    logit_scale = args.logit_scale
    optimizer = args.optimizer
    scheduler = args.scheduler
    nranks, rank = dp.world(args), getattr(args, 'rank', 0)
    split_batch = nranks > 1 and not getattr(args, 'dp_weak', False)  # every rank sees the whole --bsz batch, takes its rows
    step = FusedTrainStep(model, classifier_head, logit_scale, optimizer, world_size=nranks)
    # global-batch data parallelism with a rank-aware loader (swat_amd/dp.py RankBatchSampler): this rank's loader yields only its
    # rows (+ its CutMix partners), the iteration's CutMix draws were made by the sampler and are popped here
    rank_sampler = getattr(getattr(train_loader, 'loader', train_loader), 'batch_sampler', None)
    if not isinstance(rank_sampler, dp.RankBatchSampler):
        rank_sampler = None
    if rank_sampler is not None:
        rank_sampler.configure((lambda size: cutmix_draw(args, size)) if use_cutmix else None)

    val_loss = -1
    val_acc = -1
    test_acc = -1
    best_val_acc = -1
    best_epoch = -1
    best_iter = -1
    best_model = model
    best_head = classifier_head
    best_logit_scale = logit_scale
    best_records = {}
    num_iter = 0
    for epoch in range(1, args.epochs + 1):
        # identical shuffles and CutMix draws on every rank for this epoch: every rank continues from rank 0's generator states.  Once
        # before the loop is not enough — with --early_stop rank 0 alone validates after every epoch, and creating a DataLoader
        # iterator draws its base seed from torch's global generator even with shuffle=False (ref: the 1-process job makes exactly
        # those draws, utils/training.py:1500-1512, so rank 0's stream IS the reference's)
        dp.sync_rng(args)
        for inputs, labels, text, source in train_loader:
            num_iter += 1
            images = inputs.to(args.device, non_blocking=True)
            labels = labels.to(args.device, non_blocking=True).long()
            if rank_sampler is not None:
                lam, rand_index, box = rank_sampler.pop_draw()
                b = rank_sampler.b
                if rand_index is not None:   # rows b .. 2b - 1 are the partners of rows 0 .. b - 1
                    step(images, labels[:b], labels[b:2 * b], lam, perm=torch.arange(b, 2 * b, device=args.device), box=box, batch=b)
                else:
                    step(images, labels, labels, 1.0)
                scheduler.step()
                continue
            if use_cutmix:
                lam, rand_index, box = cutmix_draw(args, images.size())
            else:
                lam, rand_index, box = 1.0, None, None
            if rand_index is not None:
                rand_index = to_device(rand_index, args.device)
            if split_batch:
                loc, ya, yb, perm_l, b = dp.shard_batch(images, labels, rand_index, rank, nranks)
                step(loc, ya, yb, lam, perm=perm_l, box=box, batch=b)
            elif rand_index is not None:
                step(images, labels, labels[rand_index], lam, perm=rand_index, box=box)
            else:
                step(images, labels, labels, 1.0)
            scheduler.step()  # update learning rate for each iteration

        train_loss_sum = step.pop_loss_sum()  # the only host sync of the epoch
        train_loss_sum = dp.mean_over_ranks(args, train_loss_sum)  # data-parallel: one scalar all-reduce per epoch

        if args.early_stop or epoch == args.epochs:
            if dp.is_main(args):
                val_acc, val_loss, confusion_matrix = validate(args, data_loader=val_loader, model=model, logger=logger,
                                                               loss=args.loss, logit_scale=logit_scale,
                                                               classifier_head=classifier_head, show_confusion_matrix=True,
                                                               dataset=args.dataset, output_dir=args.output_dir,
                                                               device=args.device, pre_extracted=args.pre_extracted)
            # ranks > 0 take the same branches on rank 0's numbers (the models are replicas)
            val_acc, val_loss, confusion_matrix = dp.bcast_object(args, (val_acc, val_loss, confusion_matrix if dp.is_main(args) else None))
            scores = calculate_scores(confusion_matrix)

        if (args.early_stop or epoch == args.epochs) and val_acc >= best_val_acc:
            best_val_acc = val_acc
            best_logit_scale = copy.deepcopy(logit_scale)
            best_epoch = epoch
            best_iter = num_iter
            best_scores = copy.deepcopy(scores)
            best_confusion_matrix = copy.deepcopy(confusion_matrix)
            best_head = copy.deepcopy(classifier_head)
            best_model = copy.deepcopy(model)
            best_records['best_val_acc'] = best_val_acc
            best_records['best_logit_scale'] = best_logit_scale
            best_records['best_epoch'] = best_epoch
            best_records['best_iter'] = best_iter
            best_records['best_scores'] = best_scores
            best_records['best_confusion_matrix'] = best_confusion_matrix

        if args.early_stop or epoch == args.epochs:
            if dp.is_main(args):
                test_acc, _, _ = validate(args, data_loader=test_loader, model=model, logger=logger, loss=args.loss,
                                          logit_scale=logit_scale, classifier_head=classifier_head, dataset=args.dataset,
                                          output_dir=args.output_dir, device=args.device, pre_extracted=args.pre_extracted)
            test_acc = dp.bcast_object(args, test_acc)

        train_loss_avg = train_loss_sum / len(train_loader)
        loss_logger.write(f'{epoch},{num_iter},{round(train_loss_avg, 6)},{round(val_loss, 6)},{round(val_acc, 6)},{round(test_acc, 6)}\n')
        loss_logger.flush()
        from . import jpeg as _jpeg   # (the default input pipeline decodes on host threads + the GPU: say what went where)
        dec = (f", decoded on the GPU path: {_jpeg.stats['device']}, by Pillow (other formats): {_jpeg.stats['host_other_format']}"
               if _jpeg.stats['device'] or _jpeg.stats['host_other_format'] else "")
        logger.info(f"Epoch {int(epoch)}, Iter {num_iter}, Trn Loss: {round(train_loss_avg, 6)}, Val Loss: {round(val_loss, 6)}, "
                    f"Val Acc: {round(val_acc, 3)}, Test Acc: {round(test_acc, 3)}{dec}")
        if logger.handlers:
            logger.handlers[0].flush()

        if args.save_ckpt and dp.is_main(args) and (num_iter % args.save_freq == 0 or epoch == args.epochs):
            model_path = save_model_ckpt(args, best_records, model, classifier_head, optimizer, scheduler, logit_scale,
                                         val_acc, epoch, num_iter)
            logger.info(f'Model ckpt saved to: {model_path}')

        if epoch == args.stop_epochs:
            break

    logger.info('Training done.')
    logger.info(f'Best val Acc: {round(best_val_acc, 3)} at epoch {best_epoch}, iter {best_iter}')
    return best_model, best_head, best_records, best_logit_scale


def train_ce(args, logger, loss_logger, model, classifier_head, train_loader, val_loader, test_loader, reload_model=False):
    """Cross-entropy finetuning of the visual encoder + head (ref :296-412)."""
    logger.info("Start standard finetuning ......")
    if reload_model:
        load_model(args, logger, model, None, classifier_head)
    return _train_loop(args, logger, loss_logger, model, classifier_head, train_loader, val_loader, test_loader, False)


def train_cutmix(args, logger, loss_logger, model, classifier_head, train_loader, val_loader, test_loader):
    """CutMix finetuning (ref :1422-1556)."""
    logger.info("Start Training cutmix ......")
    return _train_loop(args, logger, loss_logger, model, classifier_head, train_loader, val_loader, test_loader, True)


def train_probing(args, logger, loss_logger, model, classifier_head, tokenized_text_prompts, preprocess,
                  train_loader, val_loader, test_loader, reload_model=False):
    """Linear probing on (pre-extracted) features with the visual encoder frozen (ref :47-161).
    -> (best_model, best_head, best_records, best_logit_scale, val_loader, test_loader)"""
    from .dataloader import extract_dataloader, extract_train_dataloader
    if reload_model:
        load_model(args, logger, model, None, classifier_head)
        root = f'{args.dataset_root}/pre_extracted/{args.dataset}_{args.model_cfg}_{args.seed}'
        train_loader = extract_train_dataloader(args, model, args.train_split, f'{root}_train_features_new.pth', preprocess,
                                                tokenized_text_prompts, bsz=args.bsz)
        val_loader = extract_dataloader(args, model, args.val_split, f'{root}_val_features_new.pth', preprocess, tokenized_text_prompts)
        test_loader = extract_dataloader(args, model, args.test_split, f'{root}_test_features_new.pth', preprocess, tokenized_text_prompts)
        args.pre_extracted = True
    logger.info("Start Training (linear probing) ......")
    logit_scale = args.logit_scale
    optimizer, scheduler = args.optimizer, args.scheduler
    model.eval()
    classifier_head.train()
    step = FusedTrainStep(model, classifier_head, logit_scale, optimizer, world_size=1)  # rank 0 alone (swat_amd/dp.py)
    best_records = {}
    best_val_acc = -1
    best_model, best_head, best_logit_scale = model, classifier_head, logit_scale
    num_iter = 0
    for epoch in range(1, args.epochs + 1):
        for inputs, labels, text, source in train_loader:
            num_iter += 1
            x = inputs.to(args.device)
            labels = labels.to(args.device).long()
            if not args.pre_extracted:
                from .features import l2_normalize
                with torch.no_grad():
                    x = l2_normalize(model.visual.forward_raw(x))
            step(x, labels, labels, 1.0)
            scheduler.step()
        train_loss_sum = step.pop_loss_sum()
        val_acc, val_loss, confusion_matrix = validate(args, data_loader=val_loader, model=model, logger=logger, loss=args.loss,
                                                       logit_scale=logit_scale, classifier_head=classifier_head,
                                                       show_confusion_matrix=True, dataset=args.dataset,
                                                       output_dir=args.output_dir, device=args.device,
                                                       pre_extracted=args.pre_extracted)
        scores = calculate_scores(confusion_matrix)
        if (args.early_stop or epoch == args.epochs) and val_acc >= best_val_acc:
            best_val_acc = val_acc
            best_logit_scale = copy.deepcopy(logit_scale)
            best_head = copy.deepcopy(classifier_head)
            best_model = copy.deepcopy(model)
            best_records.update(best_val_acc=best_val_acc, best_logit_scale=best_logit_scale, best_epoch=epoch,
                                best_iter=num_iter, best_scores=copy.deepcopy(scores),
                                best_confusion_matrix=copy.deepcopy(confusion_matrix))
        test_acc, _, _ = validate(args, data_loader=test_loader, model=model, logger=logger, loss=args.loss,
                                  logit_scale=logit_scale, classifier_head=classifier_head, dataset=args.dataset,
                                  output_dir=args.output_dir, device=args.device, pre_extracted=args.pre_extracted)
        train_loss_avg = train_loss_sum / len(train_loader)
        loss_logger.write(f'{epoch},{num_iter},{round(train_loss_avg, 6)},{round(val_loss, 6)},{round(val_acc, 6)},{round(test_acc, 6)}\n')
        loss_logger.flush()
        from . import jpeg as _jpeg   # (the default input pipeline decodes on host threads + the GPU: say what went where)
        dec = (f", decoded on the GPU path: {_jpeg.stats['device']}, by Pillow (other formats): {_jpeg.stats['host_other_format']}"
               if _jpeg.stats['device'] or _jpeg.stats['host_other_format'] else "")
        logger.info(f"Epoch {int(epoch)}, Iter {num_iter}, Trn Loss: {round(train_loss_avg, 6)}, Val Loss: {round(val_loss, 6)}, "
                    f"Val Acc: {round(val_acc, 3)}, Test Acc: {round(test_acc, 3)}{dec}")
        if args.save_ckpt and (num_iter % args.save_freq == 0 or epoch == args.epochs):
            save_model_ckpt(args, best_records, model, classifier_head, optimizer, scheduler, logit_scale, val_acc, epoch, num_iter)
    logger.info('Probing done.')
    return best_model, best_head, best_records, best_logit_scale, val_loader, test_loader
