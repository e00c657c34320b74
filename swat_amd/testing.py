"""Evaluation: validate / calculate_scores / load_model (mirror of the reference's testing.py).

ref: load_model :76-131, calculate_scores :134-186, validate :403-486.
Forward-only HIP path: encode_image -> L2-normalise -> head -> scaled CE (swat_head_loss with want_grad = 0);
argmax, the accuracy count and the confusion matrix accumulate on the device (swat_argmax_confusion); the
reference does them on the host per batch (.cpu() + torchmetrics, :449-458).
"""
import numpy as np
import torch

from . import _lib, ops


def load_model(args, logger, model, test_loader=None, classifier_head=None):
    logger.info(f'Loading model from: {args.model_path}')
    ckpt = torch.load(args.model_path, map_location='cpu', weights_only=False)
    if 'clip' in ckpt:
        model.load_state_dict(ckpt['clip'])
        classifier_head.load_state_dict(ckpt['head'])
        logger.info(f'ckpt[test_acc]: {ckpt["test_acc"]}')
        logger.info(f'ckpt[wsft_test_acc]: {ckpt["wsft_test_acc"]}')
    elif 'model' in ckpt:
        model.load_state_dict(ckpt['model'], strict=False)
    else:
        classifier_head.load_state_dict(ckpt['best_tau_head'])
    del ckpt
    if test_loader is not None:
        model_test_acc, _, _ = validate(args, data_loader=test_loader, model=model, logger=logger, loss=args.loss,
                                        logit_scale=args.logit_scale, classifier_head=classifier_head, dataset=args.dataset,
                                        device=args.device, pre_extracted=args.pre_extracted)
        logger.info(f"Loaded Model Test Acc: {round(model_test_acc, 3)}")


def calculate_scores(confusion_matrix):
    """ref :134-186 — rows of the confusion matrix are the TRUE class."""
    scores = {}
    num_class = confusion_matrix.shape[0]
    scores['acc'] = np.trace(confusion_matrix) / np.sum(confusion_matrix)
    class_accuracy = np.diag(confusion_matrix) / np.sum(confusion_matrix, axis=1)
    scores['avg_class_accuracy'] = class_accuracy.mean() * 100
    recall, precision, f1_score = dict(), dict(), dict()
    for i in range(num_class):
        tp = confusion_matrix[i, i]
        fn = np.sum(confusion_matrix[i, :]) - tp
        fp = np.sum(confusion_matrix[:, i]) - tp
        recall[i] = 0.0 if tp + fn == 0 else tp / (tp + fn)
        precision[i] = 0.0 if tp + fp == 0 else tp / (tp + fp)
        f1_score[i] = 0 if tp == 0 else 2 * (precision[i] * recall[i]) / (precision[i] + recall[i])
    scores['per_class_recall'] = recall
    scores['per_class_precision'] = precision
    scores['per_class_f1score'] = f1_score
    return scores


_WS = {}  # (device, B, E, C) -> head_loss workspace: validate runs ~30 times per job on the same batch shapes


def _head_ws(dev, B, E, C):
    key = (str(dev), B, E, C)
    ws = _WS.get(key)
    if ws is None:
        if len(_WS) > 16:
            _WS.clear()
        ws = _WS[key] = torch.empty(_lib.lib().swat_head_loss_workspace_bytes(B, E, C), dtype=torch.uint8, device=dev)
    return ws


def head_eval(feats, W, logit_scale, labels, normalize=True, z=None, loss=None):
    """-> (z [B,C] un-scaled head output, loss device scalar = CE(z * exp(logit_scale), labels))."""
    B, E = feats.shape
    C = W.shape[0]
    dev = feats.device
    ls = logit_scale.detach().reshape(-1)[:1].to(dev, torch.float32).contiguous()
    if z is None:
        z = torch.empty((B, C), dtype=torch.float32, device=dev)
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=dev)
    Wd = W.detach().contiguous()
    if normalize:
        ws = _head_ws(dev, B, E, C)
        rc = _lib.lib().swat_head_loss(_lib.ptr(feats), _lib.ptr(Wd), _lib.ptr(ls), _lib.ptr(labels), _lib.ptr(labels), 1.0,
                                       _lib.ptr(ws), _lib.ptr(loss), None, None, None, _lib.ptr(z), None, None, B, E, C, 0, 1,
                                       _lib.stream())
        _lib.check(rc, "swat_head_loss")
    else:
        ops.sgemm_small(feats, Wd, B, C, E, E, 1, 1, E, out=z)
        l2, _, _, _ = ops.ce_mixed(z, ls, labels, labels, 1.0, want_grad=False)
        loss.copy_(l2.reshape(-1)[:1])
    return z, loss


@_lib.retry_op_by_op
def validate(args, data_loader, model, logger, loss, logit_scale, classifier_head=None, show_confusion_matrix=False,
             device='cuda', dataset='semi-aves', output_dir='output', predict_labels=False, predict_split='u_train',
             pre_extracted=False):
    """ref :403-486 -> (acc %, mean of batch-mean CE losses, confusion matrix [C,C] int64 (rows = true) | None)."""
    model.eval()
    if classifier_head:
        classifier_head.eval()
    else:
        raise NotImplementedError("validate without a classifier head (model(inputs, texts)) is out of scope")
    # f.4: argmax, accuracy count and the confusion matrix accumulate ON THE DEVICE (swat_argmax_confusion; the
    # reference moves every batch's logits to the host, testing.py:449-458); one host synchronisation per validate call.
    val_count = 0
    num_classes = classifier_head.num_classes
    W = classifier_head.linear.weight
    dev = W.device
    L = _lib.lib()
    confusion_d = torch.zeros((num_classes, num_classes), dtype=torch.int64, device=dev)
    correct_d = torch.zeros(1, dtype=torch.int64, device=dev)
    n_batches = len(data_loader) if hasattr(data_loader, '__len__') else None
    losses_d = torch.zeros(max(1, n_batches or 1), dtype=torch.float32, device=dev)
    extra_losses, preds_d, maxl_d = [], [], []
    with torch.no_grad():
        for i, val_data in enumerate(data_loader):
            inputs, labels, texts, source = val_data
            inputs = inputs.to(device)
            labels = labels.long().to(device).contiguous()
            B = labels.numel()
            if n_batches is not None and i < n_batches:
                loss_slot = losses_d[i:i + 1]
            else:
                loss_slot = torch.empty(1, dtype=torch.float32, device=dev)
                extra_losses.append(loss_slot)
            if not pre_extracted:
                feats = model.visual.forward_raw(inputs)
                z, _ = head_eval(feats, W, logit_scale, labels, normalize=True, loss=loss_slot)
            else:
                z, _ = head_eval(inputs.float().contiguous(), W, logit_scale, labels, normalize=False, loss=loss_slot)
            pred = torch.empty(B, dtype=torch.int64, device=dev) if predict_labels else None
            maxl = torch.empty(B, dtype=torch.float32, device=dev) if predict_labels else None
            _lib.check(L.swat_argmax_confusion(_lib.ptr(z), _lib.ptr(labels), B, num_classes, _lib.ptr(pred), _lib.ptr(maxl),
                                               _lib.ptr(confusion_d) if show_confusion_matrix else None, _lib.ptr(correct_d),
                                               _lib.stream()), "swat_argmax_confusion")
            if predict_labels:
                preds_d.append(pred)
                maxl_d.append(maxl)
            val_count += B
    n_seen = i + 1
    all_losses = torch.cat([losses_d[:min(n_seen, losses_d.numel())] if n_batches is not None else losses_d[:0]] + extra_losses)
    host = torch.cat([all_losses.double(), correct_d.double()]).cpu()   # the one synchronisation of the call
    val_acc = (float(host[-1].item()) / val_count) * 100
    val_loss = np.mean(host[:-1].tolist())   # fp32 batch means averaged in double, as np.mean of the reference's .item() list
    confusion = confusion_d.cpu().numpy() if show_confusion_matrix else None
    predicted_labels = [p.cpu().numpy() for p in preds_d]
    max_logits = [m.cpu().numpy() for m in maxl_d]
    if predict_labels:
        predicted = np.concatenate(predicted_labels).tolist()
        maxl = np.concatenate(max_logits).tolist()
        predicted_label_file = f'{output_dir}/{dataset}_{predict_split}_predicted_labels.txt'
        with open(predicted_label_file, 'w') as f:
            for item, logit in zip(predicted, maxl):
                f.write("%s %s\n" % (item, logit))
        logger.info(f'Predicted labels saved to: {predicted_label_file}')
    if show_confusion_matrix:
        return val_acc, val_loss, confusion
    return val_acc, val_loss, None
