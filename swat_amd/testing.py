"""Evaluation: validate / calculate_scores / load_model (mirror of the reference's testing.py).

ref: load_model :76-131, calculate_scores :134-186, validate :403-486.
Forward-only HIP path: encode_image -> L2-normalise -> head -> scaled CE (swat_head_loss with want_grad = 0);
argmax / accuracy / confusion matrix are host-side like the reference's (.cpu() + torchmetrics there).
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


def head_eval(feats, W, logit_scale, labels, normalize=True):
    """-> (z [B,C] un-scaled head output, loss device scalar = CE(z * exp(logit_scale), labels))."""
    B, E = feats.shape
    C = W.shape[0]
    dev = feats.device
    ls = logit_scale.detach().reshape(-1)[:1].to(dev, torch.float32).contiguous()
    z = torch.empty((B, C), dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    Wd = W.detach().contiguous()
    if normalize:
        ws = torch.empty(_lib.lib().swat_head_loss_workspace_bytes(B, E, C), dtype=torch.uint8, device=dev)
        rc = _lib.lib().swat_head_loss(_lib.ptr(feats), _lib.ptr(Wd), _lib.ptr(ls), _lib.ptr(labels), _lib.ptr(labels), 1.0,
                                       _lib.ptr(ws), _lib.ptr(loss), None, None, None, _lib.ptr(z), None, None, B, E, C, 0, 1,
                                       _lib.stream())
        _lib.check(rc, "swat_head_loss")
    else:
        ops.sgemm_small(feats, Wd, B, C, E, E, 1, 1, E, out=z)
        loss, _, _, _ = ops.ce_mixed(z, ls, labels, labels, 1.0, want_grad=False)
    return z, loss


def validate(args, data_loader, model, logger, loss, logit_scale, classifier_head=None, show_confusion_matrix=False,
             device='cuda', dataset='semi-aves', output_dir='output', predict_labels=False, predict_split='u_train',
             pre_extracted=False):
    """ref :403-486 -> (acc %, mean of batch-mean CE losses, confusion matrix [C,C] int64 (rows = true) | None)."""
    model.eval()
    if classifier_head:
        classifier_head.eval()
    else:
        raise NotImplementedError("validate without a classifier head (model(inputs, texts)) is out of scope")
    val_acc = 0
    val_count = 0
    num_classes = classifier_head.num_classes
    confusion = np.zeros((num_classes, num_classes), dtype=np.int64)
    W = classifier_head.linear.weight
    predicted_labels, max_logits, val_loss_batch = [], [], []
    with torch.no_grad():
        for i, val_data in enumerate(data_loader):
            inputs, labels, texts, source = val_data
            inputs = inputs.to(device)
            labels = labels.long().to(device)
            if not pre_extracted:
                feats = model.visual.forward_raw(inputs)
                z, loss_batch = head_eval(feats, W, logit_scale, labels, normalize=True)
            else:
                z, loss_batch = head_eval(inputs.float().contiguous(), W, logit_scale, labels, normalize=False)
            val_loss_batch.append(loss_batch)
            zc = z.cpu()
            labels_c = labels.cpu()
            max_logits.append(torch.max(zc, dim=1).values.numpy())
            pred = torch.argmax(zc, dim=1)
            predicted_labels.append(pred.numpy())
            val_acc += torch.sum(pred == labels_c).item()
            val_count += labels_c.size(0)
            if show_confusion_matrix:
                np.add.at(confusion, (labels_c.numpy(), pred.numpy()), 1)
    val_acc = (val_acc / val_count) * 100
    val_loss = np.mean([v.item() for v in val_loss_batch])
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
