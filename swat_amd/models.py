"""Classifier head, model factory wrapper, checkpoint writers (mirror of the reference's utils/models.py).

ref: set_model :11-18, set_classifier :22-43, MyLinear :47-72, save_model_ckpt :85-107, save_best_model :110-146,
     save_test_scores :149-159.
"""
import json
import pickle

import torch
from torch import nn

from . import features
from .engine import get_engine

NUM_CLASSES_DICT = {'semi-aves': 200, 'flowers102': 102, 'fgvc-aircraft': 100, 'eurosat': 10, 'dtd': 47,
                    'oxford_pets': 37, 'food101': 101, 'stanford_cars': 196, 'imagenet': 1000}


def set_model(args, logger):
    model, preprocess, tokenizer = get_engine(model_cfg=args.model_cfg, device=args.device)
    args.embed_dim = model.visual.output_dim  # width of the random / dataset-cls heads (set_classifier)
    logger.info(f'Loaded model: {args.model_cfg}')
    return model, preprocess, tokenizer


class MyLinear(nn.Module):
    """Bias-free linear head (ref :47-72).  forward() is ordinary torch (used by host-side callers such as tau
    normalisation); the training / validation loops run it through the fused HIP tail (swat_head_loss)."""

    def __init__(self, weights=None, inp_dim=512, num_classes=810, bias=False):
        super().__init__()
        if torch.is_tensor(weights):
            self.linear = nn.Linear(weights.shape[1], weights.shape[0], bias=bias)
            with torch.no_grad():
                self.linear.weight.copy_(weights)
            self.num_classes = weights.shape[0]
        else:
            self.linear = nn.Linear(inp_dim, num_classes, bias=bias)
            self.num_classes = num_classes

    def forward(self, x):
        return self.linear(x)

    def update_weights(self, weights):
        with torch.no_grad():
            self.linear.weight.copy_(weights)


def set_classifier(args, prompt_tensors, logger):
    # The reference hard-codes inp_dim=512 for the random / dataset-cls heads (utils/models.py:26,36), which is the
    # embedding width of its ViT-B towers only; here the width comes from the tower (512 for ViT-B/*, 768 for ViT-L/14:
    # BASELINE config 5), so `--cls_init random` works for every --model_cfg.  Text-initialised heads take it from the
    # prompt tensor in both.
    inp_dim = getattr(args, 'embed_dim', 512)
    if args.method == "dataset-cls":
        classifier_head = MyLinear(inp_dim=inp_dim, num_classes=2, bias=False)
    elif args.cls_init in ('REAL-Prompt', 'REAL-Linear', 'text'):
        weights = features.prompt_sampler(prompt_tensors, sample_by='mean')
        classifier_head = MyLinear(weights=weights)
        logger.info(f'Initialized classifier head with text embedding. weights.shape: {weights.shape}')
    elif args.cls_init == 'random':
        num_class = NUM_CLASSES_DICT[args.dataset]
        classifier_head = MyLinear(inp_dim=inp_dim, num_classes=num_class, bias=False)
        logger.info(f'Initialized classifier head with random weights. Num of classes: {num_class}')
    else:
        raise NotImplementedError(f'Classifier head {args.cls_init} not implemented.')
    classifier_head.to(args.device)
    return classifier_head


def save_model_ckpt(args, best_records, model, classifier_head, optimizer, scheduler, logit_scale,
                    val_acc=-1, epoch=-1, num_iter=-1):
    model_path = f'{args.ckpt_path}/model_bs{args.bsz}_lr-cls{args.lr_classifier}_lr-bkb{args.lr_backbone}_wd{args.wd}_epoch_{epoch}_iter_{num_iter}.pth'
    state = {'best_val_acc': best_records['best_val_acc'], 'best_epoch': best_records['best_epoch'],
             'best_iter': best_records['best_iter'], 'val_acc': val_acc, 'epoch': epoch, 'num_iter': num_iter}
    if not args.freeze_visual:
        state['clip'] = model.state_dict()
    state['head'] = classifier_head.state_dict()
    state['optimizer'] = optimizer.state_dict()
    state['scheduler'] = scheduler.state_dict()
    state['logit_scale'] = logit_scale
    torch.save(state, model_path)
    return model_path


def save_best_model(args, best_records, best_model, best_head, best_logit_scale, test_acc, best_tau, best_tau_test_acc,
                    wsft_test_acc, best_tau_head, wsft_backbone, wsft_head, stage=1):
    best_epoch = best_records['best_epoch']
    model_path = f'{args.output_dir}/stage{stage}_model_best-epoch_{best_epoch}_best.pth'
    with open(f'{args.output_dir}/stage{stage}_val_scores_best.json', 'w') as f:
        json.dump(best_records['best_scores'], f, indent=4)
    with open(f'{args.output_dir}/stage{stage}_val_confusion_matrix_best.pkl', 'wb') as f:
        pickle.dump(best_records['best_confusion_matrix'], f)
    state = {'best_val_acc': best_records['best_val_acc'], 'best_epoch': best_records['best_epoch'],
             'best_iter': best_records['best_iter'], 'clip': best_model.state_dict(), 'head': best_head.state_dict(),
             'logit_scale': best_logit_scale, 'test_acc': round(test_acc, 3), 'best_tau': best_tau,
             'best_tau_test_acc': round(best_tau_test_acc, 3), 'wsft_test_acc': round(wsft_test_acc, 3),
             'best_tau_head': best_tau_head.state_dict() if best_tau_head is not None else None,
             'wsft_head': wsft_head.state_dict() if wsft_head is not None else None,
             'wsft_backbone': wsft_backbone.state_dict() if wsft_backbone is not None else None}
    torch.save(state, model_path)
    return model_path


def save_test_scores(scores, confusion_matrix, output_dir, tag, stage=1):
    with open(f'{output_dir}/stage{stage}_{tag}_scores.json', 'w') as f:
        json.dump(scores, f, indent=4)
    with open(f'{output_dir}/stage{stage}_{tag}_confusion_matrix.pkl', 'wb') as f:
        pickle.dump(confusion_matrix, f)
