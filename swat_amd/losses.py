"""ref: utils/losses.py:9-23 — only 'CE' (nn.CrossEntropyLoss, mean reduction) is on the hot path."""
import torch.nn as nn


def set_loss(args):
    if args.loss_name == 'CE':
        loss = nn.CrossEntropyLoss()
    else:
        raise NotImplementedError(f'Loss {args.loss_name}: out of scope (SURVEY §2 row 6); the HIP tail implements CE.')
    args.loss = loss
    return loss
