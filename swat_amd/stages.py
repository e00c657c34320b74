"""Post-training evaluation fan-out and the stage drivers of SWAT's main.py.

ref: run_tau_normalization main.py:27-73, ensemble_model :76-98, run_wsft_alpha :132-195, run_stage1_finetuning :198-398,
     run_stage2_probing :402-474, run_stage2_FSFT :478-540.  Weight-space arithmetic is host-driven torch on the arena's
     views (once per run); every forward pass behind it goes through the HIP tower / head kernels.
"""
import copy
import os
import time

import numpy as np
import torch

from . import dp
from .dataloader import extract_dataloader, extract_train_dataloader, set_dataloaders
from .losses import set_loss
from .models import MyLinear, save_best_model, save_test_scores
from .optimizers import set_optimizer, set_params
from .testing import calculate_scores, load_model, validate
from .training import run_zeroshot, train_ce, train_cutmix, train_probing


def run_tau_normalization(args, best_head, best_model, val_loader, test_loader, logit_scale, logger):
    """W /= ||W||^tau row-wise, tau in {0, .2, ..., 2}, pick the best on val (ref main.py:27-73)."""
    best_tau_head = copy.deepcopy(best_head)
    best_tau, best_tau_val_acc, best_tau_test_acc = 0.0, 0.0, 0.0
    if args.tau_norm:
        logger.info("Check Tau Normalization ......")
        for tau in np.arange(0.0, 2.2, 0.2).tolist():
            tau_head = copy.deepcopy(best_head)
            tau_head.linear.weight.data /= torch.pow(tau_head.linear.weight.data.norm(dim=-1, keepdim=True), tau)
            tau_val_acc, _, _ = validate(args, data_loader=val_loader, model=best_model, logger=logger, loss=args.loss,
                                         logit_scale=logit_scale, classifier_head=tau_head, dataset=args.dataset,
                                         output_dir=args.output_dir, device=args.device, pre_extracted=True)
            tau_test_acc, _, _ = validate(args, data_loader=test_loader, model=best_model, logger=logger, loss=args.loss,
                                          logit_scale=logit_scale, show_confusion_matrix=True, classifier_head=tau_head,
                                          dataset=args.dataset, output_dir=args.output_dir, device=args.device, pre_extracted=True)
            logger.info(f"Tau: {round(tau, 2)}, Val Acc: {round(tau_val_acc, 3)}, Test Acc: {round(tau_test_acc, 3)}")
            if tau_val_acc > best_tau_val_acc:
                best_tau, best_tau_val_acc, best_tau_test_acc = tau, tau_val_acc, tau_test_acc
                best_tau_head = copy.deepcopy(tau_head)
        logger.info(f"+++++ Best Tau: {round(best_tau, 1)}, Val Acc: {round(best_tau_val_acc, 3)}, Test Acc: {round(best_tau_test_acc, 3)}")
    return best_tau_head, best_tau, best_tau_test_acc


def ensemble_model(best_model, zeroshot_model, alpha):
    """WiSE-FT: theta = (1 - alpha) theta_0 + alpha theta_1 over every state-dict tensor (ref main.py:76-98)."""
    wsft_model = copy.deepcopy(best_model)
    theta_0, theta_1 = zeroshot_model.state_dict(), best_model.state_dict()
    assert set(theta_0.keys()) == set(theta_1.keys())
    theta = {key: (1.0 - alpha) * theta_0[key] + alpha * theta_1[key] for key in theta_0.keys()}
    wsft_model.load_state_dict(theta)
    return wsft_model


def run_wsft_alpha(args, best_model, best_head, val_loader, test_loader, zeroshot_model, zeroshot_head, logit_scale, logger):
    """ref main.py:132-195 (alpha = 0.5 only, as the reference runs it)."""
    logger.info("Checking WSFT ......")
    learned = best_head.linear.weight.data.to(args.device)
    zeroshot = zeroshot_head.linear.weight.data.to(args.device)
    best_alpha, best_wsft_test_acc, best_wsft_val_acc = 0.0, 0.0, 0.0
    best_wsft_head, best_wsft_model = best_head, best_model
    for alpha in [0.5]:
        wsft_head = MyLinear(weights=alpha * learned + (1.0 - alpha) * zeroshot).to(args.device)
        wsft_model = best_model if args.freeze_visual else ensemble_model(best_model, zeroshot_model, alpha)
        kw = dict(model=wsft_model, classifier_head=wsft_head, logger=logger, loss=args.loss, logit_scale=logit_scale,
                  dataset=args.dataset, output_dir=args.output_dir, device=args.device, pre_extracted=args.pre_extracted)
        wsft_val_acc, _, _ = validate(args, data_loader=val_loader, **kw)
        wsft_test_acc, _, _ = validate(args, data_loader=test_loader, **kw)
        logger.info(f"Alpha:{round(alpha, 3)}, Val Acc: {round(wsft_val_acc, 3)}, Test Acc: {round(wsft_test_acc, 3)}")
        if wsft_val_acc > best_wsft_val_acc:
            best_wsft_val_acc, best_wsft_test_acc, best_alpha = wsft_val_acc, wsft_test_acc, alpha
            best_wsft_head, best_wsft_model = copy.deepcopy(wsft_head), copy.deepcopy(wsft_model)
    logger.info(f"+++++ Best Alpha: {round(best_alpha, 2)}, Val Acc: {round(best_wsft_val_acc, 3)}, Test Acc: {round(best_wsft_test_acc, 3)}")
    return best_wsft_model, best_wsft_head, best_wsft_test_acc


def _prepare(args, model, classifier_head, train_loader, logger):
    loss = set_loss(args)
    params, logit_scale = set_params(args, model, classifier_head, logger)
    optimizer, scheduler, total_iter = set_optimizer(args, params, train_loader)
    args.loss, args.logit_scale, args.optimizer, args.scheduler = loss, logit_scale, optimizer, scheduler
    return logit_scale


def run_stage1_finetuning(args, logger, loss_logger, model, classifier_head, zeroshot_model, zeroshot_head, preprocess,
                          tokenized_text_prompts):
    """ref main.py:198-398 -> (test_acc, best_model_path, test_loader, wsft_test_acc)."""
    train_loader, val_loader, test_loader = set_dataloaders(args, model, tokenized_text_prompts, preprocess, logger)
    test_loader_copy = test_loader
    if args.method == 'zeroshot':
        set_loss(args)
        _, logit_scale = set_params(args, model, classifier_head, logger)
    else:
        logit_scale = _prepare(args, model, classifier_head, train_loader, logger)
    stage1_method = args.method
    if args.model_path:
        load_model(args, logger, model, test_loader, classifier_head)
    if args.check_zeroshot or args.method == 'zeroshot':
        logger.info("Check Zero-shot Acc ......")
        zs_test_acc = run_zeroshot(args, test_loader, model, logger, args.loss, logit_scale, classifier_head) if dp.is_main(args) else -1
        zs_test_acc = dp.bcast_object(args, zs_test_acc)
    if args.zeroshot_only or args.method == 'zeroshot':
        summary = f'{args.dataset},{stage1_method},{args.data_source},{args.cls_init},{args.shots},{args.seed},{args.retrieval_split},{round(zs_test_acc, 1)}'
        logger.info(summary)
        if dp.is_main(args):
            print(summary)
        raise SystemExit(0)
    reload_model = bool(args.model_path)
    if args.skip_stage1:
        logger.info("Skip stage 1 finetuning.")
        return -1, None, test_loader_copy, -1
    if args.method == 'finetune':
        out = train_ce(args, logger, loss_logger, model, classifier_head, train_loader, val_loader, test_loader, reload_model)
    elif args.method == 'cutmix':
        out = train_cutmix(args, logger, loss_logger, model, classifier_head, train_loader, val_loader, test_loader)
    elif args.method == 'probing':
        out = train_probing(args, logger, loss_logger, model, classifier_head, tokenized_text_prompts, preprocess,
                            train_loader, val_loader, test_loader, reload_model)[:4]
    else:
        raise NotImplementedError(f"Method {args.method}: out of scope of the MI355X hot path (SURVEY §2 row 3b); "
                                  "finetune / cutmix / probing / zeroshot are implemented.")
    best_model, best_head, best_records, best_logit_scale = out
    if not dp.is_main(args):  # data-parallel ranks > 0: the evaluation fan-out and the checkpoint are rank 0's
        test_acc, best_model_path, wsft_test_acc = dp.bcast_object(args, None)
        return test_acc, best_model_path, test_loader_copy, wsft_test_acc
    wsft_backbone, wsft_head, wsft_test_acc = None, None, -1
    if not args.no_wsft:
        wsft_backbone, wsft_head, wsft_test_acc = run_wsft_alpha(args, best_model, best_head, val_loader, test_loader,
                                                                 zeroshot_model, zeroshot_head, best_logit_scale, logger)
    if args.method not in ("probing", "REAL-Linear"):
        root = f'{args.dataset_root}/pre_extracted/{args.dataset}_{args.model_cfg}_{args.shots}_{args.seed}'
        val_loader = extract_dataloader(args, best_model, args.val_split, f'{root}_val_features_new.pth', preprocess, tokenized_text_prompts, seed=2)
        test_loader = extract_dataloader(args, best_model, args.test_split, f'{root}_test_features_new.pth', preprocess, tokenized_text_prompts, seed=3)
        logger.info('Extracted val, test dataloader for fast testing after training.')
    test_acc, test_loss, test_cm = validate(args, data_loader=test_loader, model=best_model, classifier_head=best_head, logger=logger,
                                            loss=args.loss, logit_scale=best_logit_scale, show_confusion_matrix=True,
                                            dataset=args.dataset, output_dir=args.output_dir, device=args.device, pre_extracted=True)
    logger.info(f"+++++ Test Acc: {round(test_acc, 3)}")
    save_test_scores(calculate_scores(test_cm), test_cm, args.output_dir, 'test')
    best_tau_head, best_tau, best_tau_test_acc = None, -1, -1
    if not args.no_tau:
        best_tau_head, best_tau, best_tau_test_acc = run_tau_normalization(args, best_head, best_model, val_loader, test_loader,
                                                                           best_logit_scale, logger)
    logger.info(f"best_logit_scale: {round(best_logit_scale.item(), 8)}")
    best_model_path = save_best_model(args, best_records, best_model, best_head, best_logit_scale, test_acc, best_tau,
                                      best_tau_test_acc, wsft_test_acc, best_tau_head, wsft_backbone, wsft_head, stage=1)
    logger.info(f'Stage 1 Best Model saved to: {best_model_path}')
    dp.bcast_object(args, (test_acc, best_model_path, wsft_test_acc))  # releases the other ranks once the file exists
    return test_acc, best_model_path, test_loader_copy, wsft_test_acc


def run_stage2_probing(args, logger, loss_logger, model, classifier_head, stage1_best_model_path, test_loader,
                       tokenized_text_prompts, preprocess):
    """Classifier retraining on few-shot features of the stage-1 backbone (ref main.py:402-474)."""
    if not dp.is_main(args):  # pre-extracted features, a head-only step: rank 0 alone; the others wait for its result
        return dp.bcast_object(args, None)
    logger.info("Run stage 2 classifier retraining ......")
    args.model_path = stage1_best_model_path
    load_model(args, logger, model, None, classifier_head)
    root = f'{args.dataset_root}/pre_extracted/{args.dataset}_{args.model_cfg}_{args.shots}_{args.seed}'
    train_loader = extract_train_dataloader(args, model, args.fewshot_data, f'{root}_fewshot_features_new.pth', preprocess,
                                            tokenized_text_prompts, args.bsz)
    val_loader = train_loader
    test_loader = extract_dataloader(args, model, args.test_split, f'{root}_test_features_new.pth', preprocess, tokenized_text_prompts, seed=3)
    args.method, args.pre_extracted, args.epochs, args.early_stop, args.freeze_visual = 'probing', True, 10, False, True
    logit_scale = _prepare(args, model, classifier_head, train_loader, logger)
    best_model, best_head, best_records, _, _, _ = train_probing(args, logger, loss_logger, model, classifier_head,
                                                                 tokenized_text_prompts, preprocess, train_loader, val_loader,
                                                                 test_loader, reload_model=False)
    test_acc, _, test_cm = validate(args, data_loader=test_loader, model=best_model, classifier_head=best_head, logger=logger,
                                    loss=args.loss, logit_scale=args.logit_scale, show_confusion_matrix=True, dataset=args.dataset,
                                    output_dir=args.output_dir, device=args.device, pre_extracted=True)
    logger.info(f"+++++ stage 2 Test Acc: {round(test_acc, 3)}")
    save_test_scores(calculate_scores(test_cm), test_cm, args.output_dir, 'test', stage=2)
    path = save_best_model(args, best_records, best_model, best_head, logit_scale, test_acc, None, -1, -1, None, None, None, stage=2)
    for suffix in ('_fewshot_features_new.pth', '_test_features_new.pth'):
        if os.path.exists(root + suffix):
            os.remove(root + suffix)
    dp.bcast_object(args, (test_acc, path))
    return test_acc, path


def run_stage2_FSFT(args, logger, loss_logger, model, classifier_head, stage1_best_model_path, test_loader, preprocess,
                    tokenized_text_prompts):
    """SWAT+: 10 epochs of full finetuning on the few-shot data only, from the stage-1 model (ref main.py:478-540)."""
    logger.info("Run stage 2 few-shot finetuning ......")
    args.model_path = stage1_best_model_path
    load_model(args, logger, model, None, classifier_head)
    args.method, args.pre_extracted, args.epochs, args.early_stop, args.freeze_visual = 'finetune', False, 10, False, False
    args.save_ckpt = False  # ref main.py:486
    args.train_split = args.fewshot_data
    train_loader, val_loader, test_loader = set_dataloaders(args, model, tokenized_text_prompts, preprocess, logger)
    logit_scale = _prepare(args, model, classifier_head, train_loader, logger)
    best_model, best_head, best_records, _ = train_ce(args, logger, loss_logger, model, classifier_head,
                                                      train_loader, val_loader, test_loader, False)
    if not dp.is_main(args):
        return dp.bcast_object(args, None)
    test_acc, _, test_cm = validate(args, data_loader=test_loader, model=best_model, classifier_head=best_head, logger=logger,
                                    loss=args.loss, logit_scale=args.logit_scale, show_confusion_matrix=True, dataset=args.dataset,
                                    output_dir=args.output_dir, device=args.device, pre_extracted=False)
    logger.info(f"+++++ stage 2 FSFT Test Acc: {round(test_acc, 3)}")
    # the reference files FSFT under stage 3 (main.py:527-536): stage-2 probing's checkpoint / scores keep their names
    save_test_scores(calculate_scores(test_cm), test_cm, args.output_dir, 'test', stage=3)
    path = save_best_model(args, best_records, best_model, best_head, logit_scale, test_acc, None, -1, -1, None, None, None, stage=3)
    logger.info(f'Stage 2 FSFT Best Model saved to: {path}')
    dp.bcast_object(args, (test_acc, path))
    return test_acc, path
