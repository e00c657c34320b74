"""ref: utils/logger.py:7-85 — output dirs, main.log, loss.csv (header Epoch,Iter,Train_loss,Val_loss,Val_acc,Test_acc)."""
import logging
import os

import torch


def get_logger(dir_path, file_name, log_mode='file'):
    logger = logging.getLogger('')
    logger.setLevel(logging.INFO)
    for h in list(logger.handlers):
        logger.removeHandler(h)
    fmt = logging.Formatter('%(asctime)s - %(levelname)s - %(message)s')
    if log_mode in ('file', 'both'):
        fh = logging.FileHandler(os.path.join(dir_path, f'{file_name}.log'), mode='w')
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    if log_mode in ('console', 'both'):
        ch = logging.StreamHandler()
        ch.setFormatter(fmt)
        logger.addHandler(ch)
    return logger


def set_logger(args):
    case_name = f'{args.prefix + "_" if args.prefix else ""}{args.dataset}_{args.method}_{args.data_source}_{args.cls_init}_{args.shots}shots_seed{args.seed}_{args.epochs}eps'
    output_dir = os.path.join(args.folder, case_name)
    os.makedirs(output_dir, exist_ok=True)
    ckpt_path = f'{output_dir}/model_ckpts'
    os.makedirs(ckpt_path, exist_ok=True)
    os.makedirs(f'{args.dataset_root}/pre_extracted/', exist_ok=True)
    if getattr(args, 'rank', 0) != 0:  # data-parallel ranks > 0: same directories, no files (swat_amd/dp.py)
        from .dp import _NullLog
        logger = logging.getLogger('')
        for h in list(logger.handlers):
            logger.removeHandler(h)
        logger.addHandler(logging.NullHandler())
        args.device = "cuda" if torch.cuda.is_available() else "cpu"
        args.output_dir, args.ckpt_path = output_dir, ckpt_path
        return logger, _NullLog()
    logger = get_logger(output_dir, 'main', args.log_mode)
    logger.info('logging started')
    logger.info(f'case_name: {case_name}')
    for arg in sorted(vars(args)):
        logger.info(f'{arg} = {getattr(args, arg)}')
    loss_logger = open(f'{output_dir}/loss.csv', 'w')
    loss_logger.write('Epoch,Iter,Train_loss,Val_loss,Val_acc,Test_acc\n')
    args.device = "cuda" if torch.cuda.is_available() else "cpu"
    logger.info(f"Device: {args.device}")
    if torch.cuda.is_available():
        logger.info(f'Number of GPUs available: {torch.cuda.device_count()}')
    args.output_dir, args.ckpt_path = output_dir, ckpt_path
    return logger, loss_logger
