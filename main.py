#!/usr/bin/env python3
"""SWAT's command line on the MI355X-native kernels: stage-1 finetune -> WiSE-FT -> tau-norm -> save -> stage-2 probing
-> stage-2 FSFT -> CSV summary line (same flags and summary format as the reference's main.py:544-606).

    python main.py --dataset semi-aves --method cutmix --data_source fewshot+retrieved --cls_init REAL-Prompt --shots 4 \
                   --seed 1 --epochs 50 --bsz 32 --retrieval_split T2T500+T2I0.25.txt
    python main.py --synthetic 512 --num_classes 20 --method cutmix --cls_init random --epochs 2 --skip_stage2   # no dataset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 main.py --dataset semi-aves ...   # data-parallel
        (one process per GPU over RCCL; the --bsz batch is split over the ranks, --dp_weak gives every rank its own --bsz)
"""
import copy
import time

from swat_amd import dp
from swat_amd.logger import set_logger
from swat_amd.models import set_classifier, set_model
from swat_amd.parser import parse_args
from swat_amd.prompt import set_prompt
from swat_amd.stages import run_stage1_finetuning, run_stage2_FSFT, run_stage2_probing
from swat_amd.training import set_training_seed


def main(argv=None):
    program_start = time.time()
    args = parse_args(argv)
    dp.init_from_env(args)  # under torch.distributed.run: process group + device BEFORE anything touches the GPU
    if args.synthetic:
        args.dataset, args.cls_init = 'synthetic', 'random'
        args.dataset_root = 'data/synthetic'
        from swat_amd import models
        models.NUM_CLASSES_DICT['synthetic'] = args.num_classes
    logger, loss_logger = set_logger(args)
    args.logger, args.loss_logger = logger, loss_logger
    set_training_seed(args)
    model, preprocess, tokenizer = set_model(args, logger)
    zeroshot_model = copy.deepcopy(model)
    prompt_tensors, text_prompts, tokenized_text_prompts, prompt_tensors_dict = set_prompt(args, model, tokenizer, logger)
    classifier_head = set_classifier(args, prompt_tensors, logger)
    zeroshot_head = copy.deepcopy(classifier_head)
    stage1_acc, stage1_path, test_loader, wsft_test_acc = run_stage1_finetuning(
        args, logger, loss_logger, model, classifier_head, zeroshot_model, zeroshot_head, preprocess, tokenized_text_prompts)
    stage1_method = args.method
    if args.skip_stage1:
        stage1_path = args.stage1_model_path
    stage2_lp_acc, stage2_fsft_acc = -1, -1
    if not args.skip_stage2:
        stage2_lp_acc, _ = run_stage2_probing(args, logger, loss_logger, model, classifier_head, stage1_path, test_loader,
                                              tokenized_text_prompts, preprocess)
        stage2_fsft_acc, _ = run_stage2_FSFT(args, logger, loss_logger, model, classifier_head, stage1_path, test_loader,
                                             preprocess, tokenized_text_prompts)
    else:
        logger.info("Skip stage 2.")
    loss_logger.close()
    logger.info(f"Total time: {round((time.time() - program_start) / 60, 1)} mins.")
    summary = (f'{args.dataset},{stage1_method},{args.data_source},{args.cls_init},{args.shots},{args.seed},{args.retrieval_split},'
               f'{round(stage1_acc, 1)},{round(wsft_test_acc, 1)},{round(stage2_lp_acc, 1)},{round(stage2_fsft_acc, 1)}')
    logger.info(summary)
    if dp.is_main(args):
        print(summary)
    if dp.world(args) > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return summary


if __name__ == '__main__':
    main()
