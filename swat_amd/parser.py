"""argparse CLI of main.py (mirror of the reference's utils/parser.py:7-218: same flag names, defaults and post-processing).

Additions that do not change existing flags: `vitl14_openclip_laion400m` in --model_cfg (BASELINE config 5),
`--synthetic N` / `--num_classes` (dataset-free smoke runs), `--config` (path of config.yml)."""
import argparse
import os

import yaml

from .engine import MODEL_CFG_CHOICES


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def build_parser():
    p = argparse.ArgumentParser(description='Arguments for script.')
    p.add_argument('--log_mode', type=str, default='both', choices=['console', 'file', 'both'])
    p.add_argument('--folder', type=str, default='output')
    p.add_argument('--prefix', type=str, default=None)
    p.add_argument('--model_cfg', type=str, default='vitb32_openclip_laion400m',
                   choices=MODEL_CFG_CHOICES + ['rn50_openclip_openai', 'vitb32_clip', 'vitb16_clip', 'rn50_clip'])
    p.add_argument('--model_path', default=None, type=str)
    p.add_argument('--prompt_name', type=str, default='most_common_name')
    p.add_argument('--use_attribute', default=False, type=str2bool)
    p.add_argument('--dataset', type=str, default='semi-aves')
    p.add_argument('--test_imagenet_ood', default=False, action='store_true')
    p.add_argument('--database', type=str, default='LAION400M')
    p.add_argument('--data_source', type=str, default='fewshot',
                   choices=['fewshot', 'retrieved', 'fewshot+retrieved', 'dataset-cls', 'ltrain', 'ltrain+val',
                            'ltrain+val+unlabeled', 'ltrain+val+unlabeled+retrieved', 'fewshot+unlabeled',
                            'fewshot+retrieved+unlabeled'])
    p.add_argument('--shots', type=int, default=16)
    p.add_argument('--retrieval_split', type=str, default='T2T500+T2I0.25.txt')
    p.add_argument('--unlabeled_split', type=str, default='u_train_in_oracle.txt')
    p.add_argument('--val_split', type=str, default='fewshotX.txt')
    p.add_argument('--test_split', type=str, default='test.txt')
    p.add_argument('--seed', type=int, default=1)
    p.add_argument('--training_seed', type=int, default=1)
    p.add_argument('--method', type=str, default='finetune',
                   choices=['zeroshot', 'REAL-Linear', 'probing', 'finetune', 'finetune-mixed', 'finetune-multitask', 'CMLP',
                            'mixup', 'mixup-fs', 'cutmix', 'cutmix-fs', 'resizemix', 'dataset-cls', 'saliencymix',
                            'attentivemix', 'CMO', 'FLYP', 'fixmatch'])
    p.add_argument('--fewshot_ratio', type=float, default=0.5)
    p.add_argument('--cls_init', type=str, default='REAL-Prompt', choices=['random', 'text', 'REAL-Prompt', 'REAL-Linear'])
    p.add_argument('--no_wsft', default=False, action='store_true')
    p.add_argument('--no_tau', default=False, action='store_true')
    p.add_argument('--mix_prob', type=float, default=0.5)
    p.add_argument('--mixup_alpha', type=float, default=1.0)
    p.add_argument('--skip_stage1', default=False, action='store_true')
    p.add_argument('--skip_stage2', default=False, action='store_true')
    p.add_argument('--stage1_model_path', default=None, type=str)
    p.add_argument('--cutmix_beta', type=float, default=1.0)
    p.add_argument('--recal_prompt', action='store_true')
    p.add_argument('--recal_fea', action='store_true')
    p.add_argument('--pre_extracted', default=False, action='store_true')
    p.add_argument('--locked_text', action='store_true')
    p.add_argument('--freeze_visual', default=False, type=str2bool)
    p.add_argument('--tau_norm', default=True, type=str2bool)
    p.add_argument('--check_zeroshot', action='store_true')
    p.add_argument('--zeroshot_only', action='store_true')
    p.add_argument('--early_stop', default=False, type=str2bool)
    p.add_argument('--epochs', type=int, default=0)
    p.add_argument('--stop_epochs', type=int, default=200)
    p.add_argument('--num_workers', type=int, default=8)
    # input pipeline (ref utils/dataloader.py:107-135: 8 PIL workers decode and transform every image).  Default for
    # image-folder data: the workers only READ the files; baseline JPEGs are entropy-decoded on host threads and inverse-DCT-ed /
    # upsampled / colour-converted / cropped / resized / flipped / normalised on the GPU, bytes identical to Pillow's (other formats
    # are decoded by Pillow per file and counted in the epoch log).  --host_decode: workers decode with Pillow, the GPU transforms.
    # --host_transform: the reference's pipeline unchanged (Pillow decode + host transform in the workers).
    p.add_argument('--host_transform', action='store_true', help='the reference\'s input pipeline: loader workers decode and transform with Pillow')
    p.add_argument('--host_decode', action='store_true', help='loader workers decode with Pillow; crop / bicubic resize / flip / normalise on the GPU')
    p.add_argument('--gpu_transform', action='store_true', help='(accepted for older command lines) same as --host_decode')
    p.add_argument('--dp_full_load', action='store_true',
                   help='data-parallel global-batch mode: every rank loads the WHOLE --bsz batch and slices its rows (the round-3 behaviour); '
                        'default: a rank-aware sampler loads only the rank\'s rows and their CutMix partners')
    p.add_argument('--gpu_decode', action='store_true', help='(accepted for older command lines) the default pipeline')
    p.add_argument('--start_validation', type=int, default=0)
    p.add_argument('--lr_classifier', type=float, default=1e-4)
    p.add_argument('--lr_backbone', type=float, default=1e-6)
    p.add_argument('--lr_projector', type=float, default=None)
    p.add_argument('--wd', type=float, default=1e-2)
    p.add_argument('--bsz', type=int, default=32)
    p.add_argument('--optim', type=str, default='AdamW', choices=['AdamW', 'SGD'])
    p.add_argument('--temperature', type=float, default=0.07)
    p.add_argument('--alpha', type=float, default=0.5)
    p.add_argument('--loss_name', type=str, default='CE', choices=['CE', 'WeightedCE', 'Focal', 'BalancedSoftmax'])
    p.add_argument('--save_ckpt', default=False, type=str2bool)
    p.add_argument('--save_freq', type=int, default=10)
    p.add_argument('--utrain', type=str, default=None)
    # build-only additions
    p.add_argument('--dist_backend', type=str, default='nccl', choices=['nccl', 'gloo'],
                   help="torch.distributed backend when launched under torch.distributed.run (nccl = RCCL on ROCm)")
    p.add_argument('--dp_weak', action='store_true',
                   help='data-parallel runs: every rank loads its own --bsz rows (global batch = world x bsz) instead of '
                        'splitting the --bsz batch')
    p.add_argument('--dp_wire', type=str, default='auto', choices=['auto', 'fp32', 'bf16'],
                   help="data-parallel runs: what a gradient bucket crosses the xGMI links as (swat_amd.ddp.GradWire).  auto = fp32 when "
                        "the --bsz batch is split over the ranks (that mode reproduces the one-GPU trajectory), bf16 with --dp_weak "
                        "(throughput: 88 M fp32 gradients are 352 MB per step and GPU, more than the links move in a ViT-B/32 step)")
    p.add_argument('--synthetic', type=int, default=0, help='train on N seeded synthetic images (no dataset needed)')
    p.add_argument('--num_classes', type=int, default=200, help='classes of the synthetic dataset')
    p.add_argument('--config', type=str, default='config.yml')
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    if os.path.exists(args.config):
        with open(args.config) as f:
            config = yaml.load(f, Loader=yaml.FullLoader)
        args.dataset_path, args.retrieved_path = config['dataset_path'], config['retrieved_path']
    else:
        args.dataset_path, args.retrieved_path = 'data', 'data/retrieved'
    if args.method == 'zeroshot':
        args.check_zeroshot = args.zeroshot_only = args.skip_stage2 = True
    if args.cls_init in ('REAL-Prompt', 'REAL-Linear', 'random'):
        args.prompt_name = 'most_common_name'
    elif args.cls_init == 'text':
        args.prompt_name = 'name'
    if args.method in ("probing", "CMLP", "REAL-Linear"):
        args.freeze_visual = True
    else:
        args.freeze_visual = False
        args.pre_extracted = False
    if not args.freeze_visual:
        assert args.pre_extracted is False, 'visual encoder not frozen, pre-extracted features are not compatible.'
    if args.model_path is not None:
        assert args.pre_extracted is False, 'reloading a trained model, pre-extracted features are not compatible.'
    ds = os.path.join(args.dataset_path, args.dataset)
    rt = os.path.join(args.retrieved_path, args.dataset)
    few = f'fewshot{args.shots}_seed{args.seed}.txt'
    args.fewshot_data = [[few], [ds]]
    args.retrieval_data = [[args.retrieval_split], [rt]]
    args.val_split = [[few], [ds]]
    args.test_split = [['test.txt'], [ds]]
    if args.data_source == 'fewshot':
        args.train_split = [[few], [ds]]
    elif args.data_source == 'retrieved':
        args.train_split = [[args.retrieval_split], [rt]]
    elif args.data_source == 'fewshot+retrieved':
        args.train_split = [[few, args.retrieval_split], [ds, rt]]
    elif args.data_source == 'dataset-cls':
        args.train_split, args.val_split, args.test_split = [['dataset_train.txt'], ['']], [['dataset_val.txt'], ['']], [['dataset_test.txt'], ['']]
    else:
        raise NotImplementedError
    args.folder = f'{args.folder}/output_{args.dataset}'
    args.dataset_root = f'data/{args.dataset}'
    return args
