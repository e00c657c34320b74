"""B1 / B2: encode every retrieved image AND its caption with CLIP -> the `mined.pth` dict the samplers read
(mirror of the reference's retrieval/extract_mined_feature.py; same function names, flags and output file).

    python -m swat_amd.retrieval.extract_mined_feature --dataset semi-aves --model_cfg vitb32_openclip_laion400m
    python -m torch.distributed.run --nproc-per-node 8 -m swat_amd.retrieval.extract_mined_feature --dataset semi-aves   # 8 GPUs

ref: MINED_DATASET_ROOT_DICT / CAPTION_MAP_DICT :19-46, extract_mined_feats :49-100, extract_mined_feats_batch :103-213
(pass 1: captions -> tokenizer -> encode_text -> L2-normalise; pass 2: images -> encode_image -> L2-normalise; both in
loader order, shuffle=False), pre_extract_mined_fea :252-280 (MinedDataset + caption map, bsz 1024, saves
{caption_features, labels, filepath, image_features} to <root>/<dataset>/<dataset>_<model_cfg>_mined.pth), __main__ :284-315.

What differs on the MI355X: both towers run on the HIP kernels; the features stay in HBM until the end (10 M x 512 fp32 x 2 =
41 GB of 288 GB), so the reference's spill-to-disk-every-100-batches-and-reload (:128-140, :181-200 — whose reload walks
os.listdir() order, i.e. can permute the batches) is not needed: `save_dir` / `batch_size` are accepted and ignored.  Under
torch.distributed.run every rank encodes files rank, rank + world, ... of the list and rank 0 writes the merged dict in the
single-process row order.
"""
import argparse
import os
import pickle

import torch
from torch.utils.data import DataLoader

from .. import _lib
from ..features import l2_normalize


def _retrieved_root(config_path=None):
    """`retrieved_path` of config.yml (the reference reads ../config.yml at import time, :16-18)."""
    import yaml
    for p in ([config_path] if config_path else []) + ['config.yml', '../config.yml']:
        if p and os.path.exists(p):
            with open(p) as f:
                return yaml.load(f, Loader=yaml.FullLoader)['retrieved_path'].rstrip('/')
    return os.environ.get('SWAT_RETRIEVED_PATH', 'data/retrieved')


_ALL = ('semi-aves', 'fgvc-aircraft', 'eurosat')  # mined with every synonym; the others with a random one (ref :19-46)
_DATASETS = ('semi-aves', 'fgvc-aircraft', 'eurosat', 'dtd', 'flowers102', 'oxford_pets', 'food101', 'stanford_cars', 'imagenet')


def mined_dataset_root(dataset, root):
    return f'{root}/{dataset}/{dataset}_retrieved_LAION400M-all_synonyms-{"all" if dataset in _ALL else "random"}'


def caption_map_path(dataset, root):
    return f'{root}/{dataset}/{dataset}_metadata-{"all" if dataset in _ALL else "random"}-0.0-LAION400M.map'


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


@torch.no_grad()
@_lib.retry_op_by_op
def extract_mined_feats_batch(model, dataloader, tokenizer, save_dir=None, batch_size=100, device="cuda"):
    """dataloader yields (imgs [B,3,H,W], labels [B], file_path list[str], captions list[str]) — MinedDataset's order.
    -> {'caption_features' [N,E], 'labels' [N], 'filepath' list[N], 'image_features' [N,E]} (CPU tensors), loader order."""
    model.eval()
    cap_feats, labels_lst, paths = [], [], []
    # pass 1: captions.  `caption_loader` (pre_extract_mined_fea): the same files in the same order without reading a pixel
    for imgs, labels, file_path, captions in getattr(dataloader, 'caption_loader', dataloader):
        tok = tokenizer(list(captions)).to(device)
        cap_feats.append(l2_normalize(model.encode_text(tok)))
        labels_lst.append(labels.long())
        paths.extend(list(file_path))
    img_feats = []
    for imgs, labels, file_path, captions in dataloader:            # pass 2: images
        img_feats.append(l2_normalize(model.visual.forward_raw(imgs.to(device, non_blocking=True))))
    return {'caption_features': torch.cat(cap_feats).cpu(), 'labels': torch.cat(labels_lst).cpu(), 'filepath': paths,
            'image_features': torch.cat(img_feats).cpu()}


extract_mined_feats = extract_mined_feats_batch  # the reference's un-spilled form (:49-100) computes the same dict


def merge_rank_dicts(parts):
    """Per-rank dicts of a rank-strided extraction (rank r holds files r, r + W, ...) -> the single-process dict."""
    world = len(parts)
    n = sum(len(p['filepath']) for p in parts)
    out = {'filepath': [None] * n}
    for key in ('caption_features', 'image_features', 'labels'):
        first = parts[0][key]
        out[key] = torch.empty((n,) + tuple(first.shape[1:]), dtype=first.dtype)
    for r, p in enumerate(parts):
        out['filepath'][r::world] = p['filepath']
        for key in ('caption_features', 'image_features', 'labels'):
            out[key][r::world] = p[key]
    return out


def gather_rank_parts(part, n_total, rank, world):
    """Rank 0 <- every rank's strided share, merged into single-process row order.  The feature matrices travel as
    tensors (dist.gather of equal-sized, zero-padded blocks: 2 x 2 KB per row), only the path lists as pickled objects."""
    import torch.distributed as dist
    n_max = (n_total + world - 1) // world
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    parts = [{} for _ in range(world)] if rank == 0 else None
    for key in ('caption_features', 'image_features', 'labels'):
        t = part[key]
        blk = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        blk[:t.shape[0]] = t.to(dev)
        out = [torch.empty_like(blk) for _ in range(world)] if rank == 0 else None
        dist.gather(blk, out, dst=0)
        if rank == 0:
            for r in range(world):
                parts[r][key] = out[r][:len(range(r, n_total, world))].cpu()
    paths = [None] * world if rank == 0 else None
    dist.gather_object(part['filepath'], paths, dst=0)
    if rank != 0:
        return None
    for r in range(world):
        parts[r]['filepath'] = paths[r]
        assert len(paths[r]) == parts[r]['labels'].shape[0], "a rank encoded a different number of files than its share"
    return merge_rank_dicts(parts)


def pre_extract_mined_fea(args, dataset_name, dataset_root, caption_path, model_cfg, model, preprocess, tokenizer):
    """ref :252-280 — same arguments, same output file; returns the dict (rank 0) for callers that go on in-process."""
    from ..dataloader import MinedDataset
    rank, world = _dist()
    if rank == 0:
        print('\nextracting features for', dataset_name, model_cfg)
        print('dataset_root:', dataset_root)
    with open(caption_path, 'rb') as f:
        caption_map = pickle.load(f)
    fnames = None
    if world > 1:
        # every rank strides over RANK 0's file list: two processes (or two mounts) need not list a directory in the same order
        import pathlib
        import torch.distributed as dist
        box = [[str(f) for f in pathlib.Path(dataset_root).glob("**/*.jpg")] if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        fnames = box[0]
    dataset = MinedDataset(dataset_root=dataset_root, transform=preprocess, caption_map=caption_map, rank=rank, world=world,
                           fnames=fnames)
    dev_tf = getattr(preprocess, 'swat_transform', None)
    if dev_tf is not None and not getattr(args, 'host_transform', False) and torch.cuda.is_available():
        # the default input pipeline (swat_amd/dataloader.py): workers only READ the files; baseline JPEGs are entropy-decoded on host
        # threads and decoded / resized / cropped / normalised on the GPU, bytes identical to what `preprocess` computes with Pillow —
        # the decode in 16 PIL workers is where the reference's extraction spends its time; the caption pass
        # reads no image at all
        import copy
        from ..dataloader import DeviceTransformLoader
        from ..transforms import ReadOnly
        dataset.loader, dataset.transform = ReadOnly.load, ReadOnly(dev_tf[0], dev_tf[1])
        loader = DeviceTransformLoader(dataset, dev_tf[0], dev_tf[1], torch.device('cuda', torch.cuda.current_device()), decode=True,
                                       batch_size=args.bsz, shuffle=False, num_workers=args.num_workers, drop_last=False)
        cap_ds = copy.copy(dataset)
        cap_ds.captions_only = True
        loader.caption_loader = DataLoader(cap_ds, batch_size=args.bsz, shuffle=False, num_workers=0, drop_last=False)
    else:
        loader = DataLoader(dataset, batch_size=args.bsz, shuffle=False, num_workers=args.num_workers, drop_last=False, pin_memory=True)
    part = extract_mined_feats_batch(model, loader, tokenizer, save_dir=f'{args.root}/{dataset_name}/temp')
    if world > 1:
        part = gather_rank_parts(part, dataset.n_total, rank, world)
        if rank != 0:
            return None
    save_root = f'{args.root}/{dataset_name}'
    os.makedirs(save_root, exist_ok=True)
    output_fn = f'{save_root}/{dataset_name}_{model_cfg}_mined.pth'
    torch.save(part, output_fn)
    print(f'\nSaved to extracted feature to: {output_fn}')
    return part


def main(argv=None):
    root = _retrieved_root()
    parser = argparse.ArgumentParser(description='Arguments for script.')
    parser.add_argument('--dataset', type=str, default='semi-aves', help='Dataset name.')
    parser.add_argument('--root', type=str, default=root, help='Root directory for storing mined data.')
    parser.add_argument('--model_cfg', type=str, default='vitb32_openclip_laion400m',
                        choices=['vitb32_openclip_laion400m', 'vitb32_openclip_laion2b', 'vitb16_openclip_laion400m',
                                 'vitl14_openclip_laion400m'], help='ViT Transformer arch.')
    parser.add_argument('--bsz', type=int, default=1024, help='Batch size.')
    parser.add_argument('--num_workers', type=int, default=16, help='Number of workers for dataloader.')
    parser.add_argument('--dist_backend', type=str, default='nccl', choices=['nccl', 'gloo'])
    parser.add_argument('--host_transform', action='store_true', help="the reference's input pipeline: loader workers decode and transform with Pillow")
    args = parser.parse_args(argv)
    from .. import dp
    dp.init_from_env(args, split_batch=False)  # before anything touches the GPU; --bsz is per rank here (nothing is split)
    if dp.is_main(args):
        for arg in vars(args):
            print(f'{arg}: {getattr(args, arg)}')
    from ..engine import get_engine
    model, preprocess, tokenizer = get_engine(model_cfg=args.model_cfg, device='cuda')
    model.eval()
    out = pre_extract_mined_fea(args=args, dataset_name=args.dataset, dataset_root=mined_dataset_root(args.dataset, args.root),
                                caption_path=caption_map_path(args.dataset, args.root), model_cfg=args.model_cfg, model=model,
                                preprocess=preprocess, tokenizer=tokenizer)
    dp.barrier(args)
    return out


if __name__ == '__main__':
    main()
