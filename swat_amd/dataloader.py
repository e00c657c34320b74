"""Datasets / loaders without torchvision (mirror of the reference's utils/dataloader.py + utils/datasets/dataset_utils.py).

ref: MyDataset dataset_utils.py:130-180 (list files `relpath label source`), TensorDataset :237-255,
     get_dataloader dataloader.py:107-135 (train bsz=args.bsz shuffle drop_last; val 128; test 256),
     extract_dataloader :11-31, extract_train_dataloader :278-296, set_dataloaders :182-254.
"""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .features import extract_test_feats
from .transforms import DecodeOnly, DeviceBatchTransform, ReadOnly, build_transform


def _pil_loader(path):
    from PIL import Image
    with open(path, 'rb') as f:
        return Image.open(f).convert('RGB')


class MyDataset(Dataset):
    def __init__(self, dataset_root, split, transform, tokenized_text_prompts=None, loader=_pil_loader):
        self.loader, self.transform, self.tokenized_text_prompts = loader, transform, tokenized_text_prompts
        lines = []
        for file, path in zip(split[0], split[1]):
            with open(os.path.join(dataset_root, file), 'r') as f:
                lines.extend(os.path.join(path, l) for l in f.readlines())
        self.data, self.labels = [], []
        for line in lines:
            path, id, is_fewshot = line.strip('\n').split(' ')
            self.data.append((path, int(id), int(is_fewshot)))
            self.labels.append(int(id))
        self.targets = self.labels

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, i):
        path, label, source = self.data[i]
        img = self.transform(self.loader(path))
        if self.tokenized_text_prompts is not None:
            tok = self.tokenized_text_prompts[str(label)]['all']
            tok = tok[torch.randint(0, tok.size(0), (1,))]  # one random prompt per sample (ref :174)
        else:
            tok = torch.zeros(1, 1, dtype=torch.long)
        return img, label, tok, source


class MinedDataset(Dataset):
    """Retrieved LAION images of one dataset: <dataset_root>/<class id>/<image id>.jpg + their captions
    (ref utils/datasets/dataset_utils.py:282-320).  __getitem__ -> (img, label, file_path, caption) — the reference's order.
    caption_map: {str(label): {image id: caption}} (the pickled `*_metadata-*.map`, retrieval/extract_mined_feature.py:252-262).
    rank / world: this process encodes files rank, rank + world, ... of the list (world 1 = all)."""

    def __init__(self, transform=None, dataset_root=".", caption_map=None, rank=0, world=1, loader=_pil_loader, fnames=None):
        import pathlib
        self.dataset_root = pathlib.Path(dataset_root)
        # directory order, like the reference (not sorted).  `fnames`: the list every rank must stride over — rank 0's,
        # broadcast by the caller, because directory order is not guaranteed to agree between processes / mounts
        self.fnames = [pathlib.Path(f) for f in fnames] if fnames is not None else list(self.dataset_root.glob("**/*.jpg"))
        if len(self.fnames) == 0:
            raise ValueError('Check image suffix. No images found in the specified dataset_root: {}'.format(dataset_root))
        self.num_classes = len(set(f.parent.name for f in self.fnames))
        self.n_total = len(self.fnames)
        self.fnames = self.fnames[rank::world]
        self.caption_map, self.transform, self.loader = caption_map, transform, loader
        self.captions_only = False

    def __len__(self):
        return len(self.fnames)

    def __getitem__(self, i):
        f = self.fnames[i]
        label = int(f.parent.name)
        img_id = f.name.split('.')[0]
        if self.captions_only:   # the caption pass of the extraction needs no pixels (the reference decodes them anyway, :113-125)
            return 0, label, str(f), self.caption_map[str(label)][img_id] if self.caption_map is not None else ''
        img = self.loader(str(f))
        if self.transform is not None:
            img = self.transform(img)
        if self.caption_map is not None:
            return img, label, str(f), self.caption_map[str(label)][img_id]
        return img, label, str(f), ''


class TensorDataset(Dataset):
    """Pre-extracted features saved by extract_test_feats: {'image_features' [N,E], 'labels' [N]}."""

    def __init__(self, pre_extracted_path=None, device='cpu', data=None):
        d = data if data is not None else torch.load(pre_extracted_path, map_location='cpu', weights_only=False)
        self.input_tensor, self.label_tensor = d['image_features'], d['labels']

    def __len__(self):
        return self.label_tensor.size(0)

    def __getitem__(self, i):
        return self.input_tensor[i], self.label_tensor[i], self.label_tensor[i], self.label_tensor[i]


class SyntheticImages(Dataset):
    """Seeded N(0,1) images with class-dependent mean shift: a dataset-free smoke path for the CLI (--synthetic)."""

    def __init__(self, n, num_classes, size=224, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.labels = torch.randint(0, num_classes, (n,), generator=g)
        self.seed, self.size, self.n = seed, size, n
        self.targets = self.labels.tolist()

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        y = int(self.labels[i])
        img = torch.randn(3, self.size, self.size, generator=g) + 0.15 * ((y % 7) - 3)
        return img, y, torch.zeros(1, 1, dtype=torch.long), 1


def _raw_collate(samples):
    """--host_decode: the decoded uint8 [H, W, 3] images of different sizes cross as ONE shared-memory tensor + their shapes (a list
    of numpy arrays is pickled through the worker's pipe: 16 MB per 32-image batch); everything else as in _file_collate"""
    from torch.utils.data import default_collate
    shapes = [tuple(s[0][0].shape) for s in samples]
    blob = torch.cat([s[0][0].reshape(-1) for s in samples])
    geo = [tuple(int(v) for v in s[0][1]) for s in samples]
    rest = tuple(t.numpy() if isinstance(t, torch.Tensor) else t for t in default_collate([tuple(s[1:]) for s in samples]))
    return (blob, shapes, geo), rest


def _file_collate(samples):
    """--gpu_decode: images stay a list of FILE byte arrays (+ geometry, + the (w, h, baseline) the worker read from the headers)"""
    from torch.utils.data import default_collate
    # ONE uint8 tensor for the batch's file bytes (+ their lengths): a worker process hands tensors to the main process through
    # shared memory, a list of numpy arrays would be pickled through its pipe — 3 MB per 32-image batch, which bounded the loader
    # at ~6 k img/s against the 22 k img/s the decode behind it sustains (bench.py: loader leg)
    blob = torch.cat([s[0][0] for s in samples])
    lens = [int(s[0][0].numel()) for s in samples]
    geo = [tuple(int(v) for v in s[0][1]) for s in samples]
    sizes = [(int(s[0][2][0]), int(s[0][2][1])) if int(s[0][2][2]) else None for s in samples]
    # ... and the blob is the ONLY tensor that crosses: every tensor of a batch costs the main process a file-descriptor hand-over
    # from the worker (~0.5 ms each, serial in the main process, whatever the number of workers); labels / prompt tokens / sources go
    # as numpy arrays inside the pickle and become the tensors default_collate would have made on the other side (_file_uncollate)
    rest = tuple(t.numpy() if isinstance(t, torch.Tensor) else t for t in default_collate([tuple(s[1:]) for s in samples]))
    return (blob, lens, geo, sizes), rest


class DeviceTransformLoader:
    """A DataLoader whose workers only decode (transform = DecodeOnly) and whose batches get crop / resize / flip /
    normalise on the GPU (`swat_preprocess_batch`) as they leave the iterator: yields the tuples the reference's loaders
    yield (utils/datasets/dataset_utils.py:165-180), with the image batch already on the device."""

    def __init__(self, dataset, n_px, mode, device, decode=False, **loader_kw):
        # decode: the workers hand over the image FILES (transform = ReadOnly) and the JPEG decode runs on host threads + the GPU too
        self.decode = decode
        self.loader = DataLoader(dataset, collate_fn=_file_collate if decode else _raw_collate, **loader_kw)
        self.dataset, self.batch_size, self.batch_sampler = dataset, self.loader.batch_size, self.loader.batch_sampler
        self.tf = DeviceBatchTransform(n_px, mode, device)
        # host threads of the entropy decode (one file per thread at a time): a thread per image of a 32-image batch where the host has
        # the cores beside the loader's worker processes (the library's own default is one per two files)
        self.threads = max(1, min(32, (os.cpu_count() or 8) - int(loader_kw.get('num_workers', 0) or 0)))

    def __len__(self):
        return len(self.loader)

    def _decoded_batches(self):
        """Decode path, one batch AHEAD of the consumer: a producer thread turns the workers' file batches into normalised device
        batches on a stream of its own (ctypes releases the GIL: the host-thread entropy decode of batch i + 1 runs while the main
        thread enqueues step i, and its upload is not queued behind that step's kernels); the consumer's stream waits for the batch's
        event.  Same batches in the same order as the inline form (geometry is drawn in the workers)."""
        import queue
        import threading
        dev = self.tf.device
        q = queue.Queue(maxsize=2)
        side = self._side = getattr(self, '_side', None) or torch.cuda.Stream(device=dev)
        stop = threading.Event()
        # the DataLoader's iterator is created HERE, in the consumer's thread, at the moment the reference's loop creates its own: it
        # draws the worker base seed and the shuffle seed from torch's GLOBAL generator — drawn from the producer
        # thread they would race with the loop's own draws (the CutMix permutation)
        it = iter(self.loader)

        def produce():
            try:
                with torch.cuda.device(dev):
                    for (blob, lens, geo, sizes), rest in it:
                        if stop.is_set():
                            return
                        rest = [torch.from_numpy(t) if isinstance(t, np.ndarray) else t for t in rest]
                        a, e = blob.numpy(), 0
                        files = []
                        for n in lens:
                            files.append(a[e:e + n])
                            e += n
                        out = self.tf.from_jpeg(files, geometry=geo, sizes=sizes, stream=side, threads=self.threads)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        q.put((out, ev, rest))
                q.put(None)
            except BaseException as e:   # noqa: re-raised in the consumer
                q.put(e)

        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                out, ev, rest = item
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                out.record_stream(cur)
                yield (out, *rest)
        finally:
            stop.set()
            while t.is_alive():   # (a producer blocked on a full queue: drain it)
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            t.join()

    def __iter__(self):
        if self.decode and self.loader.num_workers > 0:
            yield from self._decoded_batches()
            return
        if self.decode:
            # no worker processes: the dataset code (file read, geometry draws from torch's global generator) runs in whichever thread
            # pulls the batches — it must be the caller's, in the caller's order
            for (blob, lens, geo, sizes), rest in self.loader:
                a, e, files = blob.numpy(), 0, []
                for n in lens:
                    files.append(a[e:e + n])
                    e += n
                yield (self.tf.from_jpeg(files, geometry=geo, sizes=sizes, threads=self.threads),
                       *(torch.from_numpy(t) if isinstance(t, np.ndarray) else t for t in rest))
            return
        for (blob, shapes, geo), rest in self.loader:
            a, e, imgs = blob.numpy(), 0, []
            for shp in shapes:
                n = int(np.prod(shp))
                imgs.append(a[e:e + n].reshape(shp))
                e += n
            yield (self.tf(imgs, geometry=geo), *(torch.from_numpy(t) if isinstance(t, np.ndarray) else t for t in rest))


def input_pipeline(args):
    """'gpu_decode' (default: workers read files, decode + transform on host threads / the GPU), 'gpu_transform' (--host_decode:
    workers decode with Pillow, the GPU transforms) or 'host' (--host_transform: the reference's pipeline, utils/dataloader.py:107-135)."""
    if getattr(args, 'host_transform', False):
        return 'host'
    if getattr(args, 'host_decode', False) or (getattr(args, 'gpu_transform', False) and not getattr(args, 'gpu_decode', False)):
        return 'gpu_transform'
    return 'gpu_decode'


def load_dataset(dataset_root, split, preprocess, tokenized_text_prompts):
    return MyDataset(dataset_root, split, preprocess, tokenized_text_prompts)


def get_dataloader(args, train_split, val_split, test_split, tokenized_text_prompts, preprocess):
    if getattr(args, 'synthetic', 0):
        C = args.num_classes
        tr, va, te = SyntheticImages(args.synthetic, C, seed=1), SyntheticImages(max(64, args.synthetic // 4), C, seed=2), \
            SyntheticImages(max(64, args.synthetic // 4), C, seed=3)
    else:
        tr = load_dataset(args.dataset_root, train_split, build_transform(224, 'train'), tokenized_text_prompts)
        va = load_dataset(args.dataset_root, val_split, preprocess, tokenized_text_prompts)
        te = load_dataset(args.dataset_root, test_split, preprocess, tokenized_text_prompts)
    # data-parallel, --dp_weak: the shuffled epoch is dealt out by rank (swat_amd/dp.py); otherwise the reference's loader
    train_kw = dict(shuffle=True, batch_size=args.bsz, drop_last=True)
    if getattr(args, 'world', 1) > 1 and getattr(args, 'dp_weak', False):
        from .dp import RankStridedSampler
        train_kw = dict(sampler=RankStridedSampler(len(tr), args.rank, args.world), batch_size=args.bsz, drop_last=True)
    elif getattr(args, 'world', 1) > 1 and not getattr(args, 'dp_full_load', False):
        # global batch split over the ranks: each rank's loader yields its own rows (+ CutMix partners) only
        from .dp import RankBatchSampler
        train_kw = dict(batch_sampler=RankBatchSampler(len(tr), args.bsz, args.rank, args.world))
    pipe = input_pipeline(args)
    gpu_decode = pipe == 'gpu_decode'
    if pipe != 'host' and not getattr(args, 'synthetic', 0):
        dev = getattr(args, 'device', 'cuda')
        for ds, mode in ((tr, 'train'), (va, 'val'), (te, 'val')):
            if gpu_decode:   # workers read files; decode + transform on the GPU (same draws, same order: the geometry needs only (w, h))
                ds.loader, ds.transform = ReadOnly.load, ReadOnly(224, mode)
            else:
                ds.transform = DecodeOnly(224, mode)
        train_loader = DeviceTransformLoader(tr, 224, 'train', dev, decode=gpu_decode, num_workers=args.num_workers, **train_kw)
        val_loader = DeviceTransformLoader(va, 224, 'val', dev, decode=gpu_decode, batch_size=128, drop_last=False, shuffle=False, num_workers=args.num_workers)
        test_loader = DeviceTransformLoader(te, 224, 'val', dev, decode=gpu_decode, batch_size=256, drop_last=False, shuffle=False, num_workers=args.num_workers)
        return train_loader, val_loader, test_loader
    train_loader = DataLoader(tr, pin_memory=True, num_workers=args.num_workers, **train_kw)
    val_loader = DataLoader(va, batch_size=128, drop_last=False, pin_memory=True, shuffle=False, num_workers=args.num_workers)
    test_loader = DataLoader(te, batch_size=256, drop_last=False, pin_memory=True, shuffle=False, num_workers=args.num_workers)
    return train_loader, val_loader, test_loader


def image_loader(args, ds, mode, n_px=224, **loader_kw):
    """A loader over an image-folder dataset `ds` (MyDataset / MinedDataset built with the host transform of `mode`) in the input
    pipeline the arguments select (input_pipeline): the default hands the files to DeviceTransformLoader (workers read, host-thread
    entropy decode, GPU decode + transform — bytes identical to the host transform, same random draws in the same order),
    --host_decode keeps the Pillow decode in the workers, --host_transform is the reference's DataLoader unchanged."""
    pipe = input_pipeline(args)
    if pipe == 'host' or isinstance(ds, SyntheticImages) or not torch.cuda.is_available():
        return DataLoader(ds, **loader_kw)
    n_px, mode = getattr(getattr(ds, 'transform', None), 'swat_transform', (n_px, mode))   # (the host transform says what it computes)
    if pipe == 'gpu_decode':
        ds.loader, ds.transform = ReadOnly.load, ReadOnly(n_px, mode)
    else:
        ds.transform = DecodeOnly(n_px, mode)
    loader_kw.pop('pin_memory', None)
    return DeviceTransformLoader(ds, n_px, mode, getattr(args, 'device', 'cuda'), decode=pipe == 'gpu_decode', **loader_kw)


def _dataset_for(args, split, preprocess, tokenized_text_prompts, seed):
    if getattr(args, 'synthetic', 0):
        return SyntheticImages(max(64, args.synthetic // 4), args.num_classes, seed=seed)
    return load_dataset(args.dataset_root, split, preprocess, tokenized_text_prompts)


def extract_dataloader(args, model, split, fea_path, preprocess, tokenized_text_prompts, bsz=128, seed=2):
    """ref :11-31 — encode the split once with `model`, cache to fea_path, serve the features (bsz 128, no shuffle)."""
    ds = _dataset_for(args, split, preprocess, tokenized_text_prompts, seed)
    feats = extract_test_feats(model, image_loader(args, ds, 'val', batch_size=128, shuffle=False, num_workers=args.num_workers))
    os.makedirs(os.path.dirname(fea_path) or '.', exist_ok=True)
    torch.save(feats, fea_path)
    return DataLoader(TensorDataset(data=feats), batch_size=bsz, shuffle=False, drop_last=False, num_workers=0)


def extract_train_dataloader(args, model, split, fea_path, preprocess, tokenized_text_prompts, bsz=32):
    """ref :278-296 — same with the TRAIN transform, shuffled batches of args.bsz."""
    if getattr(args, 'synthetic', 0):
        ds = SyntheticImages(args.synthetic, args.num_classes, seed=1)
    else:
        ds = load_dataset(args.dataset_root, split, build_transform(224, 'train'), tokenized_text_prompts)
    # like the reference: shuffled batches of bsz with the tail batch dropped BEFORE the features are cached (the stage-2
    # probing set is len // bsz * bsz samples, and the shuffle draws from the global torch RNG at this point)
    feats = extract_test_feats(model, image_loader(args, ds, 'train', batch_size=bsz, pin_memory=True, shuffle=True,
                                                   num_workers=args.num_workers, drop_last=True))
    os.makedirs(os.path.dirname(fea_path) or '.', exist_ok=True)
    torch.save(feats, fea_path)
    return DataLoader(TensorDataset(data=feats), batch_size=bsz, shuffle=True, drop_last=True, num_workers=0)


def set_dataloaders(args, model, tokenized_text_prompts, preprocess, logger):
    """ref :182-254 (image loaders; the pre-extracted branch serves cached features)."""
    train_loader, val_loader, test_loader = get_dataloader(args, args.train_split, args.val_split, args.test_split,
                                                           tokenized_text_prompts, preprocess)
    logger.info(f'len(train_loader): {len(train_loader)}, len(val_loader): {len(val_loader)}, len(test_loader): {len(test_loader)}')
    return train_loader, val_loader, test_loader
