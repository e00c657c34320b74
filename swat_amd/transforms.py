"""Image transforms without torchvision (ref: utils/extras.py:66-85 `transform`, open_clip's val preprocess).

train: RandomResizedCrop(n, scale (0.9,1), ratio (3/4,4/3), bicubic) + RandomHorizontalFlip + ToTensor + Normalize
val  : Resize(n, bicubic) + CenterCrop(n) + ToTensor + Normalize
Randomness comes from torch's global CPU generator, as torchvision's does.
"""
import math

import numpy as np
import torch

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def _to_tensor_normalize(img):
    a = np.asarray(img.convert('RGB'), dtype=np.float32) / 255.0
    t = torch.from_numpy(a).permute(2, 0, 1).contiguous()
    return (t - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)


def _rrc_params(w, h, scale, ratio):
    area = h * w
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect = math.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1]).item())
        cw, ch = int(round(math.sqrt(target_area * aspect))), int(round(math.sqrt(target_area / aspect)))
        if 0 < cw <= w and 0 < ch <= h:
            i = torch.randint(0, h - ch + 1, size=(1,)).item()
            j = torch.randint(0, w - cw + 1, size=(1,)).item()
            return i, j, ch, cw
    in_ratio = w / h
    if in_ratio < ratio[0]:
        cw, ch = w, int(round(w / ratio[0]))
    elif in_ratio > ratio[1]:
        ch, cw = h, int(round(h * ratio[1]))
    else:
        cw, ch = w, h
    return (h - ch) // 2, (w - cw) // 2, ch, cw


def build_transform(n_px, mode='train'):
    from PIL import Image

    def train(img):
        w, h = img.size
        i, j, ch, cw = _rrc_params(w, h, (0.9, 1.0), (0.75, 1.3333))
        img = img.crop((j, i, j + cw, i + ch)).resize((n_px, n_px), Image.BICUBIC)
        if torch.rand(1).item() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        return _to_tensor_normalize(img)

    def val(img):
        w, h = img.size
        s = n_px / min(w, h)
        nw, nh = (n_px, max(n_px, int(round(h * s)))) if w <= h else (max(n_px, int(round(w * s))), n_px)
        img = img.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))
        return _to_tensor_normalize(img.crop((left, top, left + n_px, top + n_px)))

    fn = train if mode == 'train' else val
    fn.swat_transform = (n_px, mode)   # (lets a loader swap this host transform for its device form: DeviceTransformLoader)
    return fn


def _axis_ksize(in_size, out_size):
    return int(math.ceil(2.0 * max(float(in_size) / out_size, 1.0))) * 2 + 1


def sample_geometry(n, mode, w, h):
    """What the host transform would do to a w x h image, as the nine integers `swat_preprocess_batch` takes; the
    training form draws the crop box and then the flip coin from torch's CPU generator, exactly as `build_transform`."""
    if mode == 'train':
        i, j, ch, cw = _rrc_params(w, h, (0.9, 1.0), (0.75, 1.3333))
        flip = int(torch.rand(1).item() < 0.5)
        return j, i, cw, ch, n, n, 0, 0, flip
    s = n / min(w, h)
    nw, nh = (n, max(n, int(round(h * s)))) if w <= h else (max(n, int(round(w * s))), n)
    return 0, 0, w, h, nw, nh, int(round((nw - n) / 2.0)), int(round((nh - n) / 2.0)), 0


class DecodeOnly:
    """Worker-side half of the GPU transform: hands back the decoded bytes and this sample's geometry (the random draws
    happen here, in the worker, in the order the host transform makes them); `DeviceBatchTransform` does the rest for
    the collated batch in the main process."""

    def __init__(self, n_px, mode='train'):
        self.n_px, self.mode = n_px, mode

    def __call__(self, img):
        a = np.array(img.convert('RGB'), dtype=np.uint8)
        return torch.from_numpy(a), torch.tensor(sample_geometry(self.n_px, self.mode, a.shape[1], a.shape[0]), dtype=torch.int32)


class ReadOnly:
    """Worker-side half of GPU decode + transform (--gpu_decode): the worker only READS the file and draws this sample's geometry from
    the image size in its headers; `DeviceBatchTransform.from_jpeg` decodes (host threads + GPU) and transforms the collated batch.
    Used as the dataset's `loader` AND `transform`: loader(path) -> file bytes, transform(bytes) -> (uint8 [n] tensor, geometry, (w, h)).
    A file the GPU path does not decode (progressive, CMYK, PNG ...) is sized through Pillow here and decoded by Pillow there."""

    def __init__(self, n_px, mode='train'):
        self.n_px, self.mode = n_px, mode

    @staticmethod
    def load(path):
        with open(path, 'rb') as f:
            return f.read()

    def __call__(self, data):
        from . import jpeg
        wh = jpeg.info(data)
        known = wh is not None
        if not known:
            import io
            from PIL import Image
            wh = Image.open(io.BytesIO(data)).size
        geo = sample_geometry(self.n_px, self.mode, wh[0], wh[1])
        return (torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()), torch.tensor(geo, dtype=torch.int32),
                torch.tensor([wh[0], wh[1], int(known)], dtype=torch.int32))


class DeviceBatchTransform:
    """The same two transforms for a whole batch of decoded images on the GPU (include/swat_hip.h
    `swat_preprocess_batch`): crop + antialiased bicubic resize + flip + ToTensor + Normalize in three launches, bit for
    bit what `build_transform` computes per image with Pillow.  The host keeps what is inherently serial: JPEG decoding
    and the random draws (same generator, same order per image: crop box, then the flip coin).

        tf = DeviceBatchTransform(224, 'train'); batch = tf(list_of_PIL_images)   # float32 [B, 3, 224, 224] on the GPU
    ref: utils/extras.py:66-85, utils/datasets/dataset_utils.py:165-180
    """

    def __init__(self, n_px, mode='train', device='cuda'):
        from . import _lib
        self.n_px, self.mode, self.device = n_px, mode, torch.device(device)
        self._lib = _lib
        v = torch.arange(256, dtype=torch.float32) / 255.0
        lut = (v.view(1, 256) - torch.tensor(MEAN).view(3, 1)) / torch.tensor(STD).view(3, 1)
        self._lut = lut.contiguous().to(self.device)
        self._ws = None
        self._pin, self._pin_free, self._turn = [None, None], [None, None], 0  # two pinned staging buffers, reused

    def geometry(self, w, h):
        """-> (x0, y0, in_w, in_h, out_w, out_h, off_x, off_y, flip) for one image; draws from torch's CPU generator"""
        return sample_geometry(self.n_px, self.mode, w, h)

    def from_jpeg(self, files, geometry=None, sizes=None, stream=None, threads=0):
        """files: the image FILES (bytes / uint8 arrays).  Decode on host threads + GPU (swat_amd/jpeg.py: bytes identical to
        Pillow's `convert('RGB')`), then the same three transform launches on the decoded bytes where they already are — the pixels
        never visit the host.  geometry as for __call__ (drawn here, from (w, h), when not given)."""
        from . import jpeg
        dev = self.device
        with torch.cuda.device(dev):
            s = stream if stream is not None else torch.cuda.current_stream(dev)
            pixels, offsets, shapes = jpeg.decode_batch(files, dev, sizes=sizes, threads=threads, stream=s, tail_bytes=8)
            B, n = len(shapes), self.n_px
            if B == 0:
                return torch.empty(0, 3, n, n, device=dev)
            desc = np.zeros((B, 12), np.int32)
            row_start = np.zeros(B + 1, np.int64)
            ks = 5
            for b, (h, w) in enumerate(shapes):
                g = geometry[b] if geometry is not None else self.geometry(w, h)
                x0, y0, in_w, in_h, out_w, out_h, off_x, off_y, flip = g
                if not (0 <= x0 and 0 <= y0 and 0 < in_w and 0 < in_h and x0 + in_w <= w and y0 + in_h <= h
                        and 0 <= off_x and 0 <= off_y and off_x + n <= out_w and off_y + n <= out_h):
                    raise ValueError(f"image {b}: geometry {g} does not fit a {w} x {h} image / a {n}-pixel window")
                desc[b] = (w, h, x0, y0, in_w, in_h, out_w, out_h, off_x, off_y, flip, 0)
                row_start[b + 1] = row_start[b] + in_h
                ks = max(ks, _axis_ksize(in_w, out_w), _axis_ksize(in_h, out_h))
            meta = np.concatenate([offsets, row_start, desc.reshape(-1).view(np.int64)])
            lib, ptr = self._lib.lib(), self._lib.ptr
            # (the descriptor upload comes from a pinned ring: a pageable source makes the copy synchronous with the stream — the host
            # would stand still once per batch until the GPU has reached it)
            if not hasattr(self, '_meta_pin'):
                self._meta_pin, self._meta_ev, self._meta_i = [None] * 4, [None] * 4, 0
            k = self._meta_i = (self._meta_i + 1) % 4
            if self._meta_pin[k] is None or self._meta_pin[k].numel() < meta.size:
                self._meta_pin[k] = torch.empty(max(4096, 2 * meta.size), dtype=torch.int64).pin_memory()
            elif self._meta_ev[k] is not None:
                self._meta_ev[k].synchronize()
            self._meta_pin[k][:meta.size].copy_(torch.from_numpy(meta))
            with torch.cuda.stream(s):
                m = self._meta_pin[k][:meta.size].to(dev, non_blocking=True)   # (small: 8 (2 B + 1) + 48 B bytes)
                self._meta_ev[k] = torch.cuda.Event()
                self._meta_ev[k].record(s)
                need = lib.swat_preprocess_workspace_bytes(B, n, ks, int(row_start[-1]))
                if self._ws is None or self._ws.numel() < need:
                    self._ws = torch.empty(int(need * 1.25), dtype=torch.uint8, device=dev)
                out = torch.empty(B, 3, n, n, dtype=torch.float32, device=dev)
                mp = m.data_ptr()
                self._lib.check(lib.swat_preprocess_batch(ptr(pixels), mp, mp + 8 * (2 * B + 1), mp + 8 * B, ptr(self._lut), ptr(out), B, n,
                                                          int(desc[:, 4].max()), int(desc[:, 5].max()), ks, ptr(self._ws),
                                                          self._ws.numel(), s.cuda_stream), "swat_preprocess_batch")
                pixels.record_stream(s)
                m.record_stream(s)
        return out

    def __call__(self, images, geometry=None, stream=None):
        """images: PIL images or uint8 [H, W, 3] arrays.  geometry: optional per-image tuples as `geometry()` returns."""
        arrs = [np.ascontiguousarray(np.asarray(im.convert('RGB') if hasattr(im, 'convert') else im, dtype=np.uint8)) for im in images]
        B, n = len(arrs), self.n_px
        if B == 0:
            return torch.empty(0, 3, n, n, device=self.device)
        desc = np.zeros((B, 12), np.int32)
        offsets = np.zeros(B, np.int64)
        row_start = np.zeros(B + 1, np.int64)
        total, ks = 0, 5
        for b, a in enumerate(arrs):
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError(f"image {b}: expected [H, W, 3] uint8, got {a.shape}")
            h, w = a.shape[:2]
            g = geometry[b] if geometry is not None else self.geometry(w, h)
            x0, y0, in_w, in_h, out_w, out_h, off_x, off_y, flip = g
            if not (0 <= x0 and 0 <= y0 and 0 < in_w and 0 < in_h and x0 + in_w <= w and y0 + in_h <= h
                    and 0 <= off_x and 0 <= off_y and off_x + n <= out_w and off_y + n <= out_h):
                raise ValueError(f"image {b}: geometry {g} does not fit a {w} x {h} image / a {n}-pixel window")
            desc[b] = (w, h, x0, y0, in_w, in_h, out_w, out_h, off_x, off_y, flip, 0)
            offsets[b] = total
            total += a.size
            row_start[b + 1] = row_start[b] + in_h
            ks = max(ks, _axis_ksize(in_w, out_w), _axis_ksize(in_h, out_h))
        meta_bytes = 8 * (2 * B + 1) + 48 * B
        slot = self._turn = self._turn ^ 1
        if self._pin[slot] is None or self._pin[slot].numel() < total + 8 + meta_bytes:
            self._pin[slot] = torch.empty(int((total + 8 + meta_bytes) * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        elif self._pin_free[slot] is not None:
            self._pin_free[slot].synchronize()  # the copy that last read this buffer has finished
        hv = self._pin[slot].numpy()
        for b, a in enumerate(arrs):
            hv[offsets[b]:offsets[b] + a.size] = a.reshape(-1)
        m0 = (total + 4 + 7) // 8 * 8  # pixels, 4 bytes of padding, then offsets / row_start / desc (8-byte aligned)
        hv[total:m0] = 0
        hv[m0:m0 + meta_bytes] = np.concatenate([offsets, row_start, desc.reshape(-1).view(np.int64)]).view(np.uint8)
        dev = self.device
        lib, ptr = self._lib.lib(), self._lib.ptr
        with torch.cuda.device(dev):
            s = stream if stream is not None else torch.cuda.current_stream(dev)
            with torch.cuda.stream(s):
                pix = self._pin[slot][:m0 + meta_bytes].to(dev, non_blocking=True)
                self._pin_free[slot] = torch.cuda.Event()
                self._pin_free[slot].record(s)
                need = lib.swat_preprocess_workspace_bytes(B, n, ks, int(row_start[-1]))
                if self._ws is None or self._ws.numel() < need:
                    self._ws = torch.empty(int(need * 1.25), dtype=torch.uint8, device=dev)
                out = torch.empty(B, 3, n, n, dtype=torch.float32, device=dev)
                m = pix.data_ptr() + m0
                self._lib.check(lib.swat_preprocess_batch(ptr(pix), m, m + 8 * (2 * B + 1), m + 8 * B, ptr(self._lut), ptr(out), B, n,
                                                          int(desc[:, 4].max()), int(desc[:, 5].max()), ks, ptr(self._ws),
                                                          self._ws.numel(), s.cuda_stream), "swat_preprocess_batch")
                pix.record_stream(s)
        return out
