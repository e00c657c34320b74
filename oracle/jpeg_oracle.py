"""TEST INFRASTRUCTURE (oracle): baseline-JPEG decode as `PIL.Image.open(path).convert('RGB')` performs it — restated in numpy.

The reference decodes every training / retrieval image through Pillow (ref: utils/datasets/dataset_utils.py:130-180
`self.loader(path)` -> `Image.open(...).convert('RGB')`; retrieval/extract_mined_feature.py:103-117).  Pillow hands the file to
libjpeg(-turbo), an un-vendored dependency of the reference that IS installed in this image and on the GPU box: this restatement is
PINNED against Pillow itself (tests/test_oracle_jpeg.py decodes generated files with both), and the HIP decode path
(swat_amd/csrc/jpeg.hip) is held to it and to Pillow bit for bit.

Algorithm (libjpeg defaults, which Pillow keeps: dct_method = JDCT_ISLOW, do_fancy_upsampling = TRUE):
  * marker parse (SOI, DQT, SOF0, DHT, DRI, SOS, RSTn, EOI), Huffman entropy decode of the interleaved scan        [jdhuff.c]
  * dequantise + the 13-bit fixed-point "islow" inverse DCT, columns then rows, range limit to 0..255              [jidctint.c]
  * chroma "fancy" (triangle) upsampling h2v1 / h2v2 with libjpeg's edge rules                                      [jdsample.c]
  * YCbCr -> RGB with the 16-bit fixed-point tables                                                                 [jdcolor.c]
Only what `swat_jpeg_decode_batch` supports: SOF0 (baseline sequential, 8 bit), 1 or 3 components, luma sampling 1x1 / 2x1 / 2x2
with 1x1 chroma; anything else raises Unsupported (the product falls back to Pillow for such files)."""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                   54, 47, 55, 62, 63])


class Unsupported(Exception):
    pass


def parse(data):
    """-> dict(width, height, comps=[(id, h, v, tq)], qt={id: [64] natural order}, dc={id: table}, ac={id: table}, scan_comps=[(ci, td, ta)],
    restart_interval, scan=bytes of the entropy-coded segment (still byte-stuffed, RST markers inside))."""
    if data[:2] != b"\xff\xd8":
        raise Unsupported("not a JPEG")
    i, out = 2, dict(qt={}, dc={}, ac={}, restart_interval=0)
    while i < len(data):
        if data[i] != 0xFF:
            raise Unsupported("marker expected")
        while data[i] == 0xFF:
            i += 1
        m = data[i]
        i += 1
        if m == 0xD9:
            break
        if m == 0x01 or 0xD0 <= m <= 0xD7:
            continue
        L = (data[i] << 8) | data[i + 1]
        seg = data[i + 2:i + L]
        if m == 0xDB:
            j = 0
            while j < len(seg):
                pq, tq = seg[j] >> 4, seg[j] & 15
                j += 1
                if pq:
                    raise Unsupported("16-bit quantisation table")
                t = np.zeros(64, np.int32)
                t[ZIGZAG] = np.frombuffer(seg[j:j + 64], np.uint8)
                out["qt"][tq] = t
                j += 64
        elif m == 0xC0 or m == 0xC1:
            if seg[0] != 8:
                raise Unsupported("sample precision")
            out["height"], out["width"] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            out["comps"] = [(seg[6 + 3 * c], seg[7 + 3 * c] >> 4, seg[7 + 3 * c] & 15, seg[8 + 3 * c]) for c in range(seg[5])]
            if m == 0xC1:
                raise Unsupported("extended sequential")
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise Unsupported("not baseline (SOF%d)" % (m - 0xC0))
        elif m == 0xC4:
            j = 0
            while j < len(seg):
                tc, th = seg[j] >> 4, seg[j] & 15
                counts = list(seg[j + 1:j + 17])
                n = sum(counts)
                vals = list(seg[j + 17:j + 17 + n])
                (out["ac"] if tc else out["dc"])[th] = (counts, vals)
                j += 17 + n
        elif m == 0xDD:
            out["restart_interval"] = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            ns = seg[0]
            out["scan_comps"] = [(seg[1 + 2 * c], seg[2 + 2 * c] >> 4, seg[2 + 2 * c] & 15) for c in range(ns)]
            if (seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns]) != (0, 63, 0):
                raise Unsupported("spectral selection / successive approximation")
            out["scan"] = data[i + L:]
            break
        i += L
    if "comps" not in out or "scan" not in out:
        raise Unsupported("no frame / scan")
    nc = len(out["comps"])
    if nc not in (1, 3) or len(out["scan_comps"]) != nc:
        raise Unsupported("component count")
    if nc == 3:
        (_, h0, v0, _), (_, h1, v1, _), (_, h2, v2, _) = out["comps"]
        if (h1, v1, h2, v2) != (1, 1, 1, 1) or (h0, v0) not in ((1, 1), (2, 1), (2, 2)):
            raise Unsupported("sampling factors")
    return out


def _huff_lookup(counts, vals):
    """code -> (length, value) for codes of each length (canonical Huffman, JPEG Annex C)."""
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            table[(length, code)] = vals[k]
            code += 1
            k += 1
        code <<= 1
    return table


class _Bits:
    def __init__(self, data):
        self.d, self.i, self.acc, self.n, self.marker = data, 0, 0, 0, False

    def bit(self):
        if self.n == 0:
            b = 0  # past the end of the segment / in front of a marker libjpeg feeds zero bits
            if not self.marker and self.i < len(self.d):
                b = self.d[self.i]
                if b == 0xFF:
                    nxt = self.d[self.i + 1] if self.i + 1 < len(self.d) else 0xD9
                    if nxt == 0:
                        self.i += 2  # a stuffed 0xFF data byte
                    else:
                        self.marker, b = True, 0
                else:
                    self.i += 1
            self.acc, self.n = b, 8
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def restart(self):
        """Drop the bits left in the current byte and step over the RSTn marker."""
        self.n, self.marker = 0, False
        while self.i + 1 < len(self.d) and not (self.d[self.i] == 0xFF and 0xD0 <= self.d[self.i + 1] <= 0xD7):
            self.i += 1
        self.i += 2


def _decode_symbol(br, table):
    code = 0
    for length in range(1, 17):
        code = (code << 1) | br.bit()
        v = table.get((length, code))
        if v is not None:
            return v
    raise ValueError("bad Huffman code")


def _extend(v, s):
    return v if v >= (1 << (s - 1)) else v - (1 << s) + 1


def coefficients(hdr):
    """Entropy decode -> per component an int16 array [blocks_h, blocks_w, 64] in NATURAL order (still quantised), with the block grid
    padded to whole MCUs."""
    W, H, comps = hdr["width"], hdr["height"], hdr["comps"]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mx, my = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    out = [np.zeros((my * v, mx * h, 64), np.int16) for (_, h, v, _) in comps]
    dct, act = {k: _huff_lookup(*t) for k, t in hdr["dc"].items()}, {k: _huff_lookup(*t) for k, t in hdr["ac"].items()}
    sel = {cid: (td, ta) for cid, td, ta in hdr["scan_comps"]}
    br, pred, ri, left = _Bits(hdr["scan"]), [0] * len(comps), hdr["restart_interval"], hdr["restart_interval"]
    for m in range(mx * my):
        if ri and left == 0:
            br.restart()
            pred, left = [0] * len(comps), ri
        r, c = divmod(m, mx)
        for ci, (cid, h, v, _) in enumerate(comps):
            td, ta = sel[cid]
            for by in range(v):
                for bx in range(h):
                    blk = np.zeros(64, np.int32)
                    s = _decode_symbol(br, dct[td])
                    pred[ci] += _extend(br.bits(s), s) if s else 0
                    blk[0] = pred[ci]
                    k = 1
                    while k < 64:
                        rs = _decode_symbol(br, act[ta])
                        run, s = rs >> 4, rs & 15
                        if s == 0:
                            if run != 15:
                                break
                            k += 16
                            continue
                        k += run
                        blk[ZIGZAG[k]] = _extend(br.bits(s), s)
                        k += 1
                    out[ci][r * v + by, c * h + bx] = blk
        left -= 1
    return out


def idct_islow(coef, quant):
    """[..., 64] quantised coefficients (natural order), quant [64] -> [..., 8, 8] uint8 samples: jpeg_idct_islow, bit for bit."""
    F = lambda x: int(x * 8192 + 0.5)
    c = coef.astype(np.int64) * quant.astype(np.int64)
    c = c.reshape(c.shape[:-1] + (8, 8))

    def pass1d(x, shift, pre):
        """x [..., 8] along the last axis (input index k) -> 8 outputs; pre = left shift of the even part (CONST_BITS)"""
        z2, z3 = x[..., 2], x[..., 6]
        z1 = (z2 + z3) * F(0.541196100)
        tmp2 = z1 + z3 * (-F(1.847759065))
        tmp3 = z1 + z2 * F(0.765366865)
        tmp0 = (x[..., 0] + x[..., 4]) << pre
        tmp1 = (x[..., 0] - x[..., 4]) << pre
        tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
        t0, t1, t2, t3 = x[..., 7], x[..., 5], x[..., 3], x[..., 1]
        z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
        z5 = (z3 + z4) * F(1.175875602)
        t0, t1, t2, t3 = t0 * F(0.298631336), t1 * F(2.053119869), t2 * F(3.072711026), t3 * F(1.501321110)
        z1, z2 = z1 * (-F(0.899976223)), z2 * (-F(2.562915447))
        z3, z4 = z3 * (-F(1.961570560)) + z5, z4 * (-F(0.390180644)) + z5
        t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
        rnd = 1 << (shift - 1)
        outs = [tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3]
        return np.stack([(o + rnd) >> shift for o in outs], axis=-1)

    # pass 1: columns (input index = row k, for each column) -> workspace scaled by 2^PASS1_BITS
    ws = pass1d(np.swapaxes(c, -1, -2), 13 - 2, 13)          # [..., col, out_row]
    ws = np.swapaxes(ws, -1, -2)                              # [..., row, col]
    # pass 2: rows
    res = pass1d(ws, 13 + 2 + 3, 13)
    return np.clip(res + 128, 0, 255).astype(np.uint8)


def _plane(blocks):
    bh, bw = blocks.shape[:2]
    return blocks.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def _upsample_h2v1(p, cw, W):
    p = p[:, :cw].astype(np.int32)
    out = np.empty((p.shape[0], 2 * cw), np.int32)
    left = np.concatenate([p[:, :1], p[:, :-1]], 1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, 2 * cw - 1] = p[:, cw - 1]
    return out[:, :W]


def _upsample_h2v2(p, cw, ch, W, H):
    p = p[:ch, :cw].astype(np.int32)
    above = np.concatenate([p[:1], p[:-1]], 0)
    below = np.concatenate([p[1:], p[-1:]], 0)
    out = np.empty((2 * ch, 2 * cw), np.int32)
    for parity, other in ((0, above), (1, below)):
        s = 3 * p + other                                    # thiscolsum per column
        last = np.concatenate([s[:, :1], s[:, :-1]], 1)
        nxt = np.concatenate([s[:, 1:], s[:, -1:]], 1)
        ev = (3 * s + last + 8) >> 4
        od = (3 * s + nxt + 7) >> 4
        ev[:, 0] = (4 * s[:, 0] + 8) >> 4
        od[:, cw - 1] = (4 * s[:, cw - 1] + 7) >> 4
        out[parity::2, 0::2] = ev
        out[parity::2, 1::2] = od
    return out[:H, :W]


def ycc_to_rgb(y, cb, cr):
    FIX = lambda x: int(x * 65536 + 0.5)
    y, cb, cr = y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((FIX(1.40200) * cr + 32768) >> 16)
    b = y + ((FIX(1.77200) * cb + 32768) >> 16)
    g = y + ((-FIX(0.34414) * cb + 32768 - FIX(0.71414) * cr) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def decode(data):
    """bytes of a baseline JPEG -> [H, W, 3] uint8, equal to np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))."""
    hdr = parse(bytes(data))
    W, H, comps = hdr["width"], hdr["height"], hdr["comps"]
    coefs = coefficients(hdr)
    planes = [_plane(idct_islow(c, hdr["qt"][comps[i][3]])) for i, c in enumerate(coefs)]
    if len(comps) == 1:
        y = planes[0][:H, :W]
        return np.stack([y, y, y], -1)
    (h0, v0) = comps[0][1:3]
    y = planes[0][:H, :W]
    if (h0, v0) == (1, 1):
        cb, cr = planes[1][:H, :W], planes[2][:H, :W]
    elif (h0, v0) == (2, 1):
        cw = -(-W // 2)
        cb, cr = _upsample_h2v1(planes[1][:H], cw, W), _upsample_h2v1(planes[2][:H], cw, W)
    else:
        cw, ch = -(-W // 2), -(-H // 2)
        cb, cr = _upsample_h2v2(planes[1], cw, ch, W, H), _upsample_h2v2(planes[2], cw, ch, W, H)
    return ycc_to_rgb(y, cb, cr)
