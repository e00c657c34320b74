// swat_amd/csrc/jpeg.hip — baseline-JPEG decode for the input pipeline: Huffman entropy decode on host threads, everything after it
// (dequantisation, inverse DCT, chroma upsampling, colour conversion) on the GPU, straight into the byte layout
// swat_preprocess_batch reads.
//
// ref: utils/datasets/dataset_utils.py:130-180 (`self.loader(path)` = PIL `Image.open(path).convert('RGB')` per sample, in 8 worker
// processes: utils/dataloader.py:107-135), retrieval/extract_mined_feature.py:103-117.  Pillow hands the file to libjpeg(-turbo) with
// its defaults (JDCT_ISLOW, fancy upsampling); the arithmetic below is that library's (jidctint.c, jdsample.c, jdcolor.c), restated in
// oracle/jpeg_oracle.py and pinned against Pillow itself — output bytes are identical to Pillow's.
//
// Why split it here.  With crop / resize / normalise on the GPU (csrc/preprocess.hip) the loaders' remaining per-image cost is the
// decode: 2-10 ms of one host core per image, i.e. 20-100 cores to feed one GPU at 10 k img/s.  Of that, the serial part — the
// Huffman bit stream — is ~1/4; the inverse DCT, upsampling and colour conversion are independent per 8 x 8 block / per pixel.
// Host: marker parse + entropy decode into int16 coefficient blocks (natural order, quantised) in pinned memory, one std::thread per
// share of the batch; one hipMemcpyAsync; two launches.  Supported: SOF0 (baseline, 8 bit), 1 or 3 components, luma sampling
// 1x1 / 2x1 / 2x2 with 1x1 chroma (4:4:4, 4:2:2, 4:2:0), restart intervals.  Anything else is reported per file (status 1) and the
// caller decodes that file with Pillow.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "common.hpp"
#include "../../include/swat_hip.h"

namespace swat {

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---- host: header ---------------------------------------------------------------------------------------------------------------
struct JpegHuff {
    bool present = false;
    uint8_t look_len[512];   // 9-bit prefix -> code length (0: longer than 9 bits)
    uint8_t look_sym[512];
    int32_t maxcode[18];     // per length: largest code of that length (-1: none)
    int32_t valoff[17];      // vals index of the first code of that length minus that code
    uint8_t vals[256];
};
struct JpegFrame {
    int W = 0, H = 0, nc = 0;
    int h[3] = {0, 0, 0}, v[3] = {0, 0, 0}, tq[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0}, id[3] = {0, 0, 0};
    uint16_t qt[4][64];
    bool qt_ok[4] = {false, false, false, false};
    JpegHuff dc[4], ac[4];
    int restart = 0;
    const uint8_t* scan = nullptr;
    size_t scan_len = 0;
    int mx = 0, my = 0;   // MCUs
};

// -> false: the code counts are not a prefix code (libjpeg's JERR_BAD_HUFF_TABLE: a code of length `len` must be < 2^len; an
// over-subscribed table would index past the 9-bit lookup) or, for a DC table, a category above 15
static bool build_huff(JpegHuff& t, const uint8_t* counts, const uint8_t* vals, int n, bool is_dc) {
    t.present = false;
    if (n < 0 || n > 256) return false;
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            code += counts[len - 1];
            k += counts[len - 1];
            if (code > (1 << len)) return false;
            code <<= 1;
        }
        if (k != n) return false;
        if (is_dc)
            for (int i = 0; i < n; ++i)
                if (vals[i] > 15) return false;
    }
    t.present = true;
    memcpy(t.vals, vals, (size_t)n);
    memset(t.look_len, 0, sizeof(t.look_len));
    memset(t.look_sym, 0, sizeof(t.look_sym));
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
        t.valoff[len] = k - code;
        for (int i = 0; i < counts[len - 1]; ++i, ++code, ++k) {
            if (len <= 9) {
                const int lo = code << (9 - len), cnt = 1 << (9 - len);
                for (int j = 0; j < cnt; ++j) { t.look_len[lo + j] = (uint8_t)len; t.look_sym[lo + j] = vals[k]; }
            }
        }
        t.maxcode[len] = counts[len - 1] ? code - 1 : -1;
        code <<= 1;
    }
    t.maxcode[17] = 0x7fffffff;
    return true;
}

// Pillow refuses to decode beyond 2 x Image.MAX_IMAGE_PIXELS (decompression-bomb guard) and warns beyond 1 x: such a file is left to
// Pillow (status 1) so that its rule, not a 13 GB staging allocation, decides
static const int64_t kMaxPixels = 89478485;

// -> 0 ok, 1 not a file this path decodes, 2 corrupt
static int parse_jpeg(const uint8_t* d, size_t n, JpegFrame& f) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return 1;
    size_t i = 2;
    bool have_frame = false, jfif = false, adobe = false;
    int adobe_transform = -1;
    while (i + 3 < n) {
        if (d[i] != 0xFF) return 2;
        while (i < n && d[i] == 0xFF) ++i;
        if (i >= n) return 2;
        const int m = d[i++];
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (i + 2 > n) return 2;
        const size_t L = ((size_t)d[i] << 8) | d[i + 1];
        if (L < 2 || i + L > n) return 2;
        const uint8_t* s = d + i + 2;
        const size_t sl = L - 2;
        if (m == 0xDB) {
            size_t j = 0;
            while (j < sl) {
                const int pq = s[j] >> 4, tq = s[j] & 15;
                ++j;
                if (pq || tq > 3 || j + 64 > sl) return pq ? 1 : 2;
                for (int k = 0; k < 64; ++k) f.qt[tq][kZigzag[k]] = s[j + k];
                f.qt_ok[tq] = true;
                j += 64;
            }
        } else if (m == 0xC0) {
            if (sl < 6 || s[0] != 8) return 1;
            f.H = (s[1] << 8) | s[2];
            f.W = (s[3] << 8) | s[4];
            f.nc = s[5];
            if ((f.nc != 1 && f.nc != 3) || sl < (size_t)(6 + 3 * f.nc) || f.W <= 0 || f.H <= 0) return 1;
            if ((int64_t)f.W * f.H > kMaxPixels) return 1;
            for (int c = 0; c < f.nc; ++c) {
                f.h[c] = s[7 + 3 * c] >> 4;
                f.v[c] = s[7 + 3 * c] & 15;
                f.tq[c] = s[8 + 3 * c];
                f.td[c] = s[6 + 3 * c];   // component id, matched against the scan header below
                f.id[c] = s[6 + 3 * c];
                if (f.tq[c] > 3) return 2;
            }
            have_frame = true;
        } else if (m >= 0xC1 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return 1;   // extended / progressive / lossless / arithmetic
        } else if (m == 0xC4) {
            size_t j = 0;
            while (j + 17 <= sl) {
                const int tc = s[j] >> 4, th = s[j] & 15;
                int cnt = 0;
                for (int k = 0; k < 16; ++k) cnt += s[j + 1 + k];
                if (tc > 1 || th > 3 || cnt > 256 || j + 17 + cnt > sl) return 2;
                if (!build_huff(tc ? f.ac[th] : f.dc[th], s + j + 1, s + j + 17, cnt, tc == 0)) return 2;
                j += 17 + cnt;
            }
            if (j != sl) return 2;   // a truncated table at the segment's end
        } else if (m == 0xE0) {
            if (sl >= 5 && memcmp(s, "JFIF", 5) == 0) jfif = true;
        } else if (m == 0xEE) {
            if (sl >= 12 && memcmp(s, "Adobe", 5) == 0) {
                adobe = true;
                adobe_transform = s[11];
            }
        } else if (m == 0xDD) {
            if (sl < 2) return 2;
            f.restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!have_frame || sl < 1 || s[0] != f.nc || sl < (size_t)(4 + 2 * f.nc)) return have_frame ? 1 : 2;
            int ids[3] = {f.td[0], f.td[1], f.td[2]};
            for (int c = 0; c < f.nc; ++c) {
                if (s[1 + 2 * c] != ids[c]) return 1;   // components out of frame order: leave it to the library
                f.td[c] = s[2 + 2 * c] >> 4;
                f.ta[c] = s[2 + 2 * c] & 15;
                if (f.td[c] > 3 || f.ta[c] > 3) return 2;
            }
            if (s[1 + 2 * f.nc] != 0 || s[2 + 2 * f.nc] != 63 || s[3 + 2 * f.nc] != 0) return 1;
            f.scan = d + i + L;
            f.scan_len = n - (i + L);
            break;
        }
        i += L;
    }
    if (!have_frame || !f.scan) return 1;
    if (f.nc == 3) {
        // colour space as libjpeg's default_decompress_parms picks it: JFIF -> YCbCr; else an Adobe marker decides (transform 0 = the
        // planes are RGB already); else the component ids ('R','G','B' = RGB).  Everything that is not YCbCr goes to Pillow.
        if (!jfif) {
            if (adobe) {
                if (adobe_transform == 0) return 1;
            } else if (f.id[0] == 'R' && f.id[1] == 'G' && f.id[2] == 'B') {
                return 1;
            }
        }
        if (f.h[1] != 1 || f.v[1] != 1 || f.h[2] != 1 || f.v[2] != 1) return 1;
        // (jdsample.c: a chroma row of <= 2 samples is replicated, not interpolated)
        if (f.h[0] == 2 && (f.W + 1) / 2 <= 2) return 1;
        if (!((f.h[0] == 1 && f.v[0] == 1) || (f.h[0] == 2 && f.v[0] == 1) || (f.h[0] == 2 && f.v[0] == 2))) return 1;
    } else {
        f.h[0] = f.v[0] = 1;   // a single component is never interleaved: its sampling factors do not shape the MCU
    }
    for (int c = 0; c < f.nc; ++c)
        if (!f.qt_ok[f.tq[c]] || !f.dc[f.td[c]].present || !f.ac[f.ta[c]].present) return 2;
    f.mx = (f.W + 8 * f.h[0] - 1) / (8 * f.h[0]);
    f.my = (f.H + 8 * f.v[0] - 1) / (8 * f.v[0]);
    return 0;
}

// ---- host: entropy decode ---------------------------------------------------------------------------------------------------------
struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int n = 0;
    bool marker = false;
    void fill() {   // keep >= 32 bits; in front of a marker / past the end: zero bits (libjpeg's behaviour)
        while (n <= 56) {
            unsigned b = 0;
            if (!marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    const unsigned nx = p + 1 < end ? p[1] : 0xD9;
                    if (nx == 0) p += 2;
                    else { marker = true; b = 0; }
                } else {
                    ++p;
                }
            }
            acc |= (uint64_t)b << (56 - n);
            n += 8;
        }
    }
    inline unsigned peek(int k) { return (unsigned)(acc >> (64 - k)); }
    inline void drop(int k) { acc <<= k; n -= k; }
    void restart() {
        acc = 0; n = 0; marker = false;
        while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
        p += 2;
        if (p > end) p = end;
    }
};

static inline int huff_symbol(BitReader& br, const JpegHuff& t) {
    if (br.n < 16) br.fill();
    const unsigned look = br.peek(9);
    int len = t.look_len[look];
    if (len) {
        br.drop(len);
        return t.look_sym[look];
    }
    int code = (int)br.peek(10);
    len = 10;
    while (len <= 16 && code > t.maxcode[len]) {
        ++len;
        code = (int)br.peek(len);
    }
    if (len > 16) { br.drop(16); return 0; }   // corrupt code: libjpeg warns and takes zero
    br.drop(len);
    return t.vals[(code + t.valoff[len]) & 255];
}
static inline int receive_extend(BitReader& br, int s) {
    if (br.n < s) br.fill();
    const int v = (int)br.peek(s);
    br.drop(s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

// coefficient blocks of component c: [my * v][mx * h][64] int16, natural order, components back to back
static void entropy_decode(const JpegFrame& f, int16_t* coef) {
    int16_t* base[3];
    int bw[3];
    size_t off = 0;
    for (int c = 0; c < f.nc; ++c) {
        base[c] = coef + off;
        bw[c] = f.mx * f.h[c];
        off += (size_t)bw[c] * f.my * f.v[c] * 64;
    }
    memset(coef, 0, off * sizeof(int16_t));
    BitReader br{f.scan, f.scan + f.scan_len};
    int pred[3] = {0, 0, 0};
    int left = f.restart;
    for (int my = 0; my < f.my; ++my)
        for (int mx = 0; mx < f.mx; ++mx) {
            if (f.restart && left == 0) {
                br.restart();
                pred[0] = pred[1] = pred[2] = 0;
                left = f.restart;
            }
            for (int c = 0; c < f.nc; ++c) {
                const JpegHuff& dc = f.dc[f.td[c]];
                const JpegHuff& ac = f.ac[f.ta[c]];
                for (int by = 0; by < f.v[c]; ++by)
                    for (int bx = 0; bx < f.h[c]; ++bx) {
                        int16_t* blk = base[c] + ((size_t)(my * f.v[c] + by) * bw[c] + (mx * f.h[c] + bx)) * 64;
                        int s = huff_symbol(br, dc) & 15;
                        if (s) pred[c] += receive_extend(br, s);
                        blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64;) {
                            const int rs = huff_symbol(br, ac);
                            const int run = rs >> 4;
                            s = rs & 15;
                            if (s == 0) {
                                if (run != 15) break;
                                k += 16;
                                continue;
                            }
                            k += run;
                            const int val = receive_extend(br, s);
                            if (k < 64) blk[kZigzag[k]] = (int16_t)val;
                            ++k;
                        }
                    }
            }
            --left;
        }
}

// ---- device ------------------------------------------------------------------------------------------------------------------------
struct JpegDesc {        // one per image, in device memory
    int32_t W, H, nc, hs, vs;           // hs, vs: luma sampling factors (1 or 2)
    int32_t bw[3], bh[3];               // block grid per component (whole MCUs)
    int64_t coef_off[3];                // int16 index into the batch's coefficient buffer
    int64_t plane_off[3];               // byte offset into the plane workspace: [bh * 8][bw * 8] samples
    int64_t out_off;                    // byte offset of the image in the pixel buffer (HWC RGB)
    uint16_t qt[3][64];                 // quantisation table per component, natural order
    int32_t ok;                         // 0: skip this image (the caller decodes it elsewhere)
};

// jpeg_idct_islow (jidctint.c): 13-bit constants, two passes, bit for bit.  One thread per 8 x 8 block.
#define JF_0_298631336 2446
#define JF_0_390180644 3196
#define JF_0_541196100 4433
#define JF_0_765366865 6270
#define JF_0_899976223 7373
#define JF_1_175875602 9633
#define JF_1_501321110 12299
#define JF_1_847759065 15137
#define JF_1_961570560 16069
#define JF_2_053119869 16819
#define JF_2_562915447 20995
#define JF_3_072711026 25172

template <int SHIFT>
__device__ __forceinline__ void idct8(const int (&x)[8], int (&o)[8]) {
    int z2 = x[2], z3 = x[6];
    int z1 = (z2 + z3) * JF_0_541196100;
    const int tmp2 = z1 + z3 * (-JF_1_847759065);
    const int tmp3 = z1 + z2 * JF_0_765366865;
    const int tmp0 = (x[0] + x[4]) << 13, tmp1 = (x[0] - x[4]) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int t0 = x[7], t1 = x[5], t2 = x[3], t3 = x[1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
    int z4 = t1 + t3;
    const int z5 = (z3 + z4) * JF_1_175875602;
    t0 *= JF_0_298631336; t1 *= JF_2_053119869; t2 *= JF_3_072711026; t3 *= JF_1_501321110;
    z1 *= -JF_0_899976223; z2 *= -JF_2_562915447;
    z3 = z3 * (-JF_1_961570560) + z5;
    z4 = z4 * (-JF_0_390180644) + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    constexpr int R = 1 << (SHIFT - 1);
    o[0] = (tmp10 + t3 + R) >> SHIFT; o[7] = (tmp10 - t3 + R) >> SHIFT;
    o[1] = (tmp11 + t2 + R) >> SHIFT; o[6] = (tmp11 - t2 + R) >> SHIFT;
    o[2] = (tmp12 + t1 + R) >> SHIFT; o[5] = (tmp12 - t1 + R) >> SHIFT;
    o[3] = (tmp13 + t0 + R) >> SHIFT; o[4] = (tmp13 - t0 + R) >> SHIFT;
}

__global__ __launch_bounds__(128) void jpeg_idct_kernel(const JpegDesc* __restrict__ desc, const int16_t* __restrict__ coef,
                                                        uint8_t* __restrict__ planes) {
    const int img = blockIdx.y / 3, c = blockIdx.y % 3;
    const JpegDesc& D = desc[img];
    if (!D.ok || c >= D.nc) return;
    const int nb = D.bw[c] * D.bh[c];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const int16_t* src = coef + D.coef_off[c] + (int64_t)b * 64;
    int ws[8][8];   // [row][col]
    // pass 1: columns.  x[k] = coefficient row k of this column, dequantised
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int x[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (int)src[8 * k + col] * (int)D.qt[c][8 * k + col];
        idct8<13 - 2>(x, o);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r][col] = o[r];
    }
    // pass 2: rows, + 128 and the range limit
    const int by = b / D.bw[c], bx = b % D.bw[c];
    uint8_t* dst = planes + D.plane_off[c] + ((int64_t)by * 8) * (D.bw[c] * 8) + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int o[8];
        idct8<13 + 2 + 3>(ws[r], o);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int a = o[k] + 128, e = o[4 + k] + 128;
            a = a < 0 ? 0 : a > 255 ? 255 : a;
            e = e < 0 ? 0 : e > 255 ? 255 : e;
            lo |= (uint32_t)a << (8 * k);
            hi |= (uint32_t)e << (8 * k);
        }
        *reinterpret_cast<uint2*>(dst + (int64_t)r * (D.bw[c] * 8)) = make_uint2(lo, hi);
    }
}

// Chroma sample for output pixel (x, y): libjpeg's "fancy" triangle upsampling (jdsample.c h2v1_fancy_upsample /
// h2v2_fancy_upsample), with its edge rules: the first / last column copy (h2v1) or weigh 4:0 (h2v2), the row above the first and
// below the last REAL chroma row is that row itself.
__device__ __forceinline__ int jpeg_chroma(const uint8_t* __restrict__ p, int stride, int cw, int ch, int hs, int vs, int x, int y) {
    if (hs == 1) return p[(int64_t)y * stride + x];
    const int i = x >> 1;
    if (vs == 1) {
        const uint8_t* row = p + (int64_t)y * stride;
        const int v = row[i];
        if (x & 1) return i == cw - 1 ? v : (3 * v + row[i + 1] + 2) >> 2;
        return i == 0 ? v : (3 * v + row[i - 1] + 1) >> 2;
    }
    const int cy = y >> 1;
    int oy = (y & 1) ? cy + 1 : cy - 1;
    oy = oy < 0 ? 0 : oy >= ch ? ch - 1 : oy;
    const uint8_t* r0 = p + (int64_t)cy * stride;
    const uint8_t* r1 = p + (int64_t)oy * stride;
    const int s = 3 * r0[i] + r1[i];
    if (x & 1) {
        if (i == cw - 1) return (4 * s + 7) >> 4;
        return (3 * s + (3 * r0[i + 1] + r1[i + 1]) + 7) >> 4;
    }
    if (i == 0) return (4 * s + 8) >> 4;
    return (3 * s + (3 * r0[i - 1] + r1[i - 1]) + 8) >> 4;
}

// jdcolor.c ycc_rgb_convert: 16-bit fixed point, the rounding constant inside the Cb term of green
__global__ __launch_bounds__(256) void jpeg_color_kernel(const JpegDesc* __restrict__ desc, const uint8_t* __restrict__ planes,
                                                         uint8_t* __restrict__ pixels) {
    const JpegDesc& D = desc[blockIdx.y];
    if (!D.ok) return;
    const int64_t npx = (int64_t)D.W * D.H;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npx; t += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(t / D.W), x = (int)(t % D.W);
        const int Y = planes[D.plane_off[0] + (int64_t)y * (D.bw[0] * 8) + x];
        int r = Y, g = Y, b = Y;
        if (D.nc == 3) {
            const int cw = (D.W + D.hs - 1) / D.hs, ch = (D.H + D.vs - 1) / D.vs;
            const int cb = jpeg_chroma(planes + D.plane_off[1], D.bw[1] * 8, cw, ch, D.hs, D.vs, x, y) - 128;
            const int cr = jpeg_chroma(planes + D.plane_off[2], D.bw[2] * 8, cw, ch, D.hs, D.vs, x, y) - 128;
            r = Y + ((91881 * cr + 32768) >> 16);
            b = Y + ((116130 * cb + 32768) >> 16);
            g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
            r = r < 0 ? 0 : r > 255 ? 255 : r;
            g = g < 0 ? 0 : g > 255 ? 255 : g;
            b = b < 0 ? 0 : b > 255 ? 255 : b;
        }
        uint8_t* o = pixels + D.out_off + 3 * t;
        o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b;
    }
}

// ---- per-process staging (pinned host + device), grown on demand -------------------------------------------------------------------
struct JpegArena {
    // the pinned staging buffer is DOUBLE: the entropy decode of batch i + 1 fills one half while the upload of batch i (queued on the
    // stream, possibly behind a training step's kernels) still reads the other — with one buffer the host stood still until the
    // GPU had reached that copy, and the loader could not run ahead of the step
    void* hosts[2] = {nullptr, nullptr};
    size_t hosts_bytes[2] = {0, 0};
    hipEvent_t copies[2] = {nullptr, nullptr};
    bool pendings[2] = {false, false};
    int turn = 0;
    void* host = nullptr;
    void* dev = nullptr;
    void* planes = nullptr;
    size_t host_bytes = 0, dev_bytes = 0, plane_bytes = 0;
    hipEvent_t copied = nullptr;   // the last H2D copy out of `host` has finished
    hipEvent_t done = nullptr;     // the last call's kernels (they read `dev` and `planes`) have finished
    hipStream_t last = nullptr;    // ... on this stream
    bool pending = false, any = false;
};
static JpegArena g_jpeg_dev[16];   // one staging arena per device (the calls of a process are serialised by g_jpeg_call)

static int grow(void** p, size_t* have, size_t need, bool pinned) {
    if (*have >= need) return 0;
    if (*p) {
        if (hipDeviceSynchronize() != hipSuccess) return -1;   // (rare: the buffers only grow)
        if ((pinned ? hipHostFree(*p) : hipFree(*p)) != hipSuccess) return -1;
        *p = nullptr;
        *have = 0;
    }
    need += need / 4 + 4096;
    if ((pinned ? hipHostMalloc(p, need, hipHostMallocDefault) : hipMalloc(p, need)) != hipSuccess) return -1;
    *have = need;
    return 0;
}

}  // namespace swat
using namespace swat;

SWAT_API int swat_jpeg_info(const void* file, int64_t nbytes, int32_t* width, int32_t* height) {
    SWAT_REQUIRE(file && width && height && nbytes >= 0, "swat_jpeg_info: null pointer");
    JpegFrame f;
    const int rc = parse_jpeg((const uint8_t*)file, (size_t)nbytes, f);
    if (rc != 0) return rc;
    *width = f.W;
    *height = f.H;
    return 0;
}

// Host half alone (tests without a GPU; also what a caller needs to size things): coefficient blocks of every component, int16,
// natural order, [component][block row][block column][64] with the block grids padded to whole MCUs.  dims (int32[8]): W, H, nc,
// luma h, luma v, MCUs across, MCUs down, 0.  coef may be null (dims only).  -> 0 / 1 unsupported / 2 corrupt / -1 capacity
SWAT_API int swat_jpeg_entropy_decode(const void* file, int64_t nbytes, int32_t* dims, void* coef_i16, int64_t coef_capacity) {
    SWAT_REQUIRE(file && dims && nbytes >= 0, "swat_jpeg_entropy_decode: null pointer");
    JpegFrame f;
    const int rc = parse_jpeg((const uint8_t*)file, (size_t)nbytes, f);
    if (rc != 0) return rc;
    dims[0] = f.W; dims[1] = f.H; dims[2] = f.nc; dims[3] = f.h[0]; dims[4] = f.v[0]; dims[5] = f.mx; dims[6] = f.my; dims[7] = 0;
    if (!coef_i16) return 0;
    int64_t need = 0;
    for (int c = 0; c < f.nc; ++c) need += (int64_t)f.mx * f.h[c] * f.my * f.v[c] * 64;
    SWAT_REQUIRE(coef_capacity >= need, "swat_jpeg_entropy_decode: %lld coefficients needed, room for %lld", (long long)need, (long long)coef_capacity);
    entropy_decode(f, (int16_t*)coef_i16);
    return 0;
}

namespace swat {
// Worker threads of the entropy decode: kept between calls (creating and joining 15 threads cost ~0.3 ms of a 1.9 ms batch).  A call
// publishes a job (a function every participating thread runs; the function itself hands out the files), wakes `want` workers and
// works along; the last worker out signals the caller.  A child process after fork() has no workers: the pool notices (pid) and
// starts over.
struct JpegPool {
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    const std::function<void()>* job = nullptr;
    unsigned long long gen = 0;
    int want = 0, taken = 0, running = 0;
    bool stop = false;
    int pid = 0;
    void worker() {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || (gen != seen && taken < want); });
            if (stop) return;
            seen = gen;
            ++taken;
            const std::function<void()>* j = job;
            lk.unlock();
            (*j)();
            lk.lock();
            if (--running == 0) cv_done.notify_one();
        }
    }
    void run(const std::function<void()>& fn, int helpers) {   // fn on `helpers` workers and on the caller
        if (helpers > 0) {
            std::unique_lock<std::mutex> lk(m);
            const int me = (int)getpid();
            if (pid != me) {   // first use, or a forked child: the parent's threads do not exist here
                for (auto& t : threads) t.detach();
                threads.clear();
                pid = me;
            }
            while ((int)threads.size() < helpers) threads.emplace_back([this] { worker(); });
            job = &fn;
            want = helpers;
            taken = 0;
            running = helpers;
            ++gen;
            cv_work.notify_all();
        }
        fn();
        if (helpers > 0) {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return running == 0; });
            job = nullptr;
        }
    }
    ~JpegPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_work.notify_all();
        if (pid == (int)getpid())
            for (auto& t : threads) t.join();
        else
            for (auto& t : threads) t.detach();
    }
};
static JpegPool g_jpeg_pool;
static std::mutex g_jpeg_call;   // one decode call at a time uses the pool and the staging arena
}  // namespace swat

// n files -> RGB bytes (HWC) on the device, image i at byte offsets[i] of pixels_u8 (the layout swat_preprocess_batch reads; the
// caller sized it from swat_jpeg_info).  status[i] (host): 0 decoded, 1 not a baseline JPEG this path decodes, 2 corrupt — nothing
// is written for such a file.  threads: host threads for the entropy decode (<= 0: one per 2 files; at most 32), kept between calls.
SWAT_API int swat_jpeg_decode_batch(const void* const* files, const int64_t* nbytes, int n, void* pixels_u8, const int64_t* offsets,
                                    int32_t* status, int threads, swat_stream_t stream) {
    SWAT_REQUIRE(files && nbytes && pixels_u8 && offsets && status && n >= 0, "swat_jpeg_decode_batch: null pointer");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> one_call(swat::g_jpeg_call);
    hipStream_t s = (hipStream_t)stream;
    std::vector<JpegFrame> fr((size_t)n);
    std::vector<JpegDesc> desc((size_t)n);
    size_t coef_total = 0, plane_total = 0;
    int max_blocks = 0;
    int64_t max_px = 0;
    for (int i = 0; i < n; ++i) {
        JpegDesc& D = desc[(size_t)i];
        memset(&D, 0, sizeof(D));
        status[i] = files[i] ? parse_jpeg((const uint8_t*)files[i], (size_t)nbytes[i], fr[(size_t)i]) : 2;
        if (status[i] != 0) continue;
        const JpegFrame& f = fr[(size_t)i];
        D.ok = 1; D.W = f.W; D.H = f.H; D.nc = f.nc; D.hs = f.h[0]; D.vs = f.v[0]; D.out_off = offsets[i];
        for (int c = 0; c < f.nc; ++c) {
            D.bw[c] = f.mx * f.h[c];
            D.bh[c] = f.my * f.v[c];
            D.coef_off[c] = (int64_t)coef_total;
            D.plane_off[c] = (int64_t)plane_total;
            const size_t nb = (size_t)D.bw[c] * D.bh[c];
            coef_total += nb * 64;
            plane_total += nb * 64;
            if ((int)nb > max_blocks) max_blocks = (int)nb;
            memcpy(D.qt[c], f.qt[f.tq[c]], sizeof(D.qt[c]));
        }
        if ((int64_t)f.W * f.H > max_px) max_px = (int64_t)f.W * f.H;
    }
    if (coef_total == 0) return 0;
    const size_t desc_bytes = ((size_t)n * sizeof(JpegDesc) + 255) / 256 * 256;
    const size_t up_bytes = desc_bytes + coef_total * sizeof(int16_t);
    int dev_i = 0;
    SWAT_HIP(hipGetDevice(&dev_i));
    SWAT_REQUIRE(dev_i >= 0 && dev_i < 16, "swat_jpeg_decode_batch: device %d out of range", dev_i);
    JpegArena& A = g_jpeg_dev[dev_i];
    const int hb = A.turn ^= 1;   // this call's half of the pinned staging
    A.host = A.hosts[hb]; A.host_bytes = A.hosts_bytes[hb]; A.copied = A.copies[hb]; A.pending = A.pendings[hb];
    if (A.pending) {   // the copy two calls ago still reads this half
        SWAT_HIP(hipEventSynchronize(A.copied));
        A.pending = false;
    }
    const int grc = grow(&A.host, &A.host_bytes, up_bytes, true);
    A.hosts[hb] = A.host; A.hosts_bytes[hb] = A.host_bytes; A.pendings[hb] = false;   // (whatever grow left: never a stale pointer)
    SWAT_REQUIRE(grc == 0 && grow(&A.dev, &A.dev_bytes, up_bytes, false) == 0 &&
                     grow(&A.planes, &A.plane_bytes, plane_total + 64, false) == 0,
                 "swat_jpeg_decode_batch: cannot allocate %zu staging bytes", up_bytes);
    if (!A.copied) SWAT_HIP(hipEventCreateWithFlags(&A.copied, hipEventDisableTiming));
    if (!A.done) SWAT_HIP(hipEventCreateWithFlags(&A.done, hipEventDisableTiming));
    // the device buffers are re-used: a call on ANOTHER stream first waits for the previous call's kernels (same stream: ordered)
    if (A.any && A.last != s) SWAT_HIP(hipStreamWaitEvent(s, A.done, 0));
    memcpy(A.host, desc.data(), (size_t)n * sizeof(JpegDesc));
    int16_t* hcoef = (int16_t*)((char*)A.host + desc_bytes);
    // entropy decode: files are independent; threads take them from a shared counter (file sizes differ)
    int nt = threads > 0 ? threads : (n + 1) / 2;
    if (nt > 32) nt = 32;
    if (nt > n) nt = n;
    std::atomic<int> next(0);
    auto work = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1))
            if (status[i] == 0) entropy_decode(fr[(size_t)i], hcoef + desc[(size_t)i].coef_off[0]);
    };
    {
        const std::function<void()> job = work;
        swat::g_jpeg_pool.run(job, nt - 1);
    }
    SWAT_HIP(hipMemcpyAsync(A.dev, A.host, up_bytes, hipMemcpyHostToDevice, s));
    SWAT_HIP(hipEventRecord(A.copied, s));
    A.pending = true;
    A.hosts[hb] = A.host; A.hosts_bytes[hb] = A.host_bytes; A.copies[hb] = A.copied; A.pendings[hb] = true;
    const JpegDesc* ddesc = (const JpegDesc*)A.dev;
    const int16_t* dcoef = (const int16_t*)((char*)A.dev + desc_bytes);
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((max_blocks + 127) / 128), (unsigned)(3 * n)), dim3(128), 0, s, ddesc, dcoef,
                       (uint8_t*)A.planes);
    int64_t gx = (max_px + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, s, ddesc, (const uint8_t*)A.planes,
                       (uint8_t*)pixels_u8);
    const int rc = check_launch("swat_jpeg_decode_batch");
    if (rc == 0) {
        SWAT_HIP(hipEventRecord(A.done, s));
        A.last = s;
        A.any = true;
    }
    return rc;
}
