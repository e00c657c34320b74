// swat_amd/csrc/preprocess.hip — decoded RGB bytes -> normalised fp32 [B, 3, n, n] batch.
//
// ref: utils/extras.py:66-85 (`transform`: RandomResizedCrop / Resize + CenterCrop, bicubic, flip, ToTensor, Normalize),
// applied per sample by utils/datasets/dataset_utils.py:165-180.  The arithmetic is Pillow's 8-bit resampler
// (oracle/image_oracle.py restates it): per axis a table of antialiased bicubic weights computed in double precision,
// normalised and rounded to 22-bit fixed point; a horizontal pass then a vertical pass, each int32-accumulated from
// 1 << 21 and clamped to a byte.  Three launches per batch, all byte / integer work bounded by HBM:
//   resample_table_kernel  — the weight tables of every image and axis, only for the n output positions that are kept
//                            (fp contraction off: the doubles must round exactly as the host library's do);
//   resample_rows_*_kernel — horizontal pass: a wave takes up to 4 source rows, their bytes go global -> LDS as aligned
//                            words by LDS-DMA (all rows in flight at once), lanes = output columns whose taps stay in
//                            registers across the rows; result bytes [rows of the region][n][3];
//   resample_cols_*_kernel — vertical pass + mirror + (v / 255 - mean) / std through a 3 x 256 table: one output row
//                            per wave, every tap's word of the intermediate loaded before the first is used, the
//                            row transposed through LDS and written as coalesced fp32 planes.  Neither pass has a
//                            block-wide barrier: every wave is its own latency chain.
// The *_fixed<KS> forms unroll over a compile-time tap count (taps beyond an image's own count carry weight 0); the
// generic forms loop over the run-time count and serve very large reductions (more than 25 taps: > 6x) and widths
// that are not a multiple of 4.  A variable-length tap loop costs one dependent memory latency per tap: 49 us + 28 us
// per 32-image batch in the generic form vs the fixed one.
// Algorithmic bytes per image: 3 * in_w * in_h read + 12 * n * n written (+ 3 * n * in_h * 2 for the intermediate).
#include "common.hpp"
#include "../../include/swat_hip.h"

namespace swat {

constexpr int PP_BITS = 22;  // Pillow: 32 - 8 - 2
constexpr int PP_DESC = 12;

__device__ __forceinline__ double bicubic_weight(double x) {
#pragma clang fp contract(off)
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PP_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// grid (image, axis); coef [image][axis][ks][n] (tap-major so that lanes = output positions read neighbours),
// bounds [image][axis][n][2] = first source index, tap count
__global__ __launch_bounds__(256) void resample_table_kernel(const int* __restrict__ desc, int n, int ks, int* __restrict__ coef,
                                                             int* __restrict__ bounds) {
#pragma clang fp contract(off)
    const int b = blockIdx.x, axis = blockIdx.y;
    const int* dsc = desc + b * PP_DESC;
    const int in_size = dsc[4 + axis], out_size = dsc[6 + axis], off = dsc[8 + axis];
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    int* ctab = coef + (int64_t)(b * 2 + axis) * ks * n;
    int* btab = bounds + (int64_t)(b * 2 + axis) * n * 2;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int xx = off + t;
        const double center = 0.0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        if (xmax > ks) xmax = ks;  // cannot happen when ks is the maximum the header defines
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) ww += bicubic_weight((x + xmin - center + 0.5) * ss);
        for (int x = 0; x < ks; ++x) {
            int v = 0;
            if (x < xmax) {
                double w = bicubic_weight((x + xmin - center + 0.5) * ss);
                if (ww != 0.0) w /= ww;
                v = w < 0 ? (int)(-0.5 + w * (double)(1 << PP_BITS)) : (int)(0.5 + w * (double)(1 << PP_BITS));
            }
            ctab[x * n + t] = v;
        }
        btab[2 * t] = xmin;
        btab[2 * t + 1] = xmax;
    }
}

// grid (ceil(max_in_h / 4), image); 4 waves, one source row each
__global__ __launch_bounds__(256) void resample_rows_generic_kernel(const unsigned char* __restrict__ pixels, const int64_t* __restrict__ offsets,
                                                            const int* __restrict__ desc, const int64_t* __restrict__ row_start,
                                                            const int* __restrict__ coef, const int* __restrict__ bounds,
                                                            unsigned char* __restrict__ tmp, int n, int ks, int row_words) {
    extern __shared__ unsigned int pp_lds_[];
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int* dsc = desc + b * PP_DESC;
    const int W = dsc[0], x0 = dsc[2], y0 = dsc[3], in_w = dsc[4], in_h = dsc[5];
    const int row = blockIdx.x * 4 + wave;
    unsigned int* mine = pp_lds_ + wave * row_words;
    int mis = 0;
    if (row < in_h) {
        const unsigned char* src = pixels + offsets[b] + ((int64_t)(y0 + row) * W + x0) * 3;
        mis = (int)((uintptr_t)src & 3);
        const unsigned int* src4 = reinterpret_cast<const unsigned int*>(src - mis);
        const int nd = (mis + in_w * 3 + 3) >> 2;
        for (int i = lane; i < nd; i += 64) mine[i] = __builtin_nontemporal_load(src4 + i);
    }
    __syncthreads();
    if (row >= in_h) return;
    const unsigned char* rowb = reinterpret_cast<const unsigned char*>(mine) + mis;
    unsigned char* dst = tmp + (row_start[b] + row) * (int64_t)(3 * n);
    const int* ctab = coef + (int64_t)(b * 2) * ks * n;
    const int* btab = bounds + (int64_t)(b * 2) * n * 2;
    for (int t = lane; t < n; t += 64) {
        const int xmin = btab[2 * t], cnt = btab[2 * t + 1];
        int s0 = 1 << (PP_BITS - 1), s1 = s0, s2 = s0;
        const unsigned char* p = rowb + xmin * 3;
        for (int x = 0; x < cnt; ++x) {
            const int k = ctab[x * n + t];
            s0 += p[3 * x] * k;
            s1 += p[3 * x + 1] * k;
            s2 += p[3 * x + 2] * k;
        }
        dst[3 * t] = (unsigned char)clip8(s0);
        dst[3 * t + 1] = (unsigned char)clip8(s1);
        dst[3 * t + 2] = (unsigned char)clip8(s2);
    }
}

// grid (n output rows, image); threads over the 3n (channel, column) pairs of the row
__global__ __launch_bounds__(256) void resample_cols_generic_kernel(const int* __restrict__ desc, const int64_t* __restrict__ row_start,
                                                            const int* __restrict__ coef, const int* __restrict__ bounds,
                                                            const unsigned char* __restrict__ tmp, const float* __restrict__ lut,
                                                            float* __restrict__ out, int n, int ks) {
    const int yy = blockIdx.x, b = blockIdx.y;
    const int flip = desc[b * PP_DESC + 10];
    const int* ctab = coef + (int64_t)(b * 2 + 1) * ks * n + yy;
    const int* btab = bounds + (int64_t)(b * 2 + 1) * n * 2 + 2 * yy;
    const int ymin = btab[0], cnt = btab[1];
    const unsigned char* src = tmp + (row_start[b] + ymin) * (int64_t)(3 * n);
    for (int e = threadIdx.x; e < 3 * n; e += 256) {
        const int c = e / n, xx = e - c * n;
        int s = 1 << (PP_BITS - 1);
        for (int y = 0; y < cnt; ++y) s += src[(int64_t)y * 3 * n + 3 * xx + c] * ctab[y * n];
        const int xo = flip ? n - 1 - xx : xx;
        __builtin_nontemporal_store(lut[c * 256 + clip8(s)], out + ((int64_t)(b * 3 + c) * n + yy) * n + xo);
    }
}


// ---- fixed tap count forms --------------------------------------------------------------------------------------
constexpr int PP_ROWS = 4;  // source rows per wave (at most)

// grid (ceil(max_in_h / (4 * R)), image); wave w of block x owns rows (4x + w) * R .. + R - 1.  row_words: LDS words per
// staged row = the row's aligned words rounded up to a multiple of 64 (one LDS-DMA instruction moves 64 words) plus slack
// for the zero-weight taps that reach past the row's end (24 words: 25 taps x 3 bytes, read as whole words).
template <int KS>
__global__ __launch_bounds__(256) void resample_rows_fixed_kernel(const unsigned char* __restrict__ pixels, const int64_t* __restrict__ offsets,
                                                                  const int* __restrict__ desc, const int64_t* __restrict__ row_start,
                                                                  const int* __restrict__ coef, const int* __restrict__ bounds,
                                                                  unsigned char* __restrict__ tmp, int n, int ks, int row_words, int R) {
    extern __shared__ unsigned int pp_lds_[];
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int* dsc = desc + b * PP_DESC;
    const int W = dsc[0], x0 = dsc[2], y0 = dsc[3], in_w = dsc[4], in_h = dsc[5];
    const int row0 = (blockIdx.x * 4 + wave) * R;
    unsigned int* mine = pp_lds_ + wave * R * row_words;
    int mis[PP_ROWS];
#pragma unroll
    for (int r = 0; r < PP_ROWS; ++r) {
        mis[r] = 0;
        if (r < R && row0 + r < in_h) {
            const unsigned char* src = pixels + offsets[b] + ((int64_t)(y0 + row0 + r) * W + x0) * 3;
            mis[r] = (int)((uintptr_t)src & 3);
            const unsigned int* src4 = reinterpret_cast<const unsigned int*>(src - mis[r]);
            const int nd = (mis[r] + in_w * 3 + 3) >> 2;
            for (int c0 = 0; c0 < nd; c0 += 64) {
                const int i = c0 + lane < nd ? c0 + lane : nd - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src4 + i),
                                                 (__attribute__((address_space(3))) void*)(mine + r * row_words + c0), 4, 0, 0);
            }
        }
    }
    if (row0 >= in_h) return;  // no block-wide barrier below: a wave reads only the LDS rows its own DMA wrote
    const int* ctab = coef + (int64_t)(b * 2) * ks * n;
    const int* btab = bounds + (int64_t)(b * 2) * n * 2;
    for (int t0 = 0; t0 < n; t0 += 256) {
        int kreg[4][KS], xm3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + lane + 64 * j, tt = t < n ? t : n - 1;
            xm3[j] = btab[2 * tt] * 3;
#pragma unroll
            for (int x = 0; x < KS; ++x) kreg[j][x] = x < ks ? ctab[x * n + tt] : 0;
        }
        // the taps' loads above travel together with the rows' DMA; everything has landed after this wait
        if (t0 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < PP_ROWS; ++r) {
            if (r < R && row0 + r < in_h) {
                const unsigned int* roww = mine + r * row_words;
                unsigned char* dst = tmp + (row_start[b] + row0 + r) * (int64_t)(3 * n);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = t0 + lane + 64 * j;
                    // the 3 * KS bytes under this column's taps, as whole LDS words shifted into place (a third of the
                    // LDS instructions of byte reads, and bank conflicts per word instead of per byte)
                    constexpr int NU = (3 * KS + 3) / 4;
                    const int off = xm3[j] + mis[r];
                    const unsigned int* wp = roww + (off >> 2);
                    unsigned int w[NU + 1], u[NU];
#pragma unroll
                    for (int i = 0; i <= NU; ++i) w[i] = wp[i];
#pragma unroll
                    for (int i = 0; i < NU; ++i) u[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], (unsigned)(off & 3));
                    int s0 = 1 << (PP_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll
                    for (int x = 0; x < KS; ++x) {
                        s0 += (int)((u[(3 * x) >> 2] >> (8 * ((3 * x) & 3))) & 255u) * kreg[j][x];
                        s1 += (int)((u[(3 * x + 1) >> 2] >> (8 * ((3 * x + 1) & 3))) & 255u) * kreg[j][x];
                        s2 += (int)((u[(3 * x + 2) >> 2] >> (8 * ((3 * x + 2) & 3))) & 255u) * kreg[j][x];
                    }
                    if (t < n) {
                        dst[3 * t] = (unsigned char)clip8(s0);
                        dst[3 * t + 1] = (unsigned char)clip8(s1);
                        dst[3 * t + 2] = (unsigned char)clip8(s2);
                    }
                }
            }
        }
    }
}

// grid (ceil(n / 4), image), one output row per wave (four independent latency chains per block); n % 4 == 0.
// Dynamic LDS: 4 x 3n bytes (each wave's output row as bytes, pixel-interleaved).
template <int KS>
__global__ __launch_bounds__(256) void resample_cols_fixed_kernel(const int* __restrict__ desc, const int64_t* __restrict__ row_start,
                                                                  const int* __restrict__ coef, const int* __restrict__ bounds,
                                                                  const unsigned char* __restrict__ tmp, const float* __restrict__ lut,
                                                                  float* __restrict__ out, int n, int ks) {
    extern __shared__ unsigned int pp_lds_[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // scalar: taps via s_load
    const int yy = blockIdx.x * 4 + wave, b = blockIdx.y;
    if (yy >= n) return;  // waves are independent: no block-wide barrier in this kernel
    const int flip = desc[b * PP_DESC + 10];
    const int* ctab = coef + (int64_t)(b * 2 + 1) * ks * n + yy;
    const int* btab = bounds + (int64_t)(b * 2 + 1) * n * 2 + 2 * yy;
    const int ymin = btab[0], cnt = btab[1];
    int k[KS];
#pragma unroll
    for (int y = 0; y < KS; ++y) k[y] = y < ks ? ctab[y * n] : 0;
    const int nd = (3 * n) >> 2;
    unsigned int* mine = pp_lds_ + wave * nd;
    const unsigned int* src4 = reinterpret_cast<const unsigned int*>(tmp + (row_start[b] + ymin) * (int64_t)(3 * n));
    for (int i = lane; i < nd; i += 64) {
        unsigned int v[KS];
#pragma unroll
        for (int y = 0; y < KS; ++y) v[y] = src4[(int64_t)(y < cnt ? y : cnt - 1) * nd + i];
        int s[4] = {1 << (PP_BITS - 1), 1 << (PP_BITS - 1), 1 << (PP_BITS - 1), 1 << (PP_BITS - 1)};
#pragma unroll
        for (int y = 0; y < KS; ++y)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += (int)((v[y] >> (8 * j)) & 255u) * k[y];
        int c0 = clip8(s[0]), c1 = clip8(s[1]), c2 = clip8(s[2]), c3 = clip8(s[3]);
        // opaque to the optimiser: hipcc 7.2 otherwise folds shift + clamp + pack of c0, c1 into v_ashr_pk_u8_i32 and ORs
        // c2, c3 onto its result as if the instruction cleared the upper half of the register — on gfx950 it keeps it
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
        mine[i] = (unsigned)c0 | ((unsigned)c1 << 8) | ((unsigned)c2 << 16) | ((unsigned)c3 << 24);
    }
    // one wave's LDS instructions execute in order: its reads below see its writes above once the compiler keeps them apart
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned char* rowb = reinterpret_cast<const unsigned char*>(mine);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float* plane = out + ((int64_t)(b * 3 + c) * n + yy) * n;
        for (int xx = lane; xx < n; xx += 64)
            __builtin_nontemporal_store(lut[c * 256 + rowb[3 * xx + c]], plane + (flip ? n - 1 - xx : xx));
    }
}

static int64_t pp_tables_bytes(int n_images, int n_px, int ks) {
    return (int64_t)n_images * 2 * n_px * (ks + 2) * (int64_t)sizeof(int);
}

}  // namespace swat

SWAT_API int64_t swat_preprocess_workspace_bytes(int n_images, int n_px, int max_ksize, int64_t total_rows) {
    if (n_images <= 0 || n_px <= 0 || max_ksize <= 0 || total_rows < 0) return 0;
    return swat::pp_tables_bytes(n_images, n_px, max_ksize) + total_rows * 3 * n_px + 16;
}

SWAT_API int swat_preprocess_batch(const void* pixels_u8, const int64_t* offsets, const int32_t* desc, const int64_t* row_start,
                                   const float* lut, float* out, int n_images, int n_px, int max_in_w, int max_in_h,
                                   int max_ksize, void* workspace, int64_t workspace_bytes, swat_stream_t stream) {
    SWAT_REQUIRE(pixels_u8 && offsets && desc && row_start && lut && out && workspace, "swat_preprocess_batch: null pointer");
    SWAT_REQUIRE(n_images > 0 && n_px > 0 && max_in_w > 0 && max_in_h > 0 && max_ksize >= 5 && (max_ksize & 1),
                 "swat_preprocess_batch: bad sizes (images %d, n_px %d, region %d x %d, ksize %d)", n_images, n_px, max_in_w,
                 max_in_h, max_ksize);
    const int64_t tb = swat::pp_tables_bytes(n_images, n_px, max_ksize);
    SWAT_REQUIRE(workspace_bytes >= tb + 16, "swat_preprocess_batch: workspace of %lld bytes is smaller than the tables (%lld)",
                 (long long)workspace_bytes, (long long)tb);
    int* coef = (int*)workspace;
    int* bounds = coef + (int64_t)n_images * 2 * n_px * max_ksize;
    unsigned char* tmp = (unsigned char*)workspace + tb;
    const unsigned char* pix = (const unsigned char*)pixels_u8;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(swat::resample_table_kernel, dim3(n_images, 2), dim3(256), 0, s, desc, n_px, max_ksize, coef, bounds);
    const bool fixed = max_ksize <= 25 && n_px % 4 == 0;
    if (fixed) {
        // words of the widest staged row, in whole LDS-DMA instructions, + the reach of 25 zero-weight taps
        const int row_words = ((max_in_w * 3 + 3 + 3) / 4 + 63) / 64 * 64 + 24;
        SWAT_REQUIRE(16 * row_words <= 64 * 1024, "swat_preprocess_batch: regions wider than 5300 pixels are not supported (%d)", max_in_w);
        const int R = 64 * row_words <= 48 * 1024 ? 4 : (32 * row_words <= 48 * 1024 ? 2 : 1);
        const dim3 grid_r((max_in_h + 4 * R - 1) / (4 * R), n_images), grid_c((n_px + 3) / 4, n_images);
        const size_t lds_r = (size_t)16 * R * row_words, lds_c = (size_t)4 * 3 * n_px;
#define SWAT_PP_LAUNCH(KS)                                                                                                             \
    do {                                                                                                                               \
        hipLaunchKernelGGL(swat::resample_rows_fixed_kernel<KS>, grid_r, dim3(256), lds_r, s, pix, offsets, desc, row_start, coef,     \
                           bounds, tmp, n_px, max_ksize, row_words, R);                                                                \
        hipLaunchKernelGGL(swat::resample_cols_fixed_kernel<KS>, grid_c, dim3(256), lds_c, s, desc, row_start, coef, bounds, tmp, lut, \
                           out, n_px, max_ksize);                                                                                      \
    } while (0)
        if (max_ksize <= 5) SWAT_PP_LAUNCH(5);
        else if (max_ksize <= 9) SWAT_PP_LAUNCH(9);
        else if (max_ksize <= 13) SWAT_PP_LAUNCH(13);
        else if (max_ksize <= 17) SWAT_PP_LAUNCH(17);
        else SWAT_PP_LAUNCH(25);
#undef SWAT_PP_LAUNCH
    } else {
        const int row_words = (max_in_w * 3 + 3 + 3) / 4 + 1;
        SWAT_REQUIRE(16 * row_words <= 64 * 1024, "swat_preprocess_batch: regions wider than 5300 pixels are not supported (%d)", max_in_w);
        hipLaunchKernelGGL(swat::resample_rows_generic_kernel, dim3((max_in_h + 3) / 4, n_images), dim3(256), 4 * row_words * sizeof(unsigned int),
                           s, pix, offsets, desc, row_start, coef, bounds, tmp, n_px, max_ksize, row_words);
        hipLaunchKernelGGL(swat::resample_cols_generic_kernel, dim3(n_px, n_images), dim3(256), 0, s, desc, row_start, coef, bounds, tmp, lut,
                           out, n_px, max_ksize);
    }
    return swat::check_launch("swat_preprocess_batch");
}
