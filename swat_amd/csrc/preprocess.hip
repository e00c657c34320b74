// swat_amd/csrc/preprocess.hip — decoded RGB bytes -> normalised fp32 [B, 3, n, n] batch.
//
// ref: utils/extras.py:66-85 (`transform`: RandomResizedCrop / Resize + CenterCrop, bicubic, flip, ToTensor, Normalize),
// applied per sample by utils/datasets/dataset_utils.py:165-180.  The arithmetic is Pillow's 8-bit resampler
// (oracle/image_oracle.py restates it): per axis a table of antialiased bicubic weights computed in double precision,
// normalised and rounded to 22-bit fixed point; a horizontal pass then a vertical pass, each int32-accumulated from
// 1 << 21 and clamped to a byte.  Three launches per batch, all byte / integer work bounded by HBM:
//   resample_table_kernel  — the weight tables of every image and axis, only for the n output positions that are kept
//                            (fp contraction off: the doubles must round exactly as the host library's do);
//   resample_rows_kernel   — horizontal pass: one wave per source row, the row's bytes staged once in LDS as aligned
//                            words, lanes = output columns; result bytes [rows of the region][n][3];
//   resample_cols_kernel   — vertical pass + mirror + (v / 255 - mean) / std through a 3 x 256 table, written as
//                            coalesced fp32 planes.
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
__global__ __launch_bounds__(256) void resample_rows_kernel(const unsigned char* __restrict__ pixels, const int64_t* __restrict__ offsets,
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
__global__ __launch_bounds__(256) void resample_cols_kernel(const int* __restrict__ desc, const int64_t* __restrict__ row_start,
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
    const int row_words = (max_in_w * 3 + 3 + 3) / 4 + 1;
    SWAT_REQUIRE(4 * row_words * (int)sizeof(unsigned int) <= 64 * 1024, "swat_preprocess_batch: regions wider than %d pixels are not supported",
                 (64 * 1024 / 16 - 2) * 4 / 3);
    const int64_t tb = swat::pp_tables_bytes(n_images, n_px, max_ksize);
    SWAT_REQUIRE(workspace_bytes >= tb + 16, "swat_preprocess_batch: workspace of %lld bytes is smaller than the tables (%lld)",
                 (long long)workspace_bytes, (long long)tb);
    int* coef = (int*)workspace;
    int* bounds = coef + (int64_t)n_images * 2 * n_px * max_ksize;
    unsigned char* tmp = (unsigned char*)workspace + tb;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(swat::resample_table_kernel, dim3(n_images, 2), dim3(256), 0, s, desc, n_px, max_ksize, coef, bounds);
    hipLaunchKernelGGL(swat::resample_rows_kernel, dim3((max_in_h + 3) / 4, n_images), dim3(256), 4 * row_words * sizeof(unsigned int), s,
                       (const unsigned char*)pixels_u8, offsets, desc, row_start, coef, bounds, tmp, n_px, max_ksize, row_words);
    hipLaunchKernelGGL(swat::resample_cols_kernel, dim3(n_px, n_images), dim3(256), 0, s, desc, row_start, coef, bounds, tmp, lut, out, n_px,
                       max_ksize);
    return swat::check_launch("swat_preprocess_batch");
}
