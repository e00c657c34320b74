// swat_amd/csrc/elementwise.hip — HBM-bound kernels of path A for gfx950: LayerNorm fwd/bwd, row softmax
// fwd/bwd (attention), CutMix paste + patchify, bias-gradient column sums, fused multi-group AdamW.
//
// ref: open_clip VisionTransformer pieces called through model.encode_image (utils/training.py:332,1474),
// CutMix paste utils/training.py:1458-1472, AdamW + LR groups utils/optimizers.py:22-35,131-146.
#include <stdlib.h>
#include "kernels.hpp"
#include "../../include/swat_hip.h"

namespace swat {

// --------------------------------------------------------------------------------------------
// LayerNorm forward: one wave per row; x fp32 [rows, d] -> y bf16 (and/or fp32), mean/rstd fp32.
// Two-pass in registers (row cached: d <= 64*4*MAXV floats).
// --------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 5;  // float4 per lane: d <= 1280

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int64_t x_row_stride,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     unsigned short* __restrict__ y_bf, float* __restrict__ y_f32,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int d,
                                                     float eps, int nsplit, int64_t split_stride, float* __restrict__ x_sum_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * x_row_stride);
    const int nv = d >> 2;
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nv ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (nsplit > 1 && c < nv) {  // split-K partial planes of the GEMM that produced x: sum in plane order, materialise
            for (int sp = 1; sp < nsplit; ++sp) {
                const float4 t = reinterpret_cast<const float4*>(x + sp * split_stride + (int64_t)row * x_row_stride)[c];
                v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
            }
            reinterpret_cast<float4*>(x_sum_out + (int64_t)row * x_row_stride)[c] = v[i];
        }
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, dd = v[i].w - mu;
            q += a * a + b * b + cc * cc + dd * dd;
        }
    }
    const float var = wave_sum(q) / (float)d;
    const float rs = rsqrtf(var + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 gm = reinterpret_cast<const float4*>(gamma)[c];
            const float4 bt = reinterpret_cast<const float4*>(beta)[c];
            const float o0 = (v[i].x - mu) * rs * gm.x + bt.x, o1 = (v[i].y - mu) * rs * gm.y + bt.y;
            const float o2 = (v[i].z - mu) * rs * gm.z + bt.z, o3 = (v[i].w - mu) * rs * gm.w + bt.w;
            if (y_bf)
                reinterpret_cast<ushort4*>(y_bf + (int64_t)row * d)[c] = make_ushort4(f2bf(o0), f2bf(o1), f2bf(o2), f2bf(o3));
            if (y_f32) reinterpret_cast<float4*>(y_f32 + (int64_t)row * d)[c] = make_float4(o0, o1, o2, o3);
        }
    }
}

// --------------------------------------------------------------------------------------------
// LayerNorm backward.  dy fp32 [rows,d] (row stride dy_stride), x fp32 (row stride x_stride).
//   dx_out[row] = (dx_add ? dx_add[row] : 0) + rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat))
// dx_out fp32 (+ optional bf16 copy).  Per-block partial dgamma/dbeta go to part[blockIdx][2][d];
// ln_bwd_reduce sums them in a fixed order (deterministic, no float atomics).
// Each block = 4 waves; each wave walks rows  wave_global, wave_global + nwaves, ...
// --------------------------------------------------------------------------------------------
constexpr int LNB_MAX_BLOCKS = 512;   // two workgroups per CU; partial sums: 512 x 3 x d floats (1024 workgroups gave the reduction behind it 12.6 MB to
                                      // read at d = 1024: 34 us per LayerNorm on 96 workgroups, 1.7 ms of a ViT-L/14 step)
// waves per workgroup: 8 up to d = 768 (M = 1600 rows -> 200 workgroups, every wave exactly one row, ~126 VGPRs: two
// workgroups per CU); 4 for wider rows (~150 VGPRs: three 4-wave workgroups per CU instead of one 8-wave workgroup)
static inline int lnb_waves(int d) { return d <= 768 ? 8 : 4; }
static inline int lnb_blocks(int rows, int d) {
    const int w = lnb_waves(d);
    const int n = (rows + w - 1) / w;
    return n > LNB_MAX_BLOCKS ? LNB_MAX_BLOCKS : n;
}
template <int NSPLIT, int VPL, int LNB_WAVES = (VPL <= 3 ? 8 : 4)>  // VPL float4 per lane: d <= 256 * VPL
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_kernel(const float* __restrict__ dy, int64_t dy_stride,
                                                     const float* __restrict__ x, int64_t x_stride,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* dx_add,
                                                     int64_t dxa_stride, float* dx_out, int64_t dxo_stride,
                                                     unsigned short* __restrict__ dx_bf, int64_t dxb_stride,
                                                     float* __restrict__ part, int rows, int d, int64_t split_stride,
                                                     int ln_blocks, int fill_period) {
    __shared__ float sh[LNB_WAVES][64 * 4 * VPL];  // [wave][d], reused for dgamma / dbeta / colsum(dx_bf16)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2;
    if ((int)blockIdx.x >= ln_blocks) {
        // fill role (swat_layernorm_bwd_partial_fill): the rows of dx_out / dx_bf BETWEEN the LayerNorm's rows — output row j of the
        // LayerNorm is row j * fill_period of the (contiguous, d-wide) output, rows with j % fill_period != 0 are set to zero.  The
        // class-token gradient of the tower's top: one launch instead of a zero fill and a LayerNorm backward.
        const int64_t total = (int64_t)rows * fill_period;
        const int fb = (int)blockIdx.x - ln_blocks, nfb = (int)gridDim.x - ln_blocks;
        for (int64_t r = (int64_t)fb * LNB_WAVES + wave; r < total; r += (int64_t)nfb * LNB_WAVES) {
            if (r % fill_period == 0) continue;
            for (int c = lane; c < nv; c += 64) {
                if (dx_out) reinterpret_cast<float4*>(dx_out + r * d)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (dx_bf) reinterpret_cast<ushort4*>(dx_bf + r * d)[c] = make_ushort4(0, 0, 0, 0);
            }
        }
        return;
    }
    float4 dg[VPL], db[VPL], dcs[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[i] = dg[i];
        dcs[i] = dg[i];
    }
    for (int row = blockIdx.x * LNB_WAVES + wave; row < rows; row += ln_blocks * LNB_WAVES) {
        const float4* dyr = reinterpret_cast<const float4*>(dy + (int64_t)row * dy_stride);
        const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * x_stride);
        const float mu = mean[row], rs = rstd[row];
        float4 dyv[VPL], xh[VPL], pl[NSPLIT > 1 ? NSPLIT - 1 : 1][VPL], add[VPL];
        float s1 = 0.f, s2 = 0.f;
        // every load of the row is issued before the first use: one memory latency per row, not one per plane
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                dyv[i] = dyr[c];
#pragma unroll
                for (int sp = 1; sp < NSPLIT; ++sp)
                    pl[sp - 1][i] = reinterpret_cast<const float4*>(dy + sp * split_stride + (int64_t)row * dy_stride)[c];
                {   // the saved activation is read here for the last time this step: keep it out of L2's way
                    typedef float nt4 __attribute__((ext_vector_type(4)));
                    const nt4 t = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(xr) + c);
                    xh[i] = make_float4(t[0], t[1], t[2], t[3]);
                }
                if (dx_add) add[i] = reinterpret_cast<const float4*>(dx_add + (int64_t)row * dxa_stride)[c];
            }
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
#pragma unroll
                for (int sp = 1; sp < NSPLIT; ++sp) {  // split-K partial planes of the producing dgrad GEMM, fixed order
                    const float4 t = pl[sp - 1][i];
                    dyv[i].x += t.x; dyv[i].y += t.y; dyv[i].z += t.z; dyv[i].w += t.w;
                }
                const float4 xv = xh[i];
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                const float4 gmv = reinterpret_cast<const float4*>(gamma)[c];  // (L1-resident; not kept across rows)
                const float a0 = dyv[i].x * gmv.x, a1 = dyv[i].y * gmv.y, a2 = dyv[i].z * gmv.z, a3 = dyv[i].w * gmv.w;
                s1 += a0 + a1 + a2 + a3;
                s2 += a0 * xh[i].x + a1 * xh[i].y + a2 * xh[i].z + a3 * xh[i].w;
                dg[i].x += dyv[i].x * xh[i].x; dg[i].y += dyv[i].y * xh[i].y;
                dg[i].z += dyv[i].z * xh[i].z; dg[i].w += dyv[i].w * xh[i].w;
                db[i].x += dyv[i].x; db[i].y += dyv[i].y; db[i].z += dyv[i].z; db[i].w += dyv[i].w;
                dyv[i] = make_float4(a0, a1, a2, a3);  // from here on only dy * gamma is needed
            }
        }
        const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 o;
                o.x = rs * (dyv[i].x - m1 - xh[i].x * m2);
                o.y = rs * (dyv[i].y - m1 - xh[i].y * m2);
                o.z = rs * (dyv[i].z - m1 - xh[i].z * m2);
                o.w = rs * (dyv[i].w - m1 - xh[i].w * m2);
                if (dx_add) {
                    const float4 a = add[i];
                    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
                }
                if (dx_out) reinterpret_cast<float4*>(dx_out + (int64_t)row * dxo_stride)[c] = o;
                if (dx_bf) {
                    const ushort4 ob = make_ushort4(f2bf(o.x), f2bf(o.y), f2bf(o.z), f2bf(o.w));
                    reinterpret_cast<ushort4*>(dx_bf + (int64_t)row * dxb_stride)[c] = ob;
                    // column sums of the ROUNDED output = bias gradient of the linear layer that produced this stream
                    dcs[i].x += bf2f(ob.x); dcs[i].y += bf2f(ob.y); dcs[i].z += bf2f(ob.z); dcs[i].w += bf2f(ob.w);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {  // cross-wave sums in wave order, one quantity at a time through the same 40 KB
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) reinterpret_cast<float4*>(sh[wave])[c] = q == 0 ? dg[i] : (q == 1 ? db[i] : dcs[i]);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < d; c += 64 * LNB_WAVES) {
            float v = sh[0][c];
#pragma unroll
            for (int w = 1; w < LNB_WAVES; ++w) v += sh[w][c];
            part[((int64_t)blockIdx.x * 3 + q) * d + c] = v;
        }
        __syncthreads();
    }
}
// grid (ceil(d / 32), 3): 32 columns x 32 strided groups of partial blocks per workgroup, groups summed in a fixed order
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int d,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ dcolsum) {
    __shared__ float sh[32][33];
    const int q = blockIdx.y;
    float* out = q == 0 ? dgamma : (q == 1 ? dbeta : dcolsum);
    if (!out) return;
    const int t = threadIdx.x & 31, j = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + t;
    float a = 0.f;
    if (c < d) {
        // 8 loads in flight per thread (the sum keeps its order: i = j, j + 32, ...): a bare loop pays one memory latency per term
        int i = j;
        for (; i + 7 * 32 < nblk; i += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[((int64_t)(i + 32 * u) * 3 + q) * d + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; i < nblk; i += 32) a += part[((int64_t)i * 3 + q) * d + c];
    }
    sh[j][t] = a;
    __syncthreads();
    if (j == 0 && c < d) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) v += sh[g][t];
        out[c] = v;
    }
}

// --------------------------------------------------------------------------------------------
// Attention softmax over rows of S (bf16 or f32 scores, ld = lds): P = softmax(scale * S[:, :T]) in bf16,
// columns T..lds-1 are written as ZERO (they pad the reduction of the P*V GEMM).  One wave per row.
// causal != 0 masks j > (row % T_q) (text tower).
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, unsigned short* __restrict__ P,
                                                          int64_t rows, int Tq, int T, int ld, float scale, int causal) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + row * ld;
    const int qi = (int)(row % Tq);
    const int lim = causal ? (qi + 1 < T ? qi + 1 : T) : T;
    float mx = -INFINITY;
    for (int j = lane; j < lim; j += 64) mx = fmaxf(mx, s[j] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < lim; j += 64) sum += __expf(s[j] * scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    unsigned short* p = P + row * ld;
    for (int j = lane; j < ld; j += 64) p[j] = j < lim ? f2bf(__expf(s[j] * scale - mx) * inv) : (unsigned short)0;
}

// dS = scale * P o (dP - rowsum(P o dP)); P bf16, dP f32, dS bf16; pad columns -> 0.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const unsigned short* __restrict__ P,
                                                          const float* __restrict__ dP, unsigned short* __restrict__ dS,
                                                          int64_t rows, int T, int ld, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const unsigned short* p = P + row * ld;
    const float* dp = dP + row * ld;
    float dot = 0.f;
    for (int j = lane; j < T; j += 64) dot += bf2f(p[j]) * dp[j];
    dot = wave_sum(dot);
    unsigned short* ds = dS + row * ld;
    for (int j = lane; j < ld; j += 64) ds[j] = j < T ? f2bf(scale * bf2f(p[j]) * (dp[j] - dot)) : (unsigned short)0;
}

// --------------------------------------------------------------------------------------------
// CutMix paste + patchify + cast (A5 + the im2col of A2a): images fp32 [B,3,H,W] -> patches bf16
// [B*gh*gw, ldp] with k = c*p*p + py*p + px (conv1.weight's own flattening); pixels inside the box
// [x1,x2) x [y1,y2) (dims 2,3 — ref utils/training.py:1466) come from images[perm[b]].  Reads a gathered
// copy, never permutes in place.  perm == nullptr or empty box -> plain patchify.
// One thread per 8 consecutive k (16-B store); columns K..ldp-1 are zero.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patchify_cutmix_kernel(const float* __restrict__ img, const int64_t* __restrict__ perm,
                                                              unsigned short* __restrict__ out, int Bn, int H, int W,
                                                              int ps, int ldp, int x1, int x2, int y1, int y2) {
    const int gw = W / ps, gh = H / ps;
    const int K = 3 * ps * ps;
    const int chunks = ldp / 8;
    const int64_t total = (int64_t)Bn * gh * gw * chunks;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ck = (int)(idx % chunks);
        const int64_t prow = idx / chunks;
        const int b = (int)(prow / (gh * gw));
        const int pr = (int)(prow % (gh * gw));
        const int py0 = (pr / gw) * ps, px0 = (pr % gw) * ps;
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ck * 8 + e;
            float f = 0.f;
            if (k < K) {
                const int c = k / (ps * ps), rem = k % (ps * ps);
                const int yy = py0 + rem / ps, xx = px0 + rem % ps;  // yy: dim 2, xx: dim 3
                int sb = b;
                if (perm && yy >= x1 && yy < x2 && xx >= y1 && xx < y2) sb = (int)perm[b];
                f = img[(((int64_t)sb * 3 + c) * H + yy) * W + xx];
            }
            v[e] = f2bf(f);
        }
        *reinterpret_cast<uint4*>(out + prow * ldp + ck * 8) = *reinterpret_cast<const uint4*>(v);
    }
}

// In-place CutMix paste on an fp32 image batch through a gathered copy of the box (for callers that
// want the mixed images themselves, e.g. parity tests of A5): dst may not alias src.
__global__ void cutmix_paste_kernel(const float* __restrict__ src, const int64_t* __restrict__ perm, float* __restrict__ dst,
                                    int Bn, int H, int W, int x1, int x2, int y1, int y2) {
    const int64_t total = (int64_t)Bn * 3 * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W), yy = (int)((i / W) % H);
        const int64_t bc = i / ((int64_t)H * W);
        const int b = (int)(bc / 3), c = (int)(bc % 3);
        int sb = b;
        if (yy >= x1 && yy < x2 && xx >= y1 && xx < y2) sb = (int)perm[b];
        dst[i] = src[(((int64_t)sb * 3 + c) * H + yy) * W + xx];
    }
}

// x[b, 0, :] = class_embedding + positional_embedding[0]   (A2b, cls row)
__global__ void cls_row_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos, int Bn,
                               int T, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bn * d) return;
    const int b = i / d, c = i % d;
    x[((int64_t)b * T) * d + c] = cls[c] + pos[c];
}

// --------------------------------------------------------------------------------------------
// Column sums (bias gradients): out[n] = sum_m X[m, n], X bf16 or f32 [rows, ld].  Block = 64 columns x 4
// row-lanes; rows are walked in a fixed order -> deterministic.
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int64_t rows, int n, int ld,
                                                     float* __restrict__ out, int64_t group_rows, int64_t group_stride_rows,
                                                     int64_t row_off) {
    // optional row grouping: logical row r -> physical (r / group_rows) * group_stride_rows + (r % group_rows) + row_off
    __shared__ float sh[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    float acc = 0.f;
    if (col < n) {
        for (int64_t r = rl; r < rows; r += 4) {
            const int64_t pr = group_rows > 0 ? (r / group_rows) * group_stride_rows + (r % group_rows) + row_off : r;
            float v;
            if constexpr (sizeof(T) == 2) v = bf2f(X[pr * ld + col]); else v = X[pr * ld + col];
            acc += v;
        }
    }
    sh[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && col < n) out[col] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// Two-stage bf16 column sum for the bias gradients of the big GEMMs (rows = B*T): stage 1 — grid (n/256, R):
// each thread owns 4 adjacent columns (8-B loads) and one of 4 row lanes inside its row chunk; stage 2 sums
// the R partials in chunk order.  Fixed order -> deterministic.
constexpr int COLSUM_R = 32;
__global__ __launch_bounds__(256) void colsum2_stage1_kernel(const unsigned short* __restrict__ X, int64_t rows, int n, int ld,
                                                             float* __restrict__ part) {
    __shared__ float4 sh[4][64];
    const int cg = blockIdx.x * 64 + (threadIdx.x & 63);  // 4-column group
    const int rl = threadIdx.x >> 6;
    const int64_t per = (rows + COLSUM_R - 1) / COLSUM_R;
    const int64_t r0 = (int64_t)blockIdx.y * per;
    const int64_t r1 = r0 + per < rows ? r0 + per : rows;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cg * 4 < n) {
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const ushort4 v = *reinterpret_cast<const ushort4*>(X + r * ld + cg * 4);
            acc.x += bf2f(v.x); acc.y += bf2f(v.y); acc.z += bf2f(v.z); acc.w += bf2f(v.w);
        }
    }
    sh[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && cg * 4 < n) {
        const int t = threadIdx.x;
        float4 o;
        o.x = ((sh[0][t].x + sh[1][t].x) + sh[2][t].x) + sh[3][t].x;
        o.y = ((sh[0][t].y + sh[1][t].y) + sh[2][t].y) + sh[3][t].y;
        o.z = ((sh[0][t].z + sh[1][t].z) + sh[2][t].z) + sh[3][t].z;
        o.w = ((sh[0][t].w + sh[1][t].w) + sh[2][t].w) + sh[3][t].w;
        *reinterpret_cast<float4*>(part + (int64_t)blockIdx.y * n + cg * 4) = o;
    }
}
__global__ void colsum2_stage2_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    float a = 0.f;
#pragma unroll 8
    for (int r = 0; r < COLSUM_R; ++r) a += part[(int64_t)r * n + c];
    out[c] = a;
}

// gather rows + cast fp32 -> bf16: out[r, :] = in[(r / g) * gs + (r % g) + off, :]
__global__ void gather_cast_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int64_t rows, int d,
                                   int64_t g, int64_t gs, int64_t off) {
    const int64_t total = rows * (d / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (d / 4);
        const int c = (int)(i % (d / 4));
        const int64_t pr = g > 0 ? (r / g) * gs + (r % g) + off : r;
        const float4 v = reinterpret_cast<const float4*>(in + pr * d)[c];
        reinterpret_cast<ushort4*>(out + r * d)[c] = make_ushort4(f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w));
    }
}

// --------------------------------------------------------------------------------------------
// Fused multi-group AdamW over ONE flat parameter buffer (A7).  torch.optim.AdamW semantics
// (ref: utils/optimizers.py:131-146; decoupled decay, bias correction, eps outside the sqrt-correction):
//   p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// Groups are contiguous segments [seg_begin[i], seg_begin[i+1]) with their own lr (device array so the
// step is graph-capturable).  Also refreshes the bf16 shadow copy the GEMMs read.
// --------------------------------------------------------------------------------------------
struct AdamSegs {
    int64_t begin[5];
    int n;
};
// U quads per thread and trip: U = 1 is the wide form (up to 4096 blocks: fastest alone); U = 4 with ONE block per CU (the "thin"
// form, swat_adamw_set_blocks) keeps the same bytes in flight per CU from 4 waves instead of 64, so the workgroups of a GEMM on
// another stream still find wave slots, registers and all of the LDS on every CU — a wide launch occupies every slot for its
// whole duration and the GEMM behind it simply waits (tools/step_timeline.py: c_fc dgrad 51 us beside it, 14.5 us alone).
template <int U>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, unsigned short* __restrict__ shadow, int64_t n,
                                                    AdamSegs segs, const float* __restrict__ lrs, float beta1, float beta2,
                                                    float eps, float wd, const float* __restrict__ bc /* [2]: 1-b1^t, 1-b2^t */,
                                                    float grad_scale, const unsigned* __restrict__ guard) {
    // a persistent forward of this step (or of an earlier one the host has not looked at yet) gave up on a hand-off: its gradients
    // are void — leave weights, moments and the bf16 shadow as they are; the host re-runs the step (swat_amd/step.py)
    if (guard && __builtin_nontemporal_load(guard) != 0u) return;
    const float bc1 = bc[0], sbc2 = sqrtf(bc[1]);
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    typedef float nt4 __attribute__((ext_vector_type(4)));
    // every byte is touched once per step: nontemporal accesses keep the stream out of the way of the GEMMs' L2 lines
    auto load = [&](nt4 (&pn)[U], nt4 (&gn)[U], nt4 (&mn)[U], nt4 (&vn)[U], int64_t i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n4) {
                pn[u] = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(p) + i);
                gn[u] = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(g) + i);
                mn[u] = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(m) + i);
                vn[u] = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(v) + i);
            }
        }
    };
    auto update = [&](const nt4 (&pn)[U], const nt4 (&gn)[U], const nt4 (&mn)[U], const nt4 (&vn)[U], int64_t i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n4) break;
            const int64_t e = i << 2;
            float pa[4] = {pn[u][0], pn[u][1], pn[u][2], pn[u][3]}, ma[4] = {mn[u][0], mn[u][1], mn[u][2], mn[u][3]};
            float va[4] = {vn[u][0], vn[u][1], vn[u][2], vn[u][3]};
            // the learning-rate group of an element: the groups are a handful of long contiguous segments, so all but a few quads of
            // the arena lie inside one — the group of the quad's first and last element decide (one lookup per quad; the per-element
            // lookup, a vector load and three 64-bit compares per element, only for a quad that straddles a boundary)
            int sg0 = 0, sg3 = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                sg0 += (q < segs.n && e >= segs.begin[q]) ? 1 : 0;
                sg3 += (q < segs.n && e + 3 >= segs.begin[q]) ? 1 : 0;
            }
            const float lr0 = lrs[sg0];
            float decay[4], stepsz[4];   // 1 - lr wd and lr / bc1 of the four elements: computed once for a quad inside one group
            {
#pragma clang fp contract(off)   // every AdamW form of the library rounds each operation: their results are identical by construction
                const float d0 = 1.0f - lr0 * wd, s0 = lr0 / bc1;
#pragma unroll
                for (int k = 0; k < 4; ++k) { decay[k] = d0; stepsz[k] = s0; }
                if (sg0 != sg3) {
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        int sg = 0;
#pragma unroll
                        for (int q = 1; q < 4; ++q) sg += (q < segs.n && e + k >= segs.begin[q]) ? 1 : 0;
                        const float lr = lrs[sg];
                        decay[k] = 1.0f - lr * wd;
                        stepsz[k] = lr / bc1;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma clang fp contract(off)
                const float gg = gn[u][k] * grad_scale;
                pa[k] *= decay[k];
                ma[k] = beta1 * ma[k] + (1.0f - beta1) * gg;
                va[k] = beta2 * va[k] + (1.0f - beta2) * gg * gg;
                const float denom = sqrtf(va[k]) / sbc2 + eps;
                pa[k] -= stepsz[k] * (ma[k] / denom);
            }
            __builtin_nontemporal_store((nt4){pa[0], pa[1], pa[2], pa[3]}, reinterpret_cast<nt4*>(p) + i);
            __builtin_nontemporal_store((nt4){ma[0], ma[1], ma[2], ma[3]}, reinterpret_cast<nt4*>(m) + i);
            __builtin_nontemporal_store((nt4){va[0], va[1], va[2], va[3]}, reinterpret_cast<nt4*>(v) + i);
            if (shadow) reinterpret_cast<ushort4*>(shadow)[i] = make_ushort4(f2bf(pa[0]), f2bf(pa[1]), f2bf(pa[2]), f2bf(pa[3]));
        }
    };
    // (tried for the thin form, U = 4: a trip's loads issued before the previous trip is updated and stored — two register sets, two
    //  trips of reads in flight: 594 us against 560 for the 88 M-parameter arena alone; the plain loop stays)
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += U * stride) {
        nt4 pn[U], gn[U], mn[U], vn[U];
        load(pn, gn, mn, vn, i0);
        update(pn, gn, mn, vn, i0);
    }
}

// the same update for up to 128 short ranges of the arena (one workgroup per range: biases and LayerNorm weights between the
// matrices whose AdamW runs in the weight-gradient epilogue, csrc/gemm_xl.hip); one learning-rate group
struct AdamRanges {
    int64_t begin[128];
    int len[128];
};
__global__ __launch_bounds__(256) void adamw_ranges_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, unsigned short* __restrict__ shadow, const AdamRanges r,
                                                           const float* __restrict__ lrs, float beta1, float beta2, float eps, float wd,
                                                           const float* __restrict__ bc, float grad_scale, const unsigned* __restrict__ guard) {
    if (guard && __builtin_nontemporal_load(guard) != 0u) return;
    const float bc1 = bc[0], sbc2 = sqrtf(bc[1]), lr = lrs[0];
    const int64_t b = r.begin[blockIdx.x];
    const int n4 = r.len[blockIdx.x] >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const int64_t e = b + 4 * (int64_t)i;
        const float4 pn = *reinterpret_cast<const float4*>(p + e), gn = *reinterpret_cast<const float4*>(g + e);
        const float4 mn = *reinterpret_cast<const float4*>(m + e), vn = *reinterpret_cast<const float4*>(v + e);
        float pa[4] = {pn.x, pn.y, pn.z, pn.w}, ga[4] = {gn.x, gn.y, gn.z, gn.w}, ma[4] = {mn.x, mn.y, mn.z, mn.w}, va[4] = {vn.x, vn.y, vn.z, vn.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma clang fp contract(off)
            const float gg = ga[k] * grad_scale;
            pa[k] *= 1.0f - lr * wd;
            ma[k] = beta1 * ma[k] + (1.0f - beta1) * gg;
            va[k] = beta2 * va[k] + (1.0f - beta2) * gg * gg;
            const float denom = sqrtf(va[k]) / sbc2 + eps;
            pa[k] -= (lr / bc1) * (ma[k] / denom);
        }
        *reinterpret_cast<float4*>(p + e) = make_float4(pa[0], pa[1], pa[2], pa[3]);
        *reinterpret_cast<float4*>(m + e) = make_float4(ma[0], ma[1], ma[2], ma[3]);
        *reinterpret_cast<float4*>(v + e) = make_float4(va[0], va[1], va[2], va[3]);
        if (shadow) *reinterpret_cast<ushort4*>(shadow + e) = make_ushort4(f2bf(pa[0]), f2bf(pa[1]), f2bf(pa[2]), f2bf(pa[3]));
    }
}

int launch_adamw_ranges(const int64_t* begin, const int* len, int n, const XlAdam& ad, hipStream_t s) {
    if (n < 1 || n > 128) return 1;
    AdamRanges r{};
    for (int i = 0; i < n; ++i) {
        if (begin[i] % 4 || len[i] % 4) return 1;
        r.begin[i] = begin[i];
        r.len[i] = len[i];
    }
    hipLaunchKernelGGL(adamw_ranges_kernel, dim3(n), dim3(256), 0, s, ad.p, ad.gbase, ad.m, ad.v, ad.shadow, r, ad.lr, ad.b1, ad.b2, ad.eps, ad.wd, ad.bc,
                       ad.gscale, ad.guard);
    return check_launch("adamw_ranges_kernel");
}

__global__ void cast_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        reinterpret_cast<ushort4*>(out)[i] = make_ushort4(f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w));
    }
}

__global__ void widen_bf16_kernel(const unsigned short* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const ushort4 v = reinterpret_cast<const ushort4*>(in)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(bf2f(v.x), bf2f(v.y), bf2f(v.z), bf2f(v.w));
    }
}

static inline int grid_for(int64_t work, int per_block) {
    int64_t b = (work + per_block - 1) / per_block;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace swat

using namespace swat;

SWAT_API int swat_layernorm_fwd(const float* x, int64_t x_row_stride, const float* gamma, const float* beta, void* y_bf16,
                                float* y_f32, float* mean, float* rstd, int rows, int d, float eps, swat_stream_t s) {
    SWAT_REQUIRE(d % 4 == 0 && d <= 256 * LN_MAXV, "swat_layernorm_fwd: d=%d unsupported", d);
    SWAT_REQUIRE(x_row_stride % 4 == 0, "swat_layernorm_fwd: row stride must be a multiple of 4");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, x, x_row_stride, gamma, beta,
                       (unsigned short*)y_bf16, y_f32, mean, rstd, rows, d, eps, 1, 0, nullptr);
    return check_launch("swat_layernorm_fwd");
}

SWAT_API int swat_layernorm_fwd_split(const float* x_planes, int64_t x_row_stride, int nsplit, int64_t split_stride,
                                      float* x_sum_out, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                                      float* mean, float* rstd, int rows, int d, float eps, swat_stream_t s) {
    SWAT_REQUIRE(d % 4 == 0 && d <= 256 * LN_MAXV && nsplit >= 1 && (nsplit == 1 || x_sum_out), "swat_layernorm_fwd_split: bad arguments");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, x_planes, x_row_stride, gamma, beta,
                       (unsigned short*)y_bf16, y_f32, mean, rstd, rows, d, eps, nsplit, split_stride, x_sum_out);
    return check_launch("swat_layernorm_fwd_split");
}

SWAT_API int64_t swat_layernorm_bwd_workspace_bytes(int rows, int d) {
    (void)rows;
    return (int64_t)LNB_MAX_BLOCKS * 3 * d * sizeof(float);
}

#define LNB_LAUNCH(NS, V)                                                                                                    \
    hipLaunchKernelGGL((ln_bwd_kernel<NS, V>), dim3(nblk + fill_blocks), dim3(64 * lnb_waves(d)), 0, (hipStream_t)s, dy, dy_stride, x, x_stride,  \
                       gamma, mean, rstd, dx_add, dxa_stride, dx_out, dxo_stride, (unsigned short*)dx_bf16, dxb_stride,      \
                       (float*)workspace, rows, d, split_stride, nblk, fill_period)
#define LNB_SPLITS(V)                       \
    switch (nsplit) {                       \
        case 1: LNB_LAUNCH(1, V); break;    \
        case 2: LNB_LAUNCH(2, V); break;    \
        case 3: LNB_LAUNCH(3, V); break;    \
        default: LNB_LAUNCH(4, V); break;   \
    }
#define LNB_DISPATCH()                      \
    do {                                    \
        if (d <= 768) { LNB_SPLITS(3) }     \
        else if (d <= 1024) { LNB_SPLITS(4) } \
        else { LNB_SPLITS(5) }              \
    } while (0)

SWAT_API int swat_layernorm_bwd(const float* dy, int64_t dy_stride, const float* x, int64_t x_stride, const float* gamma,
                                const float* mean, const float* rstd, const float* dx_add, int64_t dxa_stride,
                                float* dx_out, int64_t dxo_stride, void* dx_bf16, int64_t dxb_stride, float* dgamma,
                                float* dbeta, float* dcolsum, void* workspace, int rows, int d, swat_stream_t s) {
    SWAT_REQUIRE(d % 4 == 0 && d <= 256 * LN_MAXV, "swat_layernorm_bwd: d=%d unsupported", d);
    if (rows <= 0) return 0;
    const int nblk = lnb_blocks(rows, d);
    const int nsplit = 1;
    const int64_t split_stride = 0;
    const int fill_blocks = 0, fill_period = 1;
    LNB_DISPATCH();
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((d + 31) / 32, 3), dim3(1024), 0, (hipStream_t)s, (const float*)workspace,
                       nblk, d, dgamma, dbeta, dcolsum);
    return check_launch("swat_layernorm_bwd");
}

SWAT_API int swat_layernorm_bwd_split(const float* dy, int64_t dy_stride, int nsplit, int64_t split_stride, const float* x,
                                      int64_t x_stride, const float* gamma, const float* mean, const float* rstd,
                                      const float* dx_add, int64_t dxa_stride, float* dx_out, int64_t dxo_stride, void* dx_bf16,
                                      int64_t dxb_stride, float* dgamma, float* dbeta, float* dcolsum, void* workspace, int rows,
                                      int d, swat_stream_t s) {
    int rc = swat_layernorm_bwd_partial(dy, dy_stride, nsplit, split_stride, x, x_stride, gamma, mean, rstd, dx_add, dxa_stride,
                                        dx_out, dxo_stride, dx_bf16, dxb_stride, workspace, rows, d, s);
    if (rc) return rc;
    return swat_layernorm_bwd_finish(workspace, rows, d, dgamma, dbeta, dcolsum, s);
}

// The two halves of swat_layernorm_bwd_split as separate calls: the row pass (dx, its bf16 copy, per-workgroup partial
// sums into `workspace`) and the fixed-order reduction of those partials to dgamma / dbeta / dcolsum.  The second half
// feeds nothing but the optimizer, so a caller may run it on another stream once the first has been ordered before it.
SWAT_API int swat_layernorm_bwd_partial(const float* dy, int64_t dy_stride, int nsplit, int64_t split_stride, const float* x,
                                        int64_t x_stride, const float* gamma, const float* mean, const float* rstd,
                                        const float* dx_add, int64_t dxa_stride, float* dx_out, int64_t dxo_stride,
                                        void* dx_bf16, int64_t dxb_stride, void* workspace, int rows, int d, swat_stream_t s) {
    SWAT_REQUIRE(d % 4 == 0 && d <= 256 * LN_MAXV && nsplit >= 1 && nsplit <= 4, "swat_layernorm_bwd_partial: bad arguments");
    if (rows <= 0) return 0;
    const int nblk = lnb_blocks(rows, d);
    const int fill_blocks = 0, fill_period = 1;
    LNB_DISPATCH();
    return check_launch("swat_layernorm_bwd_partial");
}

// swat_layernorm_bwd_partial for LayerNorm rows that sit `period` rows apart in a contiguous d-wide output (dx_out and / or dx_bf16:
// [rows * period, d]; LayerNorm row j -> output row j * period): the same launch ALSO writes zeros to every output row in between.
// The tower's top gradient — only the class-token row of every image carries one (ref: `pooled = x[:, 0]` of open_clip's
// VisionTransformer.forward; utils/training.py:1486 backward) — in one launch instead of a zero fill and a LayerNorm backward.
SWAT_API int swat_layernorm_bwd_partial_fill(const float* dy, int64_t dy_stride, const float* x, int64_t x_stride, const float* gamma,
                                             const float* mean, const float* rstd, float* dx_out, void* dx_bf16, int period,
                                             void* workspace, int rows, int d, swat_stream_t s) {
    SWAT_REQUIRE(d % 4 == 0 && d <= 256 * LN_MAXV && period >= 1 && (dx_out || dx_bf16), "swat_layernorm_bwd_partial_fill: bad arguments");
    if (rows <= 0) return 0;
    const int nblk = lnb_blocks(rows, d);
    const int nsplit = 1;
    const int64_t split_stride = 0;
    const float* dx_add = nullptr;
    const int64_t dxa_stride = 0, dxo_stride = (int64_t)period * d, dxb_stride = (int64_t)period * d;
    const int fill_period = period;
    const int fill_blocks = period > 1 ? (512 - nblk > 1 ? 512 - nblk : 1) : 0;
    LNB_DISPATCH();
    return check_launch("swat_layernorm_bwd_partial_fill");
}

namespace swat {
// the same reduction for up to 64 independent sets of partial sums in one launch: grid (ceil(d / 32), 3, sets)
struct LnRedSets {
    const float* part[64];
    float* out[64][3];
};
__global__ __launch_bounds__(1024) void ln_bwd_reduce_sets_kernel(const LnRedSets a, int nblk, int d) {
    __shared__ float sh[32][33];
    const int q = blockIdx.y, z = blockIdx.z;
    float* out = a.out[z][q];
    if (!out) return;
    const float* part = a.part[z];
    const int t = threadIdx.x & 31, j = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + t;
    float acc = 0.f;
    if (c < d) {
        int i = j;
        for (; i + 7 * 32 < nblk; i += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[((int64_t)(i + 32 * u) * 3 + q) * d + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; i < nblk; i += 32) acc += part[((int64_t)i * 3 + q) * d + c];
    }
    sh[j][t] = acc;
    __syncthreads();
    if (j == 0 && c < d) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) v += sh[g][t];
        out[c] = v;
    }
}
// sets[i] = {part, dgamma, dbeta, dcolsum (may be null)}
int ln_bwd_finish_sets(const float* const* part, float* const* dgamma, float* const* dbeta, float* const* dcolsum, int nsets, int nblk, int d,
                       hipStream_t s) {
    for (int z0 = 0; z0 < nsets; z0 += 64) {
        LnRedSets a{};
        const int n = nsets - z0 < 64 ? nsets - z0 : 64;
        for (int i = 0; i < n; ++i) {
            a.part[i] = part[z0 + i];
            a.out[i][0] = dgamma[z0 + i]; a.out[i][1] = dbeta[z0 + i]; a.out[i][2] = dcolsum[z0 + i];
        }
        hipLaunchKernelGGL(ln_bwd_reduce_sets_kernel, dim3((d + 31) / 32, 3, n), dim3(1024), 0, s, a, nblk, d);
    }
    return check_launch("ln_bwd_finish_sets");
}
int ln_bwd_finish_blocks(const float* part, int nblk, int d, float* dgamma, float* dbeta, float* dcolsum, hipStream_t s) {
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((d + 31) / 32, 3), dim3(1024), 0, s, part, nblk, d, dgamma, dbeta, dcolsum);
    return check_launch("ln_bwd_finish_blocks");
}
}  // namespace swat

SWAT_API int swat_layernorm_bwd_finish(const void* workspace, int rows, int d, float* dgamma, float* dbeta, float* dcolsum,
                                       swat_stream_t s) {
    if (rows <= 0) return 0;
    const int nblk = lnb_blocks(rows, d);
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((d + 31) / 32, 3), dim3(1024), 0, (hipStream_t)s, (const float*)workspace,
                       nblk, d, dgamma, dbeta, dcolsum);
    return check_launch("swat_layernorm_bwd_finish");
}

SWAT_API int swat_softmax_fwd(const float* S, void* P_bf16, int64_t rows, int Tq, int T, int ld, float scale, int causal,
                              swat_stream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, S,
                       (unsigned short*)P_bf16, rows, Tq, T, ld, scale, causal);
    return check_launch("swat_softmax_fwd");
}

SWAT_API int swat_softmax_bwd(const void* P_bf16, const float* dP, void* dS_bf16, int64_t rows, int T, int ld, float scale,
                              swat_stream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)s,
                       (const unsigned short*)P_bf16, dP, (unsigned short*)dS_bf16, rows, T, ld, scale);
    return check_launch("swat_softmax_bwd");
}

SWAT_API int swat_patchify_cutmix(const float* images, const int64_t* perm, void* patches_bf16, int B, int H, int W,
                                  int patch, int ldp, int x1, int x2, int y1, int y2, swat_stream_t s) {
    SWAT_REQUIRE(H % patch == 0 && W % patch == 0, "swat_patchify_cutmix: image not divisible by patch");
    SWAT_REQUIRE(ldp % 8 == 0 && ldp >= 3 * patch * patch, "swat_patchify_cutmix: bad ldp");
    if (B <= 0) return 0;
    const int64_t total = (int64_t)B * (H / patch) * (W / patch) * (ldp / 8);
    hipLaunchKernelGGL(patchify_cutmix_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s, images, perm,
                       (unsigned short*)patches_bf16, B, H, W, patch, ldp, x1, x2, y1, y2);
    return check_launch("swat_patchify_cutmix");
}

SWAT_API int swat_cutmix_paste(const float* src, const int64_t* perm, float* dst, int B, int H, int W, int x1, int x2,
                               int y1, int y2, swat_stream_t s) {
    SWAT_REQUIRE(src != dst, "swat_cutmix_paste: dst must not alias src (the paste reads a gathered copy)");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(cutmix_paste_kernel, dim3(grid_for((int64_t)B * 3 * H * W, 256)), dim3(256), 0, (hipStream_t)s, src,
                       perm, dst, B, H, W, x1, x2, y1, y2);
    return check_launch("swat_cutmix_paste");
}

SWAT_API int swat_cls_row(float* x, const float* cls, const float* pos, int B, int T, int d, swat_stream_t s) {
    hipLaunchKernelGGL(cls_row_kernel, dim3((B * d + 255) / 256), dim3(256), 0, (hipStream_t)s, x, cls, pos, B, T, d);
    return check_launch("swat_cls_row");
}

SWAT_API int swat_colsum(const void* X, int dtype, int64_t rows, int n, int ld, float* out, int64_t group_rows,
                         int64_t group_stride_rows, int64_t row_off, swat_stream_t s) {
    if (n <= 0) return 0;
    if (dtype == SWAT_BF16)
        hipLaunchKernelGGL((colsum_kernel<unsigned short>), dim3((n + 63) / 64), dim3(256), 0, (hipStream_t)s,
                           (const unsigned short*)X, rows, n, ld, out, group_rows, group_stride_rows, row_off);
    else
        hipLaunchKernelGGL((colsum_kernel<float>), dim3((n + 63) / 64), dim3(256), 0, (hipStream_t)s, (const float*)X, rows,
                           n, ld, out, group_rows, group_stride_rows, row_off);
    return check_launch("swat_colsum");
}

SWAT_API int64_t swat_colsum_bf16_workspace_bytes(int n) { return (int64_t)COLSUM_R * n * sizeof(float); }

SWAT_API int swat_colsum_bf16(const void* X_bf16, int64_t rows, int n, int ld, float* out, void* workspace,
                              swat_stream_t s) {
    SWAT_REQUIRE(n % 4 == 0 && ld % 4 == 0, "swat_colsum_bf16: n and ld must be multiples of 4");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(colsum2_stage1_kernel, dim3((n / 4 + 63) / 64, COLSUM_R), dim3(256), 0, (hipStream_t)s,
                       (const unsigned short*)X_bf16, rows, n, ld, (float*)workspace);
    hipLaunchKernelGGL(colsum2_stage2_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, (const float*)workspace, n,
                       out);
    return check_launch("swat_colsum_bf16");
}

SWAT_API int swat_gather_cast_bf16(const float* in, void* out_bf16, int64_t rows, int d, int64_t group_rows,
                                   int64_t group_stride_rows, int64_t row_off, swat_stream_t s) {
    SWAT_REQUIRE(d % 4 == 0, "swat_gather_cast_bf16: d must be a multiple of 4");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(gather_cast_kernel, dim3(grid_for(rows * (d / 4), 256)), dim3(256), 0, (hipStream_t)s, in,
                       (unsigned short*)out_bf16, rows, d, group_rows, group_stride_rows, row_off);
    return check_launch("swat_gather_cast_bf16");
}

static int g_adamw_blocks = -1;  // > 0: the thin form with this many blocks (swat_adamw_set_blocks / SWAT_ADAMW_BLOCKS)
SWAT_API int swat_adamw_set_blocks(int blocks) {
    SWAT_REQUIRE(blocks >= 0 && blocks <= 4096, "swat_adamw_set_blocks: 0 (wide form) or 1..4096 blocks");
    g_adamw_blocks = blocks;
    return 0;
}

SWAT_API int swat_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, int64_t n,
                        const int64_t* seg_begin, int n_seg, const float* lrs_dev, float beta1, float beta2, float eps,
                        float weight_decay, const float* bias_corr_dev, float grad_scale, swat_stream_t s) {
    SWAT_REQUIRE(n % 4 == 0, "swat_adamw: flat size must be a multiple of 4");
    SWAT_REQUIRE(n_seg >= 1 && n_seg <= 4, "swat_adamw: 1..4 lr groups");
    AdamSegs segs{};
    segs.n = n_seg;
    for (int i = 0; i < n_seg; ++i) segs.begin[i] = seg_begin[i];
    if (g_adamw_blocks < 0) {
        const char* e = getenv("SWAT_ADAMW_BLOCKS");
        g_adamw_blocks = e ? atoi(e) : 0;
    }
    if (g_adamw_blocks > 0 && n / 4 >= (int64_t)g_adamw_blocks * 256 * 4)
        hipLaunchKernelGGL(adamw_kernel<4>, dim3(g_adamw_blocks), dim3(256), 0, (hipStream_t)s, params, grads, exp_avg, exp_avg_sq,
                           (unsigned short*)shadow_bf16, n, segs, lrs_dev, beta1, beta2, eps, weight_decay, bias_corr_dev, grad_scale,
                           mega_guard_word());
    else
        hipLaunchKernelGGL(adamw_kernel<1>, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)s, params, grads, exp_avg,
                           exp_avg_sq, (unsigned short*)shadow_bf16, n, segs, lrs_dev, beta1, beta2, eps, weight_decay,
                           bias_corr_dev, grad_scale, mega_guard_word());
    return check_launch("swat_adamw");
}

SWAT_API int swat_cast_bf16(const float* in, void* out_bf16, int64_t n, swat_stream_t s) {
    SWAT_REQUIRE(n % 4 == 0, "swat_cast_bf16: n must be a multiple of 4");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)s, in,
                       (unsigned short*)out_bf16, n);
    return check_launch("swat_cast_bf16");
}

SWAT_API int swat_widen_bf16(const void* in_bf16, float* out, int64_t n, swat_stream_t s) {
    SWAT_REQUIRE(n % 4 == 0, "swat_widen_bf16: n must be a multiple of 4");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(widen_bf16_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)s, (const unsigned short*)in_bf16,
                       out, n);
    return check_launch("swat_widen_bf16");
}
