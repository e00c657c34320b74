// swat_amd/csrc/vit.hip — the open_clip VisionTransformer forward/backward as ONE host call each: the C++
// composite sequences the gfx950 kernels (gemm.hip, elementwise.hip, tail.hip) on a single stream with no
// allocation and no synchronisation, so the Python host pays one ctypes call per pass.
//
// ref: model.encode_image(images) — utils/training.py:332,1474, testing.py:430, utils/features.py:134;
// arithmetic per SURVEY Appendix A (conv1 patch embed, cls + positional embedding, ln_pre, L pre-LN blocks with
// packed-QKV MHA and a (Quick)GELU MLP, ln_post on the class token, bias-free proj).
//
// Numerics: fp32 master weights + bf16 shadow weights; GEMMs read bf16 operands and accumulate in fp32; the
// residual stream, LayerNorm statistics, softmax, proj and everything after it are fp32.
#include <stdlib.h>
#include <vector>
#include "kernels.hpp"
#include "../../include/swat_hip.h"

namespace swat {

__global__ void pad_cast_rows_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int rows, int k, int ld) {
    const int64_t total = (int64_t)rows * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / ld), c = (int)(i % ld);
        out[i] = c < k ? f2bf(in[(int64_t)r * k + c]) : (unsigned short)0;
    }
}

__global__ __launch_bounds__(256) void zero2_kernel(uint4* __restrict__ a, int64_t na, uint4* __restrict__ b, int64_t nb) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < na) a[i] = z;
        else b[i - na] = z;
    }
}

// The forward's tail in one launch (round 4: ln_fwd_kernel on the class-token rows, then a [B, d] x [d, E] product with 32 rows — two
// launches of 5 + 13 us on the chain between the persistent forward and the loss): workgroup (j, b) normalises the class-token row of
// image b (two-pass statistics, as ln_fwd_kernel) and computes 64 columns of feats[b] = ln_post(x[b, 0]) @ proj.  Thread (e4, part):
// columns 64 j + 4 e4 .. + 3, rows c = part, part + 16, ... of proj as ONE fp32 fma chain in ascending c; the 16 parts are added in
// part order.  Workgroup j = 0 also stores u = ln_post(x[b, 0]) and its statistics for the backward.
// ref: `pooled = ln_post(x)[:, 0]; pooled @ proj` of open_clip's VisionTransformer.forward behind model.encode_image (utils/training.py:1474)
__global__ __launch_bounds__(256) void cls_ln_proj_kernel(const float* __restrict__ x, int64_t x_stride, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ proj,
                                                          float* __restrict__ u_out, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                          float* __restrict__ feats, int d, int E, float eps) {
    extern __shared__ float clp_sh[];   // [d] the normalised row | [16][64] partial sums
    float* su = clp_sh;
    float* sp = clp_sh + d;
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, j = blockIdx.x;
    const float* xr = x + (int64_t)b * x_stride;
    float s = 0.f;
    for (int c = tid; c < d; c += 256) {
        const float v = xr[c];
        su[c] = v;
        s += v;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mu = ((red[0] + red[1]) + (red[2] + red[3])) / (float)d;
    float q = 0.f;
    for (int c = tid; c < d; c += 256) {
        const float t = su[c] - mu;
        q += t * t;
    }
    q = wave_sum(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float rs = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)d + eps);
    for (int c = tid; c < d; c += 256) {
        const float o = (su[c] - mu) * rs * gamma[c] + beta[c];
        su[c] = o;
        if (j == 0) u_out[(int64_t)b * d + c] = o;
    }
    if (j == 0 && tid == 0) {
        mean_out[b] = mu;
        rstd_out[b] = rs;
    }
    __syncthreads();
    const int e4 = tid & 15, part = tid >> 4;
    const float* pc = proj + 64 * j + 4 * e4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8   // (unroll 16: 28 us instead of 10.6)
    for (int c = part; c < d; c += 16) {
        const float4 w = *reinterpret_cast<const float4*>(pc + (int64_t)c * E);
        const float uv = su[c];
        acc.x = fmaf(uv, w.x, acc.x); acc.y = fmaf(uv, w.y, acc.y); acc.z = fmaf(uv, w.z, acc.z); acc.w = fmaf(uv, w.w, acc.w);
    }
    *reinterpret_cast<float4*>(sp + part * 64 + 4 * e4) = acc;
    __syncthreads();
    if (tid < 64) {
        float v = sp[tid];
#pragma unroll
        for (int p = 1; p < 16; ++p) v += sp[p * 64 + tid];
        feats[(int64_t)b * E + 64 * j + tid] = v;
    }
}

// d u = d feats @ proj^T for the B class rows (the first launch of the backward's chain; round 4: a k-ordered small product that stages
// 82 KB per workgroup and runs a 512-deep dependent chain per output, 12 us).  Workgroup (j, b): columns 64 j .. + 63 of row b, a WAVE
// per column: the lanes split the E features (V float4 each, all of a wave's 16 columns loaded up front), one fma chain per lane and a
// fixed xor tree across the lanes — deterministic.  E = 256 V.
template <int V>
__global__ __launch_bounds__(256) void du_rows_kernel(const float* __restrict__ df, const float* __restrict__ proj, float* __restrict__ du, int d, int E) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, c0 = blockIdx.x * 64 + wave * 16;
    float4 f[V];
#pragma unroll
    for (int v = 0; v < V; ++v) f[v] = reinterpret_cast<const float4*>(df + (int64_t)b * E)[lane + 64 * v];
    float4 w[16][V];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int v = 0; v < V; ++v) w[i][v] = reinterpret_cast<const float4*>(proj + (int64_t)(c0 + i < d ? c0 + i : d - 1) * E)[lane + 64 * v];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float a = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            a = fmaf(f[v].x, w[i][v].x, a); a = fmaf(f[v].y, w[i][v].y, a); a = fmaf(f[v].z, w[i][v].z, a); a = fmaf(f[v].w, w[i][v].w, a);
        }
        a = wave_sum(a);
        if (lane == 0 && c0 + i < d) du[(int64_t)b * d + c0 + i] = a;
    }
}

struct VitDims {
    int B, img, patch, d, L, H, mlp, E, act, save;
    int g, T, M, Kp, ldk, dh, ldp;
    int Mp;  // M rounded up to 64: row count of every buffer a weight gradient reduces over (pad rows stay zero)
};

static int parse_dims(const int32_t* v, VitDims& D) {
    D.B = v[0]; D.img = v[1]; D.patch = v[2]; D.d = v[3]; D.L = v[4]; D.H = v[5]; D.mlp = v[6]; D.E = v[7]; D.act = v[8];
    D.save = v[9];
    if (D.B <= 0 || D.patch <= 0 || D.img % D.patch || D.d % 64 || D.H <= 0 || D.d % D.H || D.L <= 0 || D.L > 64) return -1;
    D.g = D.img / D.patch; D.T = D.g * D.g + 1; D.M = D.B * D.T; D.Kp = 3 * D.patch * D.patch; D.ldk = (D.Kp + 7) / 8 * 8;
    D.dh = D.d / D.H; D.ldp = (D.T + 7) / 8 * 8; D.Mp = (D.M + 63) / 64 * 64;
    if (D.dh % 8) return -1;
    return 0;
}

// parameter offsets (elements) into the flat fp32 / bf16 buffers — order documented in include/swat_hip.h
struct VitOffs {
    int64_t conv1, cls, pos, lnpre_w, lnpre_b, lnpost_w, lnpost_b, proj;
    const int64_t* layer;  // 12 per layer: ln1_w ln1_b qkv_w qkv_b out_w out_b ln2_w ln2_b fc_w fc_b proj_w proj_b
};
static VitOffs parse_offs(const int64_t* o) {
    VitOffs r{o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o + 8};
    return r;
}
enum { LN1W = 0, LN1B, QKVW, QKVB, OUTW, OUTB, LN2W, LN2B, FCW, FCB, PRW, PRB };

// activation / scratch workspace carve-up (same function for sizing, forward and backward)
struct VitWs {
    unsigned short* patches;  // [B*g*g, ldk]
    unsigned short* convw;    // [d, ldk] padded conv weight (only when Kp % 8)
    float* xpre;              // [M, d] conv + cls + pos (ln_pre input)
    float *mean_pre, *rstd_pre, *mean_post, *rstd_post;
    float* u;                 // [B, d] ln_post(cls)
    // per layer (index l*save)
    float* xs;                // (L+1) x [M,d] residual stream at block boundaries (xs[0] = ln_pre out)
    float* x1;                // L x [M,d] mid-block residual
    unsigned short *h1, *h2, *a;  // L x [M,d]
    unsigned short* qkv;      // L x [M,3d]
    unsigned short* P;        // L x [B*H,T,ldp]
    unsigned short *f, *gact; // L x [M,mlp]
    float *mean1, *rstd1, *mean2, *rstd2;  // L x [M]
    float* lse;               // L x [B*H, T] attention row log-sum-exp (tiled attention, 64 < T <= 320)
    // scratch
    float* S;                 // [B*H,T,ldp] scores / dP
    float* xsplit;            // 4 x [M,d] forward split-K planes
    unsigned short* dS;       // [B*H,T,ldp]
    float* dx;                // [M,d] residual-stream gradient
    unsigned short* dxb;      // [M,d] its bf16 copy (block input side: feeds c_proj)
    unsigned short* dxb2;     // [M,d] bf16 copy of the mid-block gradient (feeds out_proj)
    float* dh;                // [M,d] LN-input gradient
    unsigned short* dfb;      // [M,mlp]
    unsigned short* dab;      // [M,d]
    unsigned short* dqkv;     // [M,3d]
    float* du;                // [B,d]
    void *lnws, *lnws1, *lnws2;  // LayerNorm-backward partial sums: stem / tail, ln_1, ln_2 (their reductions run on the lane)
    // block-by-block rings of the buffers the weight-gradient lane reads (slot 0 = the pointers above).  With all four weight
    // gradients of a block in ONE lane launch behind its attention backward (csrc/gemm_wgrad.hip) the lane reads a block's dxb /
    // dfb / dxb2 / dqkv / LayerNorm partials while the main chain is one or two blocks further down: block l uses slot l % 2 (its
    // output gradient, written for block l - 1 while block l + 1's lane work may still read its own: slot (l - 1) % 3 of three)
    unsigned short *dxb_r[3], *dxb2_r[2], *dfb_r[2], *dqkv_r[2];
    void *lnws1_r[2], *lnws2_r[2];
    // persistent backward (csrc/vit_mega.hip), small batches of the B/32 shape only: every buffer that crosses workgroups is per block
    unsigned short *mb_dxb, *mb_dfb, *mb_dxb2, *mb_dqkv;   // [L + 1 | L | L | L][Mp, d | mlp | d | 3d]
    float *mb_dh2, *mb_dh1;                                // [L][4][M, d]
    float *mb_lnp1, *mb_lnp2;                              // [L][panels * 16][3][d]
    void* csws;               // column-sum partials
    int64_t total;
};

static VitWs carve(const VitDims& D, void* base) {
    VitWs w{};
    char* p = (char*)base;
    int64_t off = 0;
    auto take = [&](int64_t bytes) -> void* {
        void* r = p ? (void*)(p + off) : nullptr;
        off += (bytes + 255) / 256 * 256;
        return r;
    };
    const int64_t M = D.M, Mp = D.Mp, d = D.d, Ls = D.save ? D.L : 1;
    const int64_t pbytes = (int64_t)D.B * D.H * D.T * D.ldp;
    w.patches = (unsigned short*)take((int64_t)D.B * D.g * D.g * D.ldk * 2);
    w.convw = (unsigned short*)take((int64_t)d * D.ldk * 2);
    w.xpre = (float*)take(M * d * 4);
    w.mean_pre = (float*)take(M * 4); w.rstd_pre = (float*)take(M * 4);
    w.mean_post = (float*)take(D.B * 4); w.rstd_post = (float*)take(D.B * 4);
    w.u = (float*)take((int64_t)D.B * d * 4);
    w.xs = (float*)take((D.save ? D.L + 1 : 2) * M * d * 4);
    w.x1 = (float*)take(Ls * M * d * 4);
    // operands of the weight-gradient GEMMs (reduction over rows) carry Mp rows per layer: the lean TN kernels take
    // whole 64-deep k-steps only, and rows M..Mp-1 are never written (the workspace is zero-initialised once)
    w.h1 = (unsigned short*)take(Ls * Mp * d * 2);
    w.h2 = (unsigned short*)take(Ls * Mp * d * 2);
    w.a = (unsigned short*)take(Ls * Mp * d * 2);
    w.qkv = (unsigned short*)take(Ls * M * 3 * d * 2);
    w.P = (unsigned short*)take(Ls * pbytes * 2);
    w.f = (unsigned short*)take(Ls * M * D.mlp * 2);
    w.gact = (unsigned short*)take(Ls * Mp * D.mlp * 2);
    w.mean1 = (float*)take(Ls * M * 4); w.rstd1 = (float*)take(Ls * M * 4);
    w.mean2 = (float*)take(Ls * M * 4); w.rstd2 = (float*)take(Ls * M * 4);
    w.lse = (float*)take(Ls * (int64_t)D.B * D.H * D.T * 4);
    w.S = (float*)take(pbytes * 4);
    w.xsplit = (float*)take(4 * M * d * 4);  // split-K partial planes of a forward N = d GEMM (summed by the next LayerNorm)
    if (D.save) {
        w.dS = (unsigned short*)take(pbytes * 2);
        w.dx = (float*)take(M * d * 4);
        w.dxb = (unsigned short*)take(Mp * d * 2);
        w.dxb2 = (unsigned short*)take(Mp * d * 2);
        w.dh = (float*)take(4 * M * d * 4);  // up to 4 split-K partial planes of the dgrad that feeds a LayerNorm backward
        w.dfb = (unsigned short*)take(Mp * D.mlp * 2);
        w.dab = (unsigned short*)take(M * d * 2);
        w.dqkv = (unsigned short*)take(Mp * 3 * d * 2);
        w.du = (float*)take((int64_t)D.B * d * 4);
        w.lnws = take(swat_layernorm_bwd_workspace_bytes((int)M, (int)d));
        w.lnws1 = take(swat_layernorm_bwd_workspace_bytes((int)M, (int)d));
        w.lnws2 = take(swat_layernorm_bwd_workspace_bytes((int)M, (int)d));
        w.csws = take(swat_colsum_bf16_workspace_bytes(D.mlp > 3 * D.d ? D.mlp : 3 * D.d));
        w.dxb_r[0] = w.dxb; w.dxb2_r[0] = w.dxb2; w.dfb_r[0] = w.dfb; w.dqkv_r[0] = w.dqkv; w.lnws1_r[0] = w.lnws1; w.lnws2_r[0] = w.lnws2;
        w.dxb_r[1] = (unsigned short*)take(Mp * d * 2);
        w.dxb_r[2] = (unsigned short*)take(Mp * d * 2);
        w.dxb2_r[1] = (unsigned short*)take(Mp * d * 2);
        w.dfb_r[1] = (unsigned short*)take(Mp * D.mlp * 2);
        w.dqkv_r[1] = (unsigned short*)take(Mp * 3 * d * 2);
        w.lnws1_r[1] = take(swat_layernorm_bwd_workspace_bytes((int)M, (int)d));
        w.lnws2_r[1] = take(swat_layernorm_bwd_workspace_bytes((int)M, (int)d));
        if (vit_mega_backward_fits(D.B, D.T, D.d, D.H, D.mlp)) {
            const int64_t L = D.L, lnp = vit_mega_backward_lnp_bytes(D.B, D.d);
            w.mb_dxb = (unsigned short*)take((L + 1) * Mp * d * 2);
            w.mb_dfb = (unsigned short*)take(L * Mp * D.mlp * 2);
            w.mb_dxb2 = (unsigned short*)take(L * Mp * d * 2);
            w.mb_dqkv = (unsigned short*)take(L * Mp * 3 * d * 2);
            w.mb_dh2 = (float*)take(L * 4 * M * d * 4);   // four reduction quarters each
            w.mb_dh1 = (float*)take(L * 4 * M * d * 4);
            w.mb_lnp1 = (float*)take(L * lnp);
            w.mb_lnp2 = (float*)take(L * lnp);
        }
    }
    w.total = off;
    return w;
}

static GemmArgs G(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int out_f32) {
    GemmArgs a{};
    a.A = (const unsigned short*)A; a.B = (const unsigned short*)B; a.C = C;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.inner = 1; a.out_f32 = out_f32; a.alpha = 1.0f;
    return a;
}

#define TRY(x)              \
    do {                    \
        int rc__ = (x);     \
        if (rc__) return rc__; \
    } while (0)

// multi-head attention forward on the packed qkv buffer of one layer
static int attn_forward(const VitDims& D, const unsigned short* qkv, float* S, unsigned short* P, unsigned short* out,
                        hipStream_t s, int causal = 0, float* lse = nullptr) {
    const int T = D.T, d = D.d, dh = D.dh, H = D.H, ldp = D.ldp;
    if (T <= 64 && dh == 64 && !causal) return swat_attn_fwd_small(qkv, out, D.B, T, d, H, s);  // one workgroup per head
    if (T <= 320 && dh == 64) return swat_attn_fwd_tiled(qkv, out, lse, D.B, T, d, H, causal, s);  // ... per 64-query tile
    GemmArgs a = G(qkv, qkv + d, S, T, T, dh, 3 * d, 3 * d, ldp, 1);
    a.inner = H; a.sA0 = (int64_t)T * 3 * d; a.sA1 = dh; a.sB0 = a.sA0; a.sB1 = dh; a.sC0 = (int64_t)H * T * ldp; a.sC1 = (int64_t)T * ldp;
    TRY(gemm_dispatch(L_NT, a, D.B * H, s));
    TRY(swat_softmax_fwd(S, P, (int64_t)D.B * H * T, T, T, ldp, 1.0f / sqrtf((float)dh), causal, s));
    GemmArgs b = G(P, qkv + 2 * d, out, T, dh, T, ldp, 3 * d, d, 0);
    b.inner = H; b.sA0 = (int64_t)H * T * ldp; b.sA1 = (int64_t)T * ldp; b.sB0 = (int64_t)T * 3 * d; b.sB1 = dh; b.sC0 = (int64_t)T * d; b.sC1 = dh;
    return gemm_dispatch(L_NN, b, D.B * H, s);
}

static int attn_backward(const VitDims& D, const unsigned short* qkv, const unsigned short* P, const unsigned short* dout,
                         float* dP, unsigned short* dS, unsigned short* dqkv, hipStream_t s, const unsigned short* out = nullptr,
                         const float* lse = nullptr) {
    const int T = D.T, d = D.d, dh = D.dh, H = D.H, ldp = D.ldp, BH = D.B * D.H;
    if (T <= 64 && dh == 64) return swat_attn_bwd_small(qkv, dout, dqkv, D.B, T, d, H, s);  // recomputes P from Q, K
    if (T <= 320 && dh == 64 && out && lse) return swat_attn_bwd_tiled(qkv, out, dout, lse, dqkv, D.B, T, d, H, dP, s);  // dP: [B*H, T, ldp] fp32 scratch holds the row deltas
    const int64_t sP0 = (int64_t)H * T * ldp, sP1 = (int64_t)T * ldp, sQ0 = (int64_t)T * 3 * d, sO0 = (int64_t)T * d;
    {   // dV = P^T dO
        GemmArgs a = G(P, dout, dqkv + 2 * d, T, dh, T, ldp, d, 3 * d, 0);
        a.inner = H; a.sA0 = sP0; a.sA1 = sP1; a.sB0 = sO0; a.sB1 = dh; a.sC0 = sQ0; a.sC1 = dh;
        TRY(gemm_dispatch(L_TN, a, BH, s));
    }
    {   // dP = dO V^T
        GemmArgs a = G(dout, qkv + 2 * d, dP, T, T, dh, d, 3 * d, ldp, 1);
        a.inner = H; a.sA0 = sO0; a.sA1 = dh; a.sB0 = sQ0; a.sB1 = dh; a.sC0 = sP0; a.sC1 = sP1;
        TRY(gemm_dispatch(L_NT, a, BH, s));
    }
    TRY(swat_softmax_bwd(P, dP, dS, (int64_t)BH * T, T, ldp, 1.0f / sqrtf((float)dh), s));
    {   // dQ = dS K
        GemmArgs a = G(dS, qkv + d, dqkv, T, dh, T, ldp, 3 * d, 3 * d, 0);
        a.inner = H; a.sA0 = sP0; a.sA1 = sP1; a.sB0 = sQ0; a.sB1 = dh; a.sC0 = sQ0; a.sC1 = dh;
        TRY(gemm_dispatch(L_NN, a, BH, s));
    }
    {   // dK = dS^T Q
        GemmArgs a = G(dS, qkv, dqkv + d, T, dh, T, ldp, 3 * d, 3 * d, 0);
        a.inner = H; a.sA0 = sP0; a.sA1 = sP1; a.sB0 = sQ0; a.sB1 = dh; a.sC0 = sQ0; a.sC1 = dh;
        TRY(gemm_dispatch(L_TN, a, BH, s));
    }
    return 0;
}


// ---- token embedding gather + positional add (text tower input; ref: open_clip CLIP.encode_text) ----
__global__ void embed_tokens_kernel(const int64_t* __restrict__ tok, const float* __restrict__ table, const float* __restrict__ pos,
                                    float* __restrict__ out, int64_t rows, int T, int d) {
    const int64_t total = rows * (d / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (d / 4);
        const int c = (int)(i % (d / 4));
        const float4 e = reinterpret_cast<const float4*>(table + tok[r] * d)[c];
        const float4 q = reinterpret_cast<const float4*>(pos + (r % T) * d)[c];
        reinterpret_cast<float4*>(out + r * d)[c] = make_float4(e.x + q.x, e.y + q.y, e.z + q.z, e.w + q.w);
    }
}

static int g_split_k = 0;  // split-K planes of the deep N = d products (0 / 1: off, the default: measured 3.59 vs 3.30 ms/step
                           // with 3 planes on the producer / consumer kernel's 128 x 128 tiles); swat_vit_set_split_k

// dX = dY W with the reduction split over `S` batched launches writing S partial planes (summed by the consumer): the
// N = d dgrads of the step are ~300 blocks of 64x64 — 1.2 per CU — and 36-48 k-steps deep; splitting K fills the chip.
static int dgrad_split(const VitDims& D, const unsigned short* dY, int Kred, const unsigned short* W, float* planes, int& S,
                       hipStream_t s) {
    const int M = D.M, d = D.d;
    S = 1;
    const int64_t base_blocks = (int64_t)((M + 63) / 64) * ((d + 63) / 64);
    if (base_blocks < 600) {
        // Measured on MI355X (ViT-B/32, B = 32, ms/step): split 4: 4.61, 3: 4.54, 2: 4.55, none: 4.43 — since the lean
        // register-ring kernel took over the deep-K dgrads, the extra fp32 planes cost more than the added workgroups
        // buy.  The plumbing (planes summed inside the LayerNorm backward) stays for shapes with far fewer tiles.
        const int kDgradSplitMax = g_split_k;
        for (int cand = kDgradSplitMax; cand >= 2; --cand)
            if (Kred % cand == 0 && (Kred / cand) % 64 == 0 && Kred / cand >= 512) { S = cand; break; }
    }
    GemmArgs a = G(dY, W, planes, M, d, Kred / S, Kred, d, d, 1);
    a.inner = S; a.sA1 = Kred / S; a.sB1 = (int64_t)(Kred / S) * d; a.sC1 = (int64_t)M * d;
    return gemm_dispatch(L_NN, a, S, s);
}

}  // namespace swat
using namespace swat;

// dW = dY^T X and db = column sums of dY in one launch when the reduction (rows) is a whole number of 64-deep k-steps
// (the lean TN kernels add one MFMA against a ones fragment per k-step in their n-tile-0 workgroups); otherwise the
// two-stage column-sum kernels run beside the GEMM.  workspace: swat_colsum_bf16_workspace_bytes(n_out).
SWAT_API int swat_linear_wgrad_bias(const void* dY_bf16, const void* X_bf16, float* dW, float* dbias, int rows, int n_out,
                                    int n_in, int ld_dy, int ld_x, void* workspace, swat_stream_t stream) {
    SWAT_REQUIRE(dY_bf16 && X_bf16 && dW && dbias, "swat_linear_wgrad_bias: null pointer");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs a = G(dY_bf16, X_bf16, dW, n_out, n_in, rows, ld_dy, ld_x, n_in, 1);
    const bool fused = rows % 64 == 0 && rows >= 384;
    if (fused) a.colsum = dbias;
    TRY(gemm_dispatch(L_TN, a, 1, s));
    if (!fused) {
        SWAT_REQUIRE(workspace, "swat_linear_wgrad_bias: workspace is null");
        TRY(swat_colsum_bf16(dY_bf16, rows, n_out, ld_dy, dbias, workspace, s));
    }
    return 0;
}

// The same for up to 4 linear layers in ONE persistent launch over the joint tile list (the weight gradients of a
// transformer block are independent of each other; each launch pays dispatch, set-up and first-byte latency).  dbias[i]
// may be null.  Needs rows[i] % 64 == 0 (pad rows zero) for the grouped form; anything else runs layer by layer.
SWAT_API int swat_linear_wgrad_bias_group(int n, const void* const* dY_bf16, const void* const* X_bf16, float* const* dW,
                                          float* const* dbias, const int* rows, const int* n_out, const int* n_in,
                                          const int* ld_dy, const int* ld_x, void* workspace, swat_stream_t stream) {
    SWAT_REQUIRE(n >= 1 && n <= 64 && dY_bf16 && X_bf16 && dW && dbias && rows && n_out && n_in && ld_dy && ld_x,
                 "swat_linear_wgrad_bias_group: 1-64 layers, non-null arrays");
    hipStream_t s = (hipStream_t)stream;
    bool fused = true;
    for (int i = 0; i < n; ++i) {
        SWAT_REQUIRE(dY_bf16[i] && X_bf16[i] && dW[i] && ld_dy[i] % 8 == 0 && ld_x[i] % 8 == 0 && n_in[i] % 4 == 0,
                     "swat_linear_wgrad_bias_group: layer %d: null pointer, ld not a multiple of 8 or n_in not a multiple of 4", i);
        fused = fused && rows[i] % 64 == 0 && rows[i] >= 384;
    }
    bool k64 = n <= 8;
    for (int i = 0; i < n; ++i) k64 = k64 && rows[i] % 64 == 0;
    if (k64) {  // one round of 256 x 128 tiles over the joint list when it fills the chip (csrc/gemm_wgrad.hip)
        GemmArgs a[8];
        for (int i = 0; i < n; ++i) {
            a[i] = G(dY_bf16[i], X_bf16[i], dW[i], n_out[i], n_in[i], rows[i], ld_dy[i], ld_x[i], n_in[i], 1);
            a[i].colsum = dbias[i];
        }
        const int rc = wgrad_big_try(a, n, s);
        if (rc <= 0) return rc;
    }
    if (!fused || n == 1) {
        for (int i = 0; i < n; ++i) {
            if (dbias[i]) TRY(swat_linear_wgrad_bias(dY_bf16[i], X_bf16[i], dW[i], dbias[i], rows[i], n_out[i], n_in[i], ld_dy[i], ld_x[i], workspace, stream));
            else TRY(gemm_dispatch(L_TN, G(dY_bf16[i], X_bf16[i], dW[i], n_out[i], n_in[i], rows[i], ld_dy[i], ld_x[i], n_in[i], 1), 1, s));
        }
        return 0;
    }
    GemmArgs a[64];
    for (int i = 0; i < n; ++i) {
        a[i] = G(dY_bf16[i], X_bf16[i], dW[i], n_out[i], n_in[i], rows[i], ld_dy[i], ld_x[i], n_in[i], 1);
        a[i].colsum = dbias[i];
    }
    return gemm_group_dispatch(L_TN, a, n, s);
}

SWAT_API int64_t swat_vit_workspace_bytes(const int32_t* dims) {
    VitDims D;
    if (parse_dims(dims, D)) return -1;
    return carve(D, nullptr).total;
}

static int join_pending_lane(hipStream_t s);   // the lane half of an owed join (defined with swat_vit_join_pending)

SWAT_API int swat_vit_forward(const int32_t* dims, const int64_t* offs, const float* params, const void* shadow_bf16,
                              const float* images, const int64_t* perm, int x1, int x2, int y1, int y2, void* workspace,
                              float* feats, swat_stream_t stream) {
    VitDims D;
    SWAT_REQUIRE(parse_dims(dims, D) == 0, "swat_vit_forward: bad dims");
    SWAT_REQUIRE(params && shadow_bf16 && images && workspace && feats, "swat_vit_forward: null pointer");
    const VitOffs O = parse_offs(offs);
    VitWs w = carve(D, workspace);
    hipStream_t s = (hipStream_t)stream;
    const unsigned short* sh = (const unsigned short*)shadow_bf16;
    const int M = D.M, d = D.d, T = D.T, np = D.B * D.g * D.g;
    const float eps = 1e-5f;

    // A5 + A2a: CutMix paste fused into patch extraction, then the patch-embed GEMM with cls/pos epilogue
    TRY(swat_patchify_cutmix(images, perm, w.patches, D.B, D.img, D.img, D.patch, D.ldk, x1, x2, y1, y2, s));
    const unsigned short* convw = sh + O.conv1;
    if (D.Kp % 8) {
        // reads the fp32 MASTER conv1 weights (patch 14: Kp = 588): an owed join's lane half comes first whatever stepped stage 0
        TRY(join_pending_lane(s));
        hipLaunchKernelGGL(pad_cast_rows_kernel, dim3(256), dim3(256), 0, s, params + O.conv1, w.convw, d, D.Kp, D.ldk);
        convw = w.convw;
    }
    {
        // A join owed to the previous step (swat_vit_backward_join_later): the patch-embedding product waits for the LANE's tail — beside
        // a block's weight-gradient launch both take twice their time (round-5 timeline: 106 us instead of 21 for this product, 96
        // instead of 47 for the weight gradients it met, and the lane's chain is what bounds the optimizer phase) — and runs beside
        // the last blocks' AdamW instead, which is HBM-bound.  The second stream's tail is joined behind the stem as before.
        // Measured (alternating builds): 2.715 / 2.730 against 2.728 / 2.740 ms per step; in the trace the patch extraction in front
        // of this wait starts behind the lane's tail as well (the runtime submits the two together).
        static const int stem_late = getenv("SWAT_STEM_AFTER_LANE") ? atoi(getenv("SWAT_STEM_AFTER_LANE")) : 1;
        // (SWAT_STEM_AFTER_LANE=0 is a measurement switch for the FUSED step only, where stage 0 is stepped on the main stream;
        // swat_amd/step.py owes the join in no other case)
        if (stem_late) TRY(join_pending_lane(s));
        GemmArgs a = G(w.patches, convw, w.xpre, np, d, D.Kp, D.ldk, D.ldk, d, 1);
        a.rpg = D.g * D.g; a.gstride = T; a.roff = 1; a.rowbias = params + O.pos;
        TRY(gemm_dispatch(L_NT, a, 1, s));
    }
    TRY(swat_cls_row(w.xpre, params + O.cls, params + O.pos, D.B, T, d, s));
    // A2b: ln_pre -> residual stream xs[0]
    TRY(swat_layernorm_fwd(w.xpre, d, params + O.lnpre_w, params + O.lnpre_b, nullptr, w.xs, w.mean_pre, w.rstd_pre, M, d, eps, s));

    // (the previous step's weight gradients / AdamW of the BLOCKS may still run on the lane and the second stream: the stem above read
    // stage-0 parameters only; everything below reads the blocks')
    TRY(swat_vit_join_pending(stream));
    int pending = 1;  // split-K planes of the previous block's c_proj still to be summed into xs[l]
    // ViT-B/32 regime: all blocks in ONE persistent launch (csrc/vit_mega.hip); same buffers, so the backward is unchanged
    const int mega = vit_mega_forward(D.B, T, d, D.H, D.mlp, D.L, D.act, D.save, M, D.Mp, params, sh, O.layer, w.xs, w.x1, w.h1, w.h2, w.a,
                                      w.qkv, w.f, w.gact, w.mean1, w.rstd1, w.mean2, w.rstd2, s);
    if (mega < 0) return mega;
    for (int l = 0; l < (mega == 0 ? 0 : D.L); ++l) {
        const int64_t* lo = O.layer + 12 * l;
        const int64_t li = D.save ? l : 0;
        float* xin = w.xs + (D.save ? (int64_t)l : (int64_t)(l & 1)) * M * d;
        float* xout = w.xs + (D.save ? (int64_t)(l + 1) : (int64_t)((l + 1) & 1)) * M * d;
        float* x1 = w.x1 + li * M * d;
        unsigned short* h1 = w.h1 + li * (int64_t)D.Mp * d;
        unsigned short* h2 = w.h2 + li * (int64_t)D.Mp * d;
        unsigned short* at = w.a + li * (int64_t)D.Mp * d;
        unsigned short* qkv = w.qkv + li * M * 3 * d;
        unsigned short* P = w.P + li * (int64_t)D.B * D.H * T * D.ldp;
        unsigned short* f = w.f + li * (int64_t)M * D.mlp;
        unsigned short* ga = w.gact + li * (int64_t)D.Mp * D.mlp;
        // A2c: ln_1 -> packed QKV projection
        TRY(swat_layernorm_fwd_split(pending > 1 ? w.xsplit : xin, d, pending, (int64_t)M * d, xin, params + lo[LN1W], params + lo[LN1B], h1,
                                     nullptr, w.mean1 + li * M, w.rstd1 + li * M, M, d, eps, s));
        {
            GemmArgs a = G(h1, sh + lo[QKVW], qkv, M, 3 * d, d, d, d, 3 * d, 0);
            a.bias = params + lo[QKVB];
            TRY(gemm_dispatch(L_NT, a, 1, s));
        }
        // A2d: attention
        TRY(attn_forward(D, qkv, w.S, P, at, s, 0, w.lse + li * (int64_t)D.B * D.H * T));
        // A2e: out_proj + residual
        {
            GemmArgs a = G(at, sh + lo[OUTW], x1, M, d, d, d, d, d, 1);
            a.bias = params + lo[OUTB]; a.residual = xin;
            TRY(gemm_dispatch(L_NT, a, 1, s));
        }
        // A2f: ln_2 -> c_fc (+act) -> c_proj + residual
        TRY(swat_layernorm_fwd(x1, d, params + lo[LN2W], params + lo[LN2B], h2, nullptr, w.mean2 + li * M, w.rstd2 + li * M, M, d, eps, s));
        {
            GemmArgs a = G(h2, sh + lo[FCW], f, M, D.mlp, d, d, d, D.mlp, 0);
            a.bias = params + lo[FCB]; a.act = D.act; a.C2 = ga;
            a.c2_only = !D.save;  // inference: the pre-activation is never re-read
            TRY(gemm_dispatch(L_NT, a, 1, s));
        }
        {
            // c_proj: K = mlp is 48 k-steps on only ~300 tiles: split the reduction, the next LayerNorm sums the planes
            pending = 1;
            const int kFwdSplit = g_split_k;  // 3 planes on the producer / consumer kernel's 128 x 128 tiles (swat_vit_set_split_k)
            if (kFwdSplit && (int64_t)((M + 63) / 64) * ((d + 63) / 64) < 600)
                for (int cand = kFwdSplit; cand >= 2; --cand)
                    if (D.mlp % cand == 0 && (D.mlp / cand) % 64 == 0 && D.mlp / cand >= 512) { pending = cand; break; }
            GemmArgs a = G(ga, sh + lo[PRW], pending > 1 ? w.xsplit : xout, M, d, D.mlp / pending, D.mlp, D.mlp, d, 1);
            a.bias = params + lo[PRB]; a.residual = x1; a.epi0 = 1;
            a.inner = pending; a.sA1 = D.mlp / pending; a.sB1 = D.mlp / pending; a.sC1 = (int64_t)M * d;
            TRY(gemm_dispatch(L_NT, a, pending, s));
        }
    }
    // A2g: ln_post on the class token only, then the bias-free projection (fp32)
    float* xl = w.xs + (D.save ? (int64_t)D.L : (int64_t)(D.L & 1)) * M * d;
    if (pending == 1 && D.E % 64 == 0 && d % 16 == 0) {
        hipLaunchKernelGGL(cls_ln_proj_kernel, dim3(D.E / 64, D.B), dim3(256), (size_t)(d + 16 * 64) * sizeof(float), s, xl, (int64_t)T * d,
                           params + O.lnpost_w, params + O.lnpost_b, params + O.proj, w.u, w.mean_post, w.rstd_post, feats, d, D.E, eps);
        return check_launch("cls_ln_proj_kernel");
    }
    TRY(swat_layernorm_fwd_split(pending > 1 ? w.xsplit : xl, (int64_t)T * d, pending, (int64_t)M * d, xl, params + O.lnpost_w, params + O.lnpost_b,
                                 nullptr, w.u, w.mean_post, w.rstd_post, D.B, d, eps, s));
    TRY(swat_sgemm_small(w.u, params + O.proj, feats, D.B, D.E, d, d, 1, D.E, 1, D.E, 1.0f, 0, s));
    return 0;
}

// Weight-gradient GEMMs and bias column sums feed nothing but the optimizer, while the data-gradient chain is the
// critical path and each of its GEMMs alone leaves CUs idle (300-1200 tiles on 256 CUs).  The backward therefore
// forks the weight-gradient work onto a second, lower-priority stream: events order it after the producer of its
// operand and order the next overwrite of that operand buffer after it.  Mode 1 joins the lane before the call
// returns; mode 2 leaves the join to the caller (swat_vit_backward_join), so a stage-by-stage caller can hang its
// own per-stage work (gradient all-reduce, optimizer slice) behind the lane without stalling the main chain.
struct SideLane {
    hipStream_t st = nullptr;
    std::vector<hipEvent_t> ev;  // ring: an event is re-recorded only ~a whole backward later
    size_t used = 0;
    hipEvent_t doneA = nullptr, doneB = nullptr;  // last lane read of {dxb, dfb} / {dxb2, dqkv}
    hipEvent_t blk[2] = {nullptr, nullptr};       // one-launch form: end of the lane work of the last block with this parity
    bool open = false;           // lane work issued and not joined yet
    hipEvent_t aux_done = nullptr;  // end of what the backward queued on aux[0] (the piped AdamW of the blocks): joined with the lane
    // swat_vit_backward_join_later: the join is owed — whoever next reads what the lane / aux[0] wrote (the next forward's BLOCKS; any
    // library entry that takes the arena) waits for these two events on its own stream first (swat_vit_join_pending)
    bool pending = false;
    hipEvent_t pend_lane = nullptr, pend_aux = nullptr;
    hipEvent_t next() {
        if (ev.size() < 1024) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            ev.push_back(e);
            return e;
        }
        return ev[used++ % ev.size()];
    }
    hipStream_t aux[2] = {nullptr, nullptr};  // the data-parallel step's all-reduce / optimizer streams
    int ensure() {
        if (st) return 0;
        // all side streams of the process are created here, once and back to back: streams share a small number of
        // hardware queues, and a later-created stream that lands on the queue of the main stream or of the lane was
        // measured to triple the step time (tools/micro/sec_order32.py)
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return 1;
        if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least) != hipSuccess) return 1;
        for (int i = 0; i < 2; ++i)
            if (hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking) != hipSuccess) return 1;
        return 0;
    }
};
static SideLane g_side;
// d proj = u^T d feats of the stage L+1 backward, left pending for the call that runs the blocks (see there)
struct PendingProduct {
    bool on = false;
    const float *a = nullptr, *b = nullptr;
    float* c = nullptr;
    int d = 0, E = 0, B = 0;
};
static PendingProduct g_pend_dproj;
static bool g_defer_top = false;   // swat_vit_backward_defer_top: the caller's promise that the blocks' call comes next (one call)
// ... and the reduction of the ln_post backward's partial sums of the same call (w.lnws is not written again before the stem's
// LayerNorm backward, which runs after the blocks)
struct PendingLnPost {
    bool on = false;
    const void* ws = nullptr;
    int rows = 0, d = 0;
    float *dgamma = nullptr, *dbeta = nullptr, *dcolsum = nullptr;
};
static PendingLnPost g_pend_lnpost;
// (lane: the weight-gradient stream of a call that runs the blocks as the persistent launch.  The column sums — the last block's
// d c_proj.bias — belong to that block's arena range, whose AdamW follows the LANE's events, not the main stream: they are reduced
// there; d ln_post.weight / bias and d proj belong to stage L+1, whose AdamW the caller queues on `s` behind this)
static int flush_pending_dproj(hipStream_t s, hipStream_t lane = nullptr) {
    if (g_pend_lnpost.on) {
        const PendingLnPost q = g_pend_lnpost;
        g_pend_lnpost.on = false;
        if (lane && lane != s) {
            TRY(swat_layernorm_bwd_finish(q.ws, q.rows, q.d, nullptr, nullptr, q.dcolsum, lane));
            TRY(swat_layernorm_bwd_finish(q.ws, q.rows, q.d, q.dgamma, q.dbeta, nullptr, s));
        } else {
            TRY(swat_layernorm_bwd_finish(q.ws, q.rows, q.d, q.dgamma, q.dbeta, q.dcolsum, s));
        }
    }
    if (!g_pend_dproj.on) return 0;
    const PendingProduct p = g_pend_dproj;
    g_pend_dproj.on = false;
    return swat_sgemm_small(p.a, p.b, p.c, p.d, p.E, p.B, 1, p.d, p.E, 1, p.E, 1.0f, 0, s);
}

static int g_overlap = 1;
static int g_group_wgrad = 0;  // 1: a block's weight-gradient pairs as one persistent grouped launch each (measured equal in
                               // GEMM time, 3.38 vs 3.30 ms/step: the pair's main loops dominate, not its two launches)

SWAT_API int swat_vit_set_overlap(int mode) {
    SWAT_REQUIRE(mode >= 0 && mode <= 2, "swat_vit_set_overlap: mode must be 0, 1 or 2");
    SWAT_REQUIRE(!g_side.open, "swat_vit_set_overlap: join the open backward first");
    g_overlap = mode;
    return 0;
}

SWAT_API int swat_vit_set_split_k(int planes) {
    SWAT_REQUIRE(planes >= 0 && planes <= 4, "swat_vit_set_split_k: 0-4 planes");
    swat::g_split_k = planes;
    return 0;
}

SWAT_API swat_stream_t swat_vit_lane_stream(void) {
    if (g_side.ensure()) return nullptr;
    return (swat_stream_t)g_side.st;
}

SWAT_API swat_stream_t swat_vit_aux_stream(int i) {
    if (i < 0 || i > 1 || g_side.ensure()) return nullptr;
    return (swat_stream_t)g_side.aux[i];
}

SWAT_API int swat_vit_lane_order(swat_stream_t stream) {
    SWAT_REQUIRE(g_side.ensure() == 0, "swat_vit_lane_order: cannot create the weight-gradient stream");
    hipEvent_t e = g_side.next();
    SWAT_REQUIRE(e, "cannot create an event");
    SWAT_HIP(hipEventRecord(e, g_side.st));
    SWAT_HIP(hipStreamWaitEvent((hipStream_t)stream, e, 0));
    if (g_side.aux_done) SWAT_HIP(hipStreamWaitEvent((hipStream_t)stream, g_side.aux_done, 0));
    return 0;
}

SWAT_API int swat_vit_lane_follow(swat_stream_t stream) {
    SWAT_REQUIRE(g_side.ensure() == 0, "swat_vit_lane_follow: cannot create the weight-gradient stream");
    hipEvent_t e = g_side.next();
    SWAT_REQUIRE(e, "cannot create an event");
    SWAT_HIP(hipEventRecord(e, (hipStream_t)stream));
    SWAT_HIP(hipStreamWaitEvent(g_side.st, e, 0));
    g_side.open = true;  // the caller joins (swat_vit_backward_join) before it reads what it queued on the lane
    return 0;
}

static int join_pending_lane(hipStream_t s) {
    if (g_side.pending && g_side.pend_lane) {
        SWAT_HIP(hipStreamWaitEvent(s, g_side.pend_lane, 0));
        g_side.pend_lane = nullptr;
    }
    return 0;
}

// The owed join of swat_vit_backward_join_later, paid on `stream` (no-op when nothing is owed).
SWAT_API int swat_vit_join_pending(swat_stream_t stream) {
    if (!g_side.pending) return 0;
    g_side.pending = false;
    if (g_side.pend_lane) SWAT_HIP(hipStreamWaitEvent((hipStream_t)stream, g_side.pend_lane, 0));
    if (g_side.pend_aux) SWAT_HIP(hipStreamWaitEvent((hipStream_t)stream, g_side.pend_aux, 0));
    g_side.pend_lane = g_side.pend_aux = nullptr;
    return 0;
}

// swat_vit_backward_join without making `stream` wait now: what the lane and the second stream still hold (the blocks' weight
// gradients and their AdamW) is only read again by the blocks of the NEXT forward, so the next step's stem — CutMix / patch
// extraction, patch-embedding product, class row, ln_pre: ~80 us that touch stage-0 parameters only, stepped on the main stream —
// may run beside the optimizer's tail.  swat_vit_forward pays the join behind its stem, swat_vit_backward / swat_transformer_forward
// / swat_vit_join_pending at their start.  A caller that reads the arena itself (checkpoint, validation through torch ops, logging)
// calls swat_vit_join_pending(its stream) first.
SWAT_API int swat_vit_backward_join_later(swat_stream_t stream) {
    TRY(flush_pending_dproj((hipStream_t)stream));
    TRY(swat_vit_join_pending(stream));   // (an older debt first: its events are about to be replaced)
    if (!g_side.open) return 0;
    hipEvent_t e = g_side.next();
    SWAT_REQUIRE(e, "cannot create an event");
    SWAT_HIP(hipEventRecord(e, g_side.st));
    g_side.pend_lane = e;
    g_side.pend_aux = g_side.aux_done;
    g_side.aux_done = nullptr;
    g_side.pending = true;
    g_side.open = false;
    g_side.doneA = g_side.doneB = nullptr;
    g_side.blk[0] = g_side.blk[1] = nullptr;
    return 0;
}

SWAT_API int swat_vit_backward_join(swat_stream_t stream) {
    TRY(flush_pending_dproj((hipStream_t)stream));
    TRY(swat_vit_join_pending(stream));
    if (!g_side.open) return 0;
    hipEvent_t e = g_side.next();
    SWAT_REQUIRE(e, "cannot create an event");
    SWAT_HIP(hipEventRecord(e, g_side.st));
    SWAT_HIP(hipStreamWaitEvent((hipStream_t)stream, e, 0));
    if (g_side.aux_done) {
        SWAT_HIP(hipStreamWaitEvent((hipStream_t)stream, g_side.aux_done, 0));
        g_side.aux_done = nullptr;
    }
    g_side.open = false;
    g_side.doneA = g_side.doneB = nullptr;
    g_side.blk[0] = g_side.blk[1] = nullptr;
    return 0;
}

// Backward in stages so the host can overlap gradient reduction with the remaining backward:
//   stage L+1 : proj + ln_post (needs dfeats);  stage l+1 : block l;  stage 0 : ln_pre + embeddings + conv1.
// Call with stage_hi >= stage_lo; stages run from stage_hi down to stage_lo.  Gradients are WRITTEN (not
// accumulated) into `grads` at the parameters' own offsets.
// AdamW inside the weight-gradient launch of the persistent backward (csrc/gemm_xl.hip): armed by the optimizer's owner for the NEXT
// swat_vit_backward call that covers all blocks; that call clears it whether it used it or not
struct FusedAdamState {
    bool armed = false, consumed = false;
    XlAdam ad{};
};
static FusedAdamState g_fused_adam;
static int g_block_adamw = -1;   // swat_vit_set_block_adamw / SWAT_BLOCK_ADAMW: what an armed AdamW becomes (0 ignored, 1 piped, 2 epilogue)

SWAT_API int swat_vit_arm_fused_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                                      const float* lr_dev, const float* bias_corr_dev, float beta1, float beta2, float eps,
                                      float weight_decay, float grad_scale) {
    SWAT_REQUIRE(params && grads && exp_avg && exp_avg_sq && shadow_bf16 && lr_dev && bias_corr_dev, "swat_vit_arm_fused_adamw: null pointer");
    XlAdam& a = g_fused_adam.ad;
    a.on = 1; a.gbase = grads; a.p = params; a.m = exp_avg; a.v = exp_avg_sq; a.shadow = (unsigned short*)shadow_bf16;
    a.lr = lr_dev; a.bc = bias_corr_dev; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay; a.gscale = grad_scale;
    a.guard = nullptr;
    g_fused_adam.armed = true;
    g_fused_adam.consumed = false;
    return 0;
}
// The NEXT swat_vit_backward call for stage L+1 alone may leave d proj (which only the optimizer reads) pending: the call for the
// blocks that follows issues it behind the persistent backward (any other backward / join call flushes it at once).  For callers
// that read the stage's gradients only after the blocks' call (swat_amd/step.py, one GPU).
SWAT_API int swat_vit_backward_defer_top(void) {
    g_defer_top = true;
    return 0;
}

SWAT_API int swat_vit_set_block_adamw(int mode) {
    SWAT_REQUIRE(mode >= 0 && mode <= 2, "swat_vit_set_block_adamw: mode must be 0, 1 or 2");
    g_block_adamw = mode;
    return 0;
}
// 1: the last swat_vit_backward over all blocks applied AdamW to every parameter of the blocks itself (the caller skips its own
// launch for that arena range); reading clears it
SWAT_API int swat_vit_fused_adamw_consumed(void) {
    const bool c = g_fused_adam.consumed;
    g_fused_adam.consumed = false;
    return c ? 1 : 0;
}

// Data-parallel form of the blocks' persistent backward (swat_amd/step.py at N > 1): the caller names BUCKETS of transformer blocks,
// from the top of the tower down; the next swat_vit_backward call that runs the blocks as the persistent launch then issues, on the
// lane, the LayerNorm reductions of all blocks and per bucket ONE weight-gradient launch (the joint tile list of its blocks) with an
// event behind it.  swat_vit_backward_bucket_wait(i, stream) orders `stream` (the caller's all-reduce stream) behind bucket i's
// event: the all-reduce + AdamW of a bucket then trail the weight gradients of the buckets below.  No AdamW inside the call.
struct DpBuckets {
    int n = 0, recorded = 0;
    int blocks[16];
    hipEvent_t ev[16] = {};
};
static DpBuckets g_dp;

SWAT_API int swat_vit_backward_set_buckets(const int32_t* blocks, int n) {
    SWAT_REQUIRE(n >= 0 && n <= 16 && (n == 0 || blocks), "swat_vit_backward_set_buckets: 0 .. 16 buckets");
    for (int i = 0; i < n; ++i) {
        SWAT_REQUIRE(blocks[i] >= 1, "swat_vit_backward_set_buckets: a bucket holds at least one block");
        g_dp.blocks[i] = blocks[i];
        if (!g_dp.ev[i]) SWAT_HIP(hipEventCreateWithFlags(&g_dp.ev[i], hipEventDisableTiming));
    }
    g_dp.n = n;
    g_dp.recorded = 0;
    return 0;
}
SWAT_API int swat_vit_backward_bucket_wait(int i, swat_stream_t stream) {
    SWAT_REQUIRE(i >= 0 && i < g_dp.recorded, "swat_vit_backward_bucket_wait: bucket %d has no event (the last backward recorded %d)", i, g_dp.recorded);
    SWAT_HIP(hipStreamWaitEvent((hipStream_t)stream, g_dp.ev[i], 0));
    return 0;
}
// buckets the last persistent backward recorded an event for (0: the blocks did not run as the persistent launch)
SWAT_API int swat_vit_backward_buckets_recorded(void) { return g_dp.recorded; }

SWAT_API int swat_vit_backward_blocks_fused(const int32_t* dims) {
    VitDims D;
    if (parse_dims(dims, D) != 0 || !D.save || g_group_wgrad) return 0;
    if (!vit_mega_backward_fits(D.B, D.T, D.d, D.H, D.mlp) || !vit_mega_backward_shape(D.B, D.T, D.d, D.H, D.mlp, D.L)) return 0;
    const int d = D.d, mlp = D.mlp;
    GemmArgs q[4] = {G(nullptr, nullptr, nullptr, d, mlp, D.Mp, d, mlp, mlp, 1), G(nullptr, nullptr, nullptr, mlp, d, D.Mp, mlp, d, d, 1),
                     G(nullptr, nullptr, nullptr, d, d, D.Mp, d, d, d, 1), G(nullptr, nullptr, nullptr, 3 * d, d, D.Mp, 3 * d, d, d, 1)};
    return wgrad_big_try(q, 4, nullptr, true) == 0 ? 1 : 0;
}

SWAT_API int swat_vit_backward(const int32_t* dims, const int64_t* offs, const float* params, const void* shadow_bf16,
                               void* workspace, const float* dfeats, float* grads, int stage_hi, int stage_lo,
                               swat_stream_t stream) {
    VitDims D;
    SWAT_REQUIRE(parse_dims(dims, D) == 0 && D.save, "swat_vit_backward: bad dims or forward ran without save");
    SWAT_REQUIRE(stage_hi <= D.L + 1 && stage_lo >= 0 && stage_hi >= stage_lo, "swat_vit_backward: bad stage range");
    const VitOffs O = parse_offs(offs);
    VitWs w = carve(D, workspace);
    hipStream_t s = (hipStream_t)stream;
    const unsigned short* sh = (const unsigned short*)shadow_bf16;
    const int M = D.M, d = D.d, T = D.T, np = D.B * D.g * D.g, mlp = D.mlp;

    TRY(swat_vit_join_pending(stream));
    const bool fork = g_overlap != 0;
    if (fork) SWAT_REQUIRE(g_side.ensure() == 0, "swat_vit_backward: cannot create the weight-gradient stream");
    hipStream_t ws = fork ? g_side.st : s;  // weight-gradient lane
    hipEvent_t &doneA = g_side.doneA, &doneB = g_side.doneB;
    if (fork) g_side.open = true;
    bool lane_ok = true;  // an event call failed: reported once the stages are enqueued
    auto to_lane = [&]() {  // everything issued so far on the main stream precedes what the lane runs next
        if (!fork) return;
        hipEvent_t e = g_side.next();
        if (!e || hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(ws, e, 0) != hipSuccess) lane_ok = false;
    };
    auto lane_done = [&](hipEvent_t& slot) {
        if (!fork) return;
        slot = g_side.next();
        if (!slot || hipEventRecord(slot, ws) != hipSuccess) lane_ok = false;
    };
    auto wait_lane = [&](hipEvent_t& slot) {
        if (!slot) return;
        if (hipStreamWaitEvent(s, slot, 0) != hipSuccess) lane_ok = false;
        slot = nullptr;
    };

    // One-launch form of a block's weight gradients (csrc/gemm_wgrad.hip) when the joint 256 x 128 tile list of its four products
    // is one round of the chip (ViT-B/32 and B/16 widths at B*T up to a few thousand rows): the lane gets ONE product launch per
    // block, behind the attention backward, and reads that block's ring slots of dxb / dfb / dxb2 / dqkv (carve) while the main
    // chain runs on.  Other shapes keep the two-pairs form below.
    bool one = false;
    if (D.L >= 1) {
        GemmArgs q[4] = {G(nullptr, nullptr, nullptr, d, mlp, D.Mp, d, mlp, mlp, 1), G(nullptr, nullptr, nullptr, mlp, d, D.Mp, mlp, d, d, 1),
                         G(nullptr, nullptr, nullptr, d, d, D.Mp, d, d, d, 1), G(nullptr, nullptr, nullptr, 3 * d, d, D.Mp, 3 * d, d, d, 1)};
        one = !g_group_wgrad && wgrad_big_try(q, 4, nullptr, true) == 0;
    }
    hipEvent_t* blk = g_side.blk;
    // Persistent backward (csrc/vit_mega.hip): the data-gradient chain of ALL blocks as one launch when the call covers them all
    // (stage_hi >= L, stage_lo <= 1) at the small-batch B/32 shape; the blocks' weight gradients and LayerNorm reductions follow
    // on the lane.  The bf16 copy of the top gradient goes where that launch reads it whenever it MAY run (the op-by-op path of
    // block L-1 reads the same slot).
    const bool mega_may = w.mb_dxb && one && vit_mega_backward_shape(D.B, T, d, D.H, mlp, D.L);
    bool mega = mega_may && stage_hi >= D.L && stage_lo <= 1;
    unsigned short* dxb_top = mega_may ? w.mb_dxb + (int64_t)(D.L - 1) * D.Mp * d : nullptr;
    unsigned short* dxb_head = dxb_top ? dxb_top : (one ? w.dxb_r[(D.L - 1) % 3] : w.dxb);

    if (!(mega && stage_hi <= D.L)) {
        // (a pending d proj of an earlier stage L+1 call and no persistent backward in this call to put it behind: now)
        if (stage_hi <= D.L) TRY(flush_pending_dproj(s));
    }
    int ln1_pending = -1, mega_rc = 0;
    void* ln1_ws = nullptr;
    if (stage_hi >= 1 && stage_lo <= D.L) g_dp.recorded = 0;   // (a blocks' call: what bucket_wait may refer to is decided below)
    if (!mega && stage_hi >= 1 && stage_lo <= D.L) g_dp.n = 0;  // bucket list of a caller whose shape is not served: dropped
    for (int stage = stage_hi; stage >= stage_lo; --stage) {
        if (stage == D.L + 1) {
            SWAT_REQUIRE(dfeats, "swat_vit_backward: dfeats is null");
            // dproj[c,e] = sum_b u[b,c] df[b,e] ; du[b,c] = sum_e df[b,e] proj[c,e]
            // (d proj feeds only the optimizer: when the blocks will run as the persistent backward it is issued BEHIND that launch —
            // one launch less in front of it; the pending product is flushed by whatever backward / join call comes next otherwise)
            const bool defer = g_defer_top && mega_may && stage_lo == D.L + 1;
            g_defer_top = false;
            if (defer) {
                g_pend_dproj = PendingProduct{true, w.u, dfeats, grads + O.proj, d, D.E, D.B};
            } else {
                TRY(swat_sgemm_small(w.u, dfeats, grads + O.proj, d, D.E, D.B, 1, d, D.E, 1, D.E, 1.0f, 0, s));
            }
            if (D.E == 512 || D.E == 768) {
                if (D.E == 512) hipLaunchKernelGGL(du_rows_kernel<2>, dim3((d + 63) / 64, D.B), dim3(256), 0, s, dfeats, params + O.proj, w.du, d, D.E);
                else hipLaunchKernelGGL(du_rows_kernel<3>, dim3((d + 63) / 64, D.B), dim3(256), 0, s, dfeats, params + O.proj, w.du, d, D.E);
                TRY(check_launch("du_rows_kernel"));
            } else {
                TRY(swat_sgemm_small(dfeats, params + O.proj, w.du, D.B, d, D.E, D.E, 1, 1, D.E, d, 1.0f, 0, s));
            }
            wait_lane(doneA);
            wait_lane(blk[0]);
            wait_lane(blk[1]);
            // ONE launch: LayerNorm backward of the class-token rows into dx / its bf16 copy, zeros in every other row (round 4: a zero
            // fill, the LayerNorm's row pass and its reduction — three launches on the chain between the two persistent kernels).  The
            // reduction of its partial sums (d ln_post.weight / bias, the last block's d c_proj.bias) feeds only the optimizer: with d
            // proj it waits for the call that runs the blocks and is issued behind the persistent backward.
            const float* xl = w.xs + (int64_t)D.L * M * d;
            TRY(swat_layernorm_bwd_partial_fill(w.du, d, xl, (int64_t)T * d, params + O.lnpost_w, w.mean_post, w.rstd_post, w.dx, dxb_head, T,
                                                w.lnws, D.B, d, s));
            if (defer) {
                g_pend_lnpost = PendingLnPost{true, w.lnws, D.B, d, grads + O.lnpost_w, grads + O.lnpost_b, grads + (O.layer + 12 * (D.L - 1))[PRB]};
            } else {
                TRY(swat_layernorm_bwd_finish(w.lnws, D.B, d, grads + O.lnpost_w, grads + O.lnpost_b, grads + (O.layer + 12 * (D.L - 1))[PRB], s));
            }
        } else if (stage >= 1 && mega && stage < D.L) {
            continue;   // ran inside the persistent launch of stage L
        } else if (stage == D.L && mega && [&]() -> bool {
                       const int rc = vit_mega_backward(D.B, T, d, D.H, mlp, D.L, D.act, M, D.Mp, params, sh, O.layer, w.xs, w.x1, w.qkv, w.f,
                                                        w.mean1, w.rstd1, w.mean2, w.rstd2, w.dx, w.mb_dxb, w.mb_dfb, w.mb_dxb2, w.mb_dqkv,
                                                        w.mb_dh2, w.mb_dh1, w.mb_lnp1, w.mb_lnp2, s);
                       if (rc < 0) mega_rc = rc;
                       if (rc != 0) mega = false;   // not eligible after all (no persistent forward ran): op by op from here
                       if (rc != 0 && flush_pending_dproj(s) != 0) mega_rc = -1;
                       return rc == 0;
                   }()) {
            // lane: per block the four weight gradients as one launch; the 2 L LayerNorm reductions as one launch behind them
            to_lane();
            const int nblk = ((D.B + 1) / 2) * 16;
            const int64_t lnp = vit_mega_backward_lnp_bytes(D.B, d) / 4;
            std::vector<const float*> parts;
            std::vector<float*> dgs, dbs_, dcs;
            std::vector<GemmArgs> wg;
            for (int l = D.L - 1; l >= 0; --l) {
                const int64_t* lo = O.layer + 12 * l;
                const void* dys[4] = {w.mb_dxb + (int64_t)l * D.Mp * d, w.mb_dfb + (int64_t)l * D.Mp * mlp, w.mb_dxb2 + (int64_t)l * D.Mp * d,
                                      w.mb_dqkv + (int64_t)l * D.Mp * 3 * d};
                const void* xs4[4] = {w.gact + (int64_t)l * D.Mp * mlp, w.h2 + (int64_t)l * D.Mp * d, w.a + (int64_t)l * D.Mp * d,
                                      w.h1 + (int64_t)l * D.Mp * d};
                float* dws[4] = {grads + lo[PRW], grads + lo[FCW], grads + lo[OUTW], grads + lo[QKVW]};
                float* dbs[4] = {nullptr, grads + lo[FCB], nullptr, grads + lo[QKVB]};
                const int nout[4] = {d, mlp, d, 3 * d}, nin[4] = {mlp, d, d, d};
                for (int i = 0; i < 4; ++i) {
                    GemmArgs a = G(dys[i], xs4[i], dws[i], nout[i], nin[i], D.Mp, nout[i], nin[i], nin[i], 1);
                    a.colsum = dbs[i];
                    wg.push_back(a);
                }
                parts.push_back(w.mb_lnp2 + l * lnp); dgs.push_back(grads + lo[LN2W]); dbs_.push_back(grads + lo[LN2B]); dcs.push_back(grads + lo[OUTB]);
                parts.push_back(w.mb_lnp1 + l * lnp); dgs.push_back(grads + lo[LN1W]); dbs_.push_back(grads + lo[LN1B]);
                dcs.push_back(l > 0 ? grads + (O.layer + 12 * (l - 1))[PRB] : nullptr);
            }
            // all 4 L weight gradients as one launch over the joint tile list (csrc/gemm_xl.hip); else one launch per block
            const XlAdam* adam = nullptr;
            if (g_fused_adam.armed && g_fused_adam.ad.p == params && g_fused_adam.ad.gbase == grads && g_fused_adam.ad.shadow == sh) {
                g_fused_adam.ad.guard = mega_guard_word();
                static const int stagger_us = getenv("SWAT_XL_STAGGER") ? atoi(getenv("SWAT_XL_STAGGER")) : 12;
                g_fused_adam.ad.stagger = 100 * stagger_us;
                adam = &g_fused_adam.ad;
            }
            g_fused_adam.armed = false;
            if (g_block_adamw < 0) g_block_adamw = getenv("SWAT_BLOCK_ADAMW") ? atoi(getenv("SWAT_BLOCK_ADAMW")) : 1;
            if (g_block_adamw == 0) adam = nullptr;
            const int adam_pipe = g_block_adamw == 1 ? 1 : 0;
            // Piped form on one GPU (round 6): the LayerNorm reductions (ln_post's column sums, then the 2 L sets of the blocks: 12 + 18 us)
            // used to open the LANE, in front of the top block's weight gradients — and the optimizer chain, which bounds this phase,
            // starts behind those.  They run on the optimizer's stream instead (idle until the first block's gradients are there), beside
            // the top block's weight gradients: the first AdamW of the blocks starts ~30 us earlier.
            const bool piped = adam && adam_pipe && fork && !(g_dp.n > 0);
            if (piped) {
                hipEvent_t e = g_side.next();
                if (!e || hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(g_side.aux[0], e, 0) != hipSuccess) lane_ok = false;
            }
            TRY(flush_pending_dproj(s, piped ? g_side.aux[0] : ws));   // (behind the persistent backward, beside the lane)
            if (g_dp.n > 0 && fork) {
                // data-parallel: LayerNorm reductions of every block first (their vectors ride in the blocks' buckets), then one
                // weight-gradient launch + event per bucket, top of the tower first; the last bucket takes whatever blocks remain
                TRY(ln_bwd_finish_sets(parts.data(), dgs.data(), dbs_.data(), dcs.data(), (int)parts.size(), nblk, d, ws));
                size_t l4 = 0;
                int nb = 0;
                for (; nb < g_dp.n && l4 < wg.size(); ++nb) {
                    const int left = (int)((wg.size() - l4) / 4);
                    const int nl = (nb + 1 == g_dp.n || g_dp.blocks[nb] > left) ? left : g_dp.blocks[nb];
                    int wrc1 = nl > 1 ? wgrad_all_try(wg.data() + l4, 4 * nl, ws, nullptr) : 1;
                    if (wrc1 < 0) return wrc1;
                    for (int b4 = 0; wrc1 == 1 && b4 < nl; ++b4) {
                        const void* dys[4]; const void* xs4[4]; float* dws[4]; float* dbs[4];
                        int rws[4], nout[4], nin[4], ldy[4], ldx[4];
                        for (int i = 0; i < 4; ++i) {
                            const GemmArgs& a = wg[l4 + 4 * b4 + i];
                            dys[i] = a.A; xs4[i] = a.B; dws[i] = (float*)a.C; dbs[i] = a.colsum;
                            rws[i] = a.K; nout[i] = a.M; nin[i] = a.N; ldy[i] = a.lda; ldx[i] = a.ldb;
                        }
                        TRY(swat_linear_wgrad_bias_group(4, dys, xs4, dws, dbs, rws, nout, nin, ldy, ldx, w.csws, ws));
                    }
                    if (hipEventRecord(g_dp.ev[nb], ws) != hipSuccess) lane_ok = false;
                    l4 += (size_t)4 * nl;
                }
                g_dp.recorded = nb;
                g_dp.n = 0;   // one call
                continue;
            }
            if (adam && adam_pipe && fork) {
                // Piped form (the default): per block, the weight-gradient launch on the lane and the block's AdamW on a second stream —
                // beside the NEXT block's weight gradients.  The 256 x 128 kernel is bound by the matrix cores and holds 3 x 152 of a
                // SIMD's 512 registers: one 52-register wave of the HBM-bound AdamW fits beside its three (measured: 12 x (33 + 36) us
                // one after the other -> 12 x ~51 us; the epilogue form below — AdamW on the accumulators — spares the gradient's
                // round trip through memory but leaves its IEEE divisions and square roots to 8 waves per CU: no faster than apart)
                hipStream_t os = g_side.aux[0];
                TRY(ln_bwd_finish_sets(parts.data(), dgs.data(), dbs_.data(), dcs.data(), (int)parts.size(), nblk, d, os));
                const int64_t stride = D.L > 1 ? O.layer[12] - O.layer[0] : 0;
                const int64_t zero = 0;
                static const int pipe_grp = getenv("SWAT_PIPE_GROUP") ? atoi(getenv("SWAT_PIPE_GROUP")) : 1;
                const int grp = pipe_grp < 1 ? 1 : pipe_grp > 12 ? 12 : pipe_grp;   // blocks per weight-gradient launch
                for (size_t l4 = 0; l4 < wg.size(); l4 += 4 * grp) {
                    const int l_hi = D.L - 1 - (int)(l4 / 4);
                    const int nl = l_hi + 1 < grp ? l_hi + 1 : grp, l_lo = l_hi - nl + 1;
                    int wrc1 = nl > 1 ? wgrad_all_try(wg.data() + l4, 4 * nl, ws, nullptr) : 1;   // (table form: several blocks per launch)
                    if (wrc1 < 0) return wrc1;
                    for (int b4 = 0; wrc1 == 1 && b4 < nl; ++b4) {
                        const void* dys[4]; const void* xs4[4]; float* dws[4]; float* dbs[4];
                        int rws[4], nout[4], nin[4], ldy[4], ldx[4];
                        for (int i = 0; i < 4; ++i) {
                            const GemmArgs& a = wg[l4 + 4 * b4 + i];
                            dys[i] = a.A; xs4[i] = a.B; dws[i] = (float*)a.C; dbs[i] = a.colsum;
                            rws[i] = a.K; nout[i] = a.M; nin[i] = a.N; ldy[i] = a.lda; ldx[i] = a.ldb;
                        }
                        TRY(swat_linear_wgrad_bias_group(4, dys, xs4, dws, dbs, rws, nout, nin, ldy, ldx, w.csws, ws));
                    }
                    hipEvent_t e = g_side.next();
                    if (!e || hipEventRecord(e, ws) != hipSuccess || hipStreamWaitEvent(os, e, 0) != hipSuccess) lane_ok = false;
                    const int64_t b = O.layer[12 * l_lo + LN1W];
                    const int64_t n = l_hi + 1 < D.L ? stride * nl : stride * (nl - 1) + O.layer[12 * l_hi + PRB] + d - O.layer[12 * l_hi + LN1W];
                    TRY(swat_adamw(adam->p + b, adam->gbase + b, adam->m + b, adam->v + b, adam->shadow + b, n, &zero, 1, adam->lr, adam->b1, adam->b2,
                                   adam->eps, adam->wd, adam->bc, adam->gscale, os));
                }
                // the second stream's tail is joined with the lane's (swat_vit_backward_join / swat_vit_lane_order), not in front of what
                // the caller queues on the lane next: the small optimizer launches of the head run beside the last block's AdamW
                hipEvent_t e2 = g_side.next();
                if (!e2 || hipEventRecord(e2, os) != hipSuccess) lane_ok = false;
                g_side.aux_done = e2;
                g_fused_adam.consumed = true;
                continue;
            }
            int wrc = (int)wg.size() <= 64 ? wgrad_all_try(wg.data(), (int)wg.size(), ws, adam_pipe ? nullptr : adam) : 1;
            if (wrc < 0) return wrc;
            const bool fused_adam = wrc == 0 && adam && !adam_pipe;
            if (wrc == 1) {
                for (size_t l4 = 0; l4 < wg.size(); l4 += 4) {
                    const void* dys[4]; const void* xs4[4]; float* dws[4]; float* dbs[4];
                    int rws[4], nout[4], nin[4], ldy[4], ldx[4];
                    for (int i = 0; i < 4; ++i) {
                        const GemmArgs& a = wg[l4 + i];
                        dys[i] = a.A; xs4[i] = a.B; dws[i] = (float*)a.C; dbs[i] = a.colsum;
                        rws[i] = a.K; nout[i] = a.M; nin[i] = a.N; ldy[i] = a.lda; ldx[i] = a.ldb;
                    }
                    TRY(swat_linear_wgrad_bias_group(4, dys, xs4, dws, dbs, rws, nout, nin, ldy, ldx, w.csws, ws));
                }
            }
            TRY(ln_bwd_finish_sets(parts.data(), dgs.data(), dbs_.data(), dcs.data(), (int)parts.size(), nblk, d, ws));
            if (fused_adam) {
                // the matrices are done; biases and LayerNorm weights of the blocks (their gradients are complete now): one launch
                std::vector<int64_t> rb;
                std::vector<int> rl;
                for (int l = 0; l < D.L; ++l) {
                    const int64_t* lo = O.layer + 12 * l;
                    const int which[8] = {LN1W, LN1B, QKVB, OUTB, LN2W, LN2B, FCB, PRB};
                    const int len[8] = {d, d, 3 * d, d, d, d, mlp, d};
                    for (int k = 0; k < 8; ++k) {
                        rb.push_back(lo[which[k]]);
                        rl.push_back(len[k]);
                    }
                }
                for (size_t r0 = 0; r0 < rb.size(); r0 += 128) {
                    const int n = (int)(rb.size() - r0 < 128 ? rb.size() - r0 : 128);
                    const int rc = launch_adamw_ranges(rb.data() + r0, rl.data() + r0, n, *adam, ws);
                    SWAT_REQUIRE(rc == 0, "swat_vit_backward: AdamW of the blocks' vector parameters failed to launch");
                }
                g_fused_adam.consumed = true;
            }
            continue;
        } else if (stage >= 1) {
            SWAT_REQUIRE(mega_rc == 0, "swat_vit_backward: the persistent backward failed to launch");
            const int l = stage - 1;
            const int64_t* lo = O.layer + 12 * l;
            const float* xin = w.xs + (int64_t)l * M * d;
            const float* x1 = w.x1 + (int64_t)l * M * d;
            const unsigned short* h1 = w.h1 + (int64_t)l * D.Mp * d;
            const unsigned short* h2 = w.h2 + (int64_t)l * D.Mp * d;
            const unsigned short* at = w.a + (int64_t)l * D.Mp * d;
            const unsigned short* qkv = w.qkv + (int64_t)l * M * 3 * d;
            const unsigned short* P = w.P + (int64_t)l * D.B * D.H * T * D.ldp;
            const unsigned short* f = w.f + (int64_t)l * M * mlp;
            const unsigned short* ga = w.gact + (int64_t)l * D.Mp * mlp;
            if (one) {
                const int si = l & 1;
                unsigned short* dxb_in = (dxb_top && l == D.L - 1) ? dxb_top : w.dxb_r[l % 3];
                unsigned short* dxb_out = w.dxb_r[(l + 2) % 3];   // block l - 1's slot
                unsigned short *dfb = w.dfb_r[si], *dxb2 = w.dxb2_r[si], *dqkv = w.dqkv_r[si];
                // this block's slots were last read by the lane work of block l + 2 (same parity; dxb_out: (l + 2) % 3 too)
                wait_lane(blk[si]);
                // c_proj: dF = (dx W) * act'(F)   (c_proj bias gradient = column sums of dx: from the LayerNorm backward that wrote dx)
                {
                    GemmArgs a = G(dxb_in, sh + lo[PRW], dfb, M, mlp, d, d, mlp, mlp, 0);
                    a.actin = f; a.ldact = mlp; a.act = D.act;
                    TRY(gemm_dispatch(L_NN, a, 1, s));
                }
                // c_fc: dh2 = dF W
                int S2 = 1;
                TRY(dgrad_split(D, dfb, mlp, sh + lo[FCW], w.dh, S2, s));
                // ln_2 backward: dx1 = dx + LN'(dh2) (in place on dx) + bf16 copy; its partial sums are reduced on the lane
                TRY(swat_layernorm_bwd_partial(w.dh, d, S2, (int64_t)M * d, x1, d, params + lo[LN2W], w.mean2 + (int64_t)l * M, w.rstd2 + (int64_t)l * M, w.dx, d,
                                               w.dx, d, dxb2, d, w.lnws2_r[si], M, d, s));
                // out_proj: da = dx1 W
                TRY(gemm_dispatch(L_NN, G(dxb2, sh + lo[OUTW], w.dab, M, d, d, d, d, d, 0), 1, s));
                // attention backward -> dqkv
                TRY(attn_backward(D, qkv, P, w.dab, w.S, w.dS, dqkv, s, at, w.lse + (int64_t)l * D.B * D.H * T));
                to_lane();
                // in_proj: dh1 = dqkv W (enqueued before the lane's launches: the host needs a few us for them)
                int S1 = 1;
                TRY(dgrad_split(D, dqkv, 3 * d, sh + lo[QKVW], w.dh, S1, s));
                // lane: the block's four weight gradients (+ the c_fc / in_proj bias gradients: column sums of dF / dqkv) as one
                // launch, then the ln_2 reductions (d ln_2.weight / bias, d out_proj.bias)
                {
                    const void* dys[4] = {dxb_in, dfb, dxb2, dqkv};
                    const void* xs4[4] = {ga, h2, at, h1};
                    float* dws[4] = {grads + lo[PRW], grads + lo[FCW], grads + lo[OUTW], grads + lo[QKVW]};
                    float* dbs[4] = {nullptr, grads + lo[FCB], nullptr, grads + lo[QKVB]};
                    const int rws[4] = {D.Mp, D.Mp, D.Mp, D.Mp}, nout[4] = {d, mlp, d, 3 * d}, nin[4] = {mlp, d, d, d};
                    const int ldy[4] = {d, mlp, d, 3 * d}, ldx[4] = {mlp, d, d, d};
                    TRY(swat_linear_wgrad_bias_group(4, dys, xs4, dws, dbs, rws, nout, nin, ldy, ldx, w.csws, ws));
                }
                TRY(swat_layernorm_bwd_finish(w.lnws2_r[si], M, d, grads + lo[LN2W], grads + lo[LN2B], grads + lo[OUTB], ws));
                // ln_1 backward: dx0 = dx1 + LN'(dh1), bf16 copy into block l - 1's slot
                TRY(swat_layernorm_bwd_partial(w.dh, d, S1, (int64_t)M * d, xin, d, params + lo[LN1W], w.mean1 + (int64_t)l * M, w.rstd1 + (int64_t)l * M, w.dx, d,
                                               w.dx, d, dxb_out, d, w.lnws1_r[si], M, d, s));
                ln1_pending = l;
                ln1_ws = w.lnws1_r[si];
            } else {
                // c_proj: dF = (dx W) * act'(F); its weight gradient joins c_fc's on the lane below (one fork per pair)
                // (c_proj bias gradient = column sums of dx: produced by the LayerNorm backward that wrote dx)
                {
                    GemmArgs a = G(w.dxb, sh + lo[PRW], w.dfb, M, mlp, d, d, mlp, mlp, 0);
                    a.actin = f; a.ldact = mlp; a.act = D.act;
                    TRY(gemm_dispatch(L_NN, a, 1, s));
                }
                // lane: c_proj dW = dx^T g and c_fc dW = dF^T h2, db = colsum(dF) (reductions over Mp rows: pad rows are zero)
                to_lane();
                // c_fc: dh2 = dF W — enqueued BEFORE the lane's launches: the host needs ~20 us for those five calls, and a main chain
                // that waits for them leaves the GPU idle (tools/step_timeline.py: 7 us per block behind the 12-us attention backward)
                int S2 = 1;
                TRY(dgrad_split(D, w.dfb, mlp, sh + lo[FCW], w.dh, S2, s));
                {
                    const void* dys[2] = {w.dxb, w.dfb};
                    const void* xs2[2] = {ga, h2};
                    float* dws[2] = {grads + lo[PRW], grads + lo[FCW]};
                    float* dbs[2] = {nullptr, grads + lo[FCB]};
                    const int rws[2] = {D.Mp, D.Mp}, nout[2] = {d, mlp}, nin[2] = {mlp, d}, ldy[2] = {d, mlp}, ldx[2] = {mlp, d};
                    if (g_group_wgrad) {
                        TRY(swat_linear_wgrad_bias_group(2, dys, xs2, dws, dbs, rws, nout, nin, ldy, ldx, w.csws, ws));
                    } else {
                        TRY(swat_linear_wgrad_bias_group(1, dys, xs2, dws, dbs, rws, nout, nin, ldy, ldx, w.csws, ws));
                        TRY(swat_linear_wgrad_bias_group(1, dys + 1, xs2 + 1, dws + 1, dbs + 1, rws + 1, nout + 1, nin + 1, ldy + 1, ldx + 1, w.csws, ws));
                    }
                }
                lane_done(doneA);  // last lane read of dxb and dfb
                // ln_2 backward: dx1 = dx + LN'(dh2)   (in place on dx) + bf16 copy
                wait_lane(doneB);
                // (row pass here; the reduction of its partial sums to d(ln_2.weight / bias) and d(out_proj.bias) runs on the lane)
                TRY(swat_layernorm_bwd_partial(w.dh, d, S2, (int64_t)M * d, x1, d, params + lo[LN2W], w.mean2 + (int64_t)l * M, w.rstd2 + (int64_t)l * M, w.dx, d,
                                               w.dx, d, w.dxb2, d, w.lnws2, M, d, s));
                // out_proj: da = dx1 W (its weight gradient joins in_proj's on the lane below; bias gradient: from the ln_2 backward)
                TRY(gemm_dispatch(L_NN, G(w.dxb2, sh + lo[OUTW], w.dab, M, d, d, d, d, d, 0), 1, s));
                // attention backward -> dqkv
                TRY(attn_backward(D, qkv, P, w.dab, w.S, w.dS, w.dqkv, s, at, w.lse + (int64_t)l * D.B * D.H * T));
                // lane: out_proj dW = dx1^T a and in_proj dW = dqkv^T h1, db = colsum(dqkv)
                to_lane();
                // in_proj: dh1 = dqkv W (enqueued before the lane's launches, as above)
                int S1 = 1;
                TRY(dgrad_split(D, w.dqkv, 3 * d, sh + lo[QKVW], w.dh, S1, s));
                TRY(swat_layernorm_bwd_finish(w.lnws2, M, d, grads + lo[LN2W], grads + lo[LN2B], grads + lo[OUTB], ws));
                {
                    const void* dys[2] = {w.dxb2, w.dqkv};
                    const void* xs2[2] = {at, h1};
                    float* dws[2] = {grads + lo[OUTW], grads + lo[QKVW]};
                    float* dbs[2] = {nullptr, grads + lo[QKVB]};
                    const int rws[2] = {D.Mp, D.Mp}, nout[2] = {d, 3 * d}, nin[2] = {d, d}, ldy[2] = {d, 3 * d}, ldx[2] = {d, d};
                    if (g_group_wgrad) {
                        TRY(swat_linear_wgrad_bias_group(2, dys, xs2, dws, dbs, rws, nout, nin, ldy, ldx, w.csws, ws));
                    } else {
                        TRY(swat_linear_wgrad_bias_group(1, dys, xs2, dws, dbs, rws, nout, nin, ldy, ldx, w.csws, ws));
                        TRY(swat_linear_wgrad_bias_group(1, dys + 1, xs2 + 1, dws + 1, dbs + 1, rws + 1, nout + 1, nin + 1, ldy + 1, ldx + 1, w.csws, ws));
                    }
                }
                lane_done(doneB);  // last lane read of dxb2 and dqkv
                // ln_1 backward: dx0 = dx1 + LN'(dh1)
                wait_lane(doneA);
                TRY(swat_layernorm_bwd_partial(w.dh, d, S1, (int64_t)M * d, xin, d, params + lo[LN1W], w.mean1 + (int64_t)l * M, w.rstd1 + (int64_t)l * M, w.dx, d,
                                               w.dx, d, w.dxb, d, w.lnws1, M, d, s));
                ln1_pending = l;  // its reduction is queued on the lane at the end of this stage
                ln1_ws = w.lnws1;
            }
        } else {
            // stem: ln_pre backward, positional / class embedding, conv1 weight
            TRY(swat_layernorm_bwd(w.dx, d, w.xpre, d, params + O.lnpre_w, w.mean_pre, w.rstd_pre, nullptr, 0, w.dh, d, nullptr, 0,
                                   grads + O.lnpre_w, grads + O.lnpre_b, nullptr, w.lnws, M, d, s));
            TRY(swat_colsum(w.dh, SWAT_F32, D.B, T * d, T * d, grads + O.pos, 0, 0, 0, s));  // dpos[t,:] = sum_b
            TRY(swat_colsum(w.dh, SWAT_F32, D.B, d, T * d, grads + O.cls, 0, 0, 0, s));      // dcls = sum_b row 0
            // (one-launch form: the lane may still read every dxb slot — the attention-gradient buffer is the main chain's own)
            unsigned short* gsc = one ? w.dab : w.dxb;
            wait_lane(doneA);
            TRY(swat_gather_cast_bf16(w.dh, gsc, np, d, D.g * D.g, T, 1, s));              // drop the cls rows
            TRY(gemm_dispatch(L_TN, G(gsc, w.patches, grads + O.conv1, d, D.Kp, np, d, D.ldk, D.Kp, 1), 1, s));
        }
        to_lane();  // the lane's tail now also follows everything this stage ran on the main stream
        if (ln1_pending >= 0) {
            const int64_t* lo = O.layer + 12 * ln1_pending;
            TRY(swat_layernorm_bwd_finish(ln1_ws, M, d, grads + lo[LN1W], grads + lo[LN1B],
                                          ln1_pending > 0 ? grads + (O.layer + 12 * (ln1_pending - 1))[PRB] : nullptr, ws));
            if (one) lane_done(blk[ln1_pending & 1]);   // end of this block's lane work (the caller's AdamW slice comes behind it)
            ln1_pending = -1;
        }
    }
    SWAT_REQUIRE(lane_ok, "swat_vit_backward: a stream-ordering call (hipEventRecord / hipStreamWaitEvent) failed");
    if (g_overlap == 1) TRY(swat_vit_backward_join(stream));  // the caller's stream sees every gradient of the requested stages
    return 0;
}

// A3 + A4 + A6 and their backward in one call: feats [B,E] -> loss, dfeats, dW_head, dlogit_scale.
//   fn = feats/||feats|| ; z = fn W^T ; logits = z*exp(ls) ; loss = lam*CE(logits,ya) + (1-lam)*CE(logits,yb)
// workspace: (B*E + B + B*C + B*C + B*E + 2*B) floats.  want_grad = 0 -> forward only (validate).
SWAT_API int64_t swat_head_loss_workspace_bytes(int B, int E, int C) {
    return ((int64_t)2 * B * E + (int64_t)2 * B * C + 3 * B) * 4 + 1024;
}

static int g_head_rows = getenv("SWAT_HEAD_ROWS") ? atoi(getenv("SWAT_HEAD_ROWS")) : 1;   // 0: the five separate launches (A/B runs)
SWAT_API int swat_head_set_fused(int on) {
    g_head_rows = on ? 1 : 0;
    return 0;
}

SWAT_API int swat_head_loss(const float* feats, const float* W, const float* logit_scale_dev, const int64_t* ya,
                            const int64_t* yb, float lam, void* workspace, float* loss, float* dfeats, float* dW,
                            float* dlogit_scale, float* z_out, float* feat_norm_out, double* loss_accum, int B, int E,
                            int C, int want_grad, int normalize, swat_stream_t s) {
    SWAT_REQUIRE(feats && W && logit_scale_dev && ya && yb && workspace && loss, "swat_head_loss: null pointer");
    float* fn = (float*)workspace;
    float* nrm = fn + (int64_t)B * E;
    float* z = nrm + B;
    float* dz = z + (int64_t)B * C;
    float* dfn = dz + (int64_t)B * C;
    float* ps = dfn + (int64_t)B * E;
    if (feat_norm_out) fn = feat_norm_out;
    if (z_out) z = z_out;
    if (!normalize) fn = const_cast<float*>(feats);  // pre-extracted, already unit-norm features (train_probing, ref utils/training.py:96-97)
    // want_grad = 2 / 3: the call in two parts.  2 = what the backward waits for (normalise, logits, per-sample CE and d logits, d feats);
    // 3 = the rest, on the same workspace: the batch means (loss, d logit_scale, running loss sum) and d W_head, which only the
    // optimizer and the logger read — the step issues it behind the tower's backward (two launches off the critical path)
    if (want_grad == 3) {
        SWAT_REQUIRE(dW && dlogit_scale, "swat_head_loss: gradient outputs are null");
        TRY(launch_ce_finish(ps, B, loss, dlogit_scale, loss_accum, (hipStream_t)s));
        TRY(swat_sgemm_small(dz, fn, dW, C, E, B, 1, C, E, 1, E, 1.0f, 0, s));   // dW[c,e]  = sum_b dz[b,c] fn[b,e]
        return 0;
    }
    if (want_grad && normalize && dfeats && g_head_rows) {
        // training: the sample-local path as one launch (csrc/tail.hip head_rows_kernel); the batch-wide rest below / in part 3
        const int rc = launch_head_rows(feats, W, logit_scale_dev, ya, yb, lam, fn, nrm, z, dz, ps, dfeats, B, E, C, (hipStream_t)s);
        if (rc < 0) return rc;
        if (rc == 0) {
            SWAT_REQUIRE(dW && dlogit_scale, "swat_head_loss: gradient outputs are null");
            if (want_grad != 2) {
                TRY(launch_ce_finish(ps, B, loss, dlogit_scale, loss_accum, (hipStream_t)s));
                TRY(swat_sgemm_small(dz, fn, dW, C, E, B, 1, C, E, 1, E, 1.0f, 0, s));   // dW[c,e]  = sum_b dz[b,c] fn[b,e]
            }
            return 0;
        }
    }
    if (normalize) TRY(swat_l2norm_fwd(feats, fn, nrm, B, E, s));
    TRY(swat_sgemm_small(fn, W, z, B, C, E, E, 1, 1, E, C, 1.0f, 0, s));  // z[b,c] = sum_e fn[b,e] W[c,e]
    TRY(launch_ce_rows(z, logit_scale_dev, ya, yb, lam, ps, want_grad ? dz : nullptr, nullptr, B, C, (hipStream_t)s));
    if (want_grad != 2) TRY(launch_ce_finish(ps, B, loss, dlogit_scale, loss_accum, (hipStream_t)s));
    if (want_grad) {
        SWAT_REQUIRE((dfeats || !normalize) && dW && dlogit_scale, "swat_head_loss: gradient outputs are null");
        if (want_grad != 2) TRY(swat_sgemm_small(dz, fn, dW, C, E, B, 1, C, E, 1, E, 1.0f, 0, s));   // dW[c,e]  = sum_b dz[b,c] fn[b,e]
        TRY(swat_sgemm_small(dz, W, dfn, B, E, C, C, 1, E, 1, E, 1.0f, 0, s));   // dfn[b,e] = sum_c dz[b,c] W[c,e]
        if (normalize) TRY(swat_l2norm_bwd(fn, nrm, dfn, dfeats, B, E, s));
    }
    return 0;
}

SWAT_API int swat_embed_tokens(const int64_t* tokens, const float* table, const float* pos, float* out, int64_t rows, int T,
                               int d, swat_stream_t s) {
    SWAT_REQUIRE(d % 4 == 0, "swat_embed_tokens: d must be a multiple of 4");
    if (rows <= 0) return 0;
    int64_t blocks = (rows * (d / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, tokens, table, pos, out, rows, T, d);
    return check_launch("swat_embed_tokens");
}

// Forward-only stack of L pre-LN residual blocks on x [B*T, d] fp32 (in place): the CLIP text tower (causal = 1; ref:
// model.encode_text, utils/features.py:27-36, retrieval/extract_mined_feature.py:119-121) or any frozen tower.
// dims (HOST int32[8]): B, T, d, L, H, mlp, act, causal.  offs: 12 per block, as in swat_vit_forward.
SWAT_API int64_t swat_transformer_workspace_bytes(const int32_t* dims) {
    const int64_t B = dims[0], T = dims[1], d = dims[2], H = dims[4], mlp = dims[5];
    const int64_t M = B * T, ldp = (T + 7) / 8 * 8;
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    return al(M * d * 2) * 2 + al(M * 3 * d * 2) + al(B * H * T * ldp * 4) + al(B * H * T * ldp * 2) + al(M * mlp * 2) * 2 + al(M * d * 4) + 1024;
}

SWAT_API int swat_transformer_forward(const int32_t* dims, const int64_t* offs, const float* params, const void* shadow_bf16,
                                      float* x, void* workspace, swat_stream_t stream) {
    VitDims D{};
    D.B = dims[0]; D.T = dims[1]; D.d = dims[2]; D.L = dims[3]; D.H = dims[4]; D.mlp = dims[5]; D.act = dims[6];
    const int causal = dims[7];
    SWAT_REQUIRE(D.B > 0 && D.T > 0 && D.d % 64 == 0 && D.H > 0 && D.d % D.H == 0 && (D.d / D.H) % 8 == 0, "swat_transformer_forward: bad dims");
    D.M = D.B * D.T; D.dh = D.d / D.H; D.ldp = (D.T + 7) / 8 * 8;
    const int M = D.M, d = D.d;
    hipStream_t s = (hipStream_t)stream;
    const unsigned short* sh = (const unsigned short*)shadow_bf16;
    char* p = (char*)workspace;
    auto take = [&](int64_t bytes) { void* r = p; p += (bytes + 255) / 256 * 256; return r; };
    unsigned short* h = (unsigned short*)take((int64_t)M * d * 2);
    unsigned short* at = (unsigned short*)take((int64_t)M * d * 2);
    unsigned short* qkv = (unsigned short*)take((int64_t)M * 3 * d * 2);
    float* S = (float*)take((int64_t)D.B * D.H * D.T * D.ldp * 4);
    unsigned short* P = (unsigned short*)take((int64_t)D.B * D.H * D.T * D.ldp * 2);
    unsigned short* f = (unsigned short*)take((int64_t)M * D.mlp * 2);
    unsigned short* ga = (unsigned short*)take((int64_t)M * D.mlp * 2);
    float* x1 = (float*)take((int64_t)M * d * 4);
    for (int l = 0; l < D.L; ++l) {
        const int64_t* lo = offs + 12 * l;
        TRY(swat_layernorm_fwd(x, d, params + lo[LN1W], params + lo[LN1B], h, nullptr, nullptr, nullptr, M, d, 1e-5f, s));
        {
            GemmArgs a = G(h, sh + lo[QKVW], qkv, M, 3 * d, d, d, d, 3 * d, 0);
            a.bias = params + lo[QKVB];
            TRY(gemm_dispatch(L_NT, a, 1, s));
        }
        TRY(attn_forward(D, qkv, S, P, at, s, causal));
        {
            GemmArgs a = G(at, sh + lo[OUTW], x1, M, d, d, d, d, d, 1);
            a.bias = params + lo[OUTB]; a.residual = x;
            TRY(gemm_dispatch(L_NT, a, 1, s));
        }
        TRY(swat_layernorm_fwd(x1, d, params + lo[LN2W], params + lo[LN2B], h, nullptr, nullptr, nullptr, M, d, 1e-5f, s));
        {
            GemmArgs a = G(h, sh + lo[FCW], f, M, D.mlp, d, d, d, D.mlp, 0);
            a.bias = params + lo[FCB]; a.act = D.act; a.C2 = ga;
            a.c2_only = 1;  // a frozen tower: no backward re-reads the pre-activation
            TRY(gemm_dispatch(L_NT, a, 1, s));
        }
        {
            GemmArgs a = G(ga, sh + lo[PRW], x, M, d, D.mlp, D.mlp, D.mlp, d, 1);
            a.bias = params + lo[PRB]; a.residual = x1;
            TRY(gemm_dispatch(L_NT, a, 1, s));
        }
    }
    return 0;
}
