// swat_amd/csrc/gemm.hip — bf16 MFMA GEMM for gfx950 with fused epilogues; the workhorse of path A.
//
// Replaces the cuBLAS calls behind open_clip's nn.Linear / nn.MultiheadAttention / conv1 forward and
// their autograd backward (ref: model.encode_image call sites utils/training.py:332,1474).
//
//   C[M,N] = op(A) * op(B)      fp32 accumulate in v_mfma_f32_16x16x32_bf16
//   layout NT: A[M,K] (k contiguous)  B[N,K] (k contiguous)      forward   Y = X W^T
//   layout NN: A[M,K] (k contiguous)  B[K,N] (n contiguous)      dgrad     dX = dY W
//   layout TN: A[K,M] (m contiguous)  B[K,N] (n contiguous)      wgrad     dW = dY^T X
//
// Structure: 256 threads = 4 waves (2x2), block tile BMxBN (128x128 or 64x64), BK = 32, register-staged
// global->LDS with two LDS buffers (loads for step t+1 are issued before the MFMAs of step t).
// Operands whose reduction index is contiguous live in LDS as [rows][32] with a 16-B-slot XOR swizzle and
// are read with ds_read_b64; operands whose reduction index is strided live as [32][rows+16] and are read
// with ds_read_b64_tr_b16 (hardware transpose).  Both use ONE logical-k assignment per fragment element
// (j<4 -> k=4g+j, j>=4 -> k=16+4g+j-4, g = lane>>4) so that every LDS read is bank-conflict-free.
// The MFMA is issued as mfma(Bfrag, Afrag) so each lane ends up with 4 CONSECUTIVE n of one row m:
// epilogue loads/stores are 8-16 B per lane.
#include "kernels.hpp"
#include "../../include/swat_hip.h"

namespace swat {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ float act_fwd(float x, int act) {
    if (act == 1) return x / (1.0f + __expf(-1.702f * x));
    if (act == 2) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    return x;
}
__device__ __forceinline__ float act_bwd(float x, int act) {
    if (act == 1) {
        const float s = 1.0f / (1.0f + __expf(-1.702f * x));
        return s * (1.0f + 1.702f * x * (1.0f - s));
    }
    if (act == 2) {
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
        const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
        return cdf + x * pdf;
    }
    return 1.0f;
}

template <int LAYOUT, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmArgs p) {
    constexpr bool A_KC = (LAYOUT != L_TN);  // A reduction index contiguous
    constexpr bool B_KC = (LAYOUT == L_NT);
    constexpr int BK = 32;
    constexpr int WM = BM / 2, WN = BN / 2;  // wave tile
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int RSA = 2 * BM + 32, RSB = 2 * BN + 32;          // strided-tile row stride (bytes)
    constexpr int A_BYTES = A_KC ? BM * 64 : 32 * RSA;
    constexpr int B_BYTES = B_KC ? BN * 64 : 32 * RSB;
    constexpr int A_CH = BM * BK / 8 / 256;  // 16-B chunks per thread per step
    constexpr int B_CH = BN * BK / 8 / 256;
    static_assert(A_CH >= 1 && B_CH >= 1, "tile too small");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM;
    const int wn0 = (wave & 1) * WN;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const int zo = z / p.inner, zi = z % p.inner;
    const unsigned short* __restrict__ A = p.A + zo * p.sA0 + zi * p.sA1;
    const unsigned short* __restrict__ B = p.B + zo * p.sB0 + zi * p.sB1;
    const int M = p.M, N = p.N, K = p.K;

    // ---- staging: per-thread chunk coordinates ----
    s16x8_t ra[A_CH], rb[B_CH];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int c = tid + 256 * i;
            if constexpr (A_KC) {
                const int row = c >> 2, t = c & 3;
                int gm = m0 + row; gm = gm < M ? gm : M - 1;
                const int gk = k0 + 8 * t;
                s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (gk < K) {
                    v = *reinterpret_cast<const s16x8_t*>(A + (int64_t)gm * p.lda + gk);
                    if (gk + 8 > K) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (gk + e >= K) v[e] = 0;
                    }
                }
                ra[i] = v;
            } else {
                const int krow = c / (BM / 8), c8 = c % (BM / 8);
                const int gk = k0 + krow;
                int gm = m0 + 8 * c8; gm = gm + 8 <= p.lda ? gm : 0;  // keep the 16-B load inside the row
                s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (gk < K) v = *reinterpret_cast<const s16x8_t*>(A + (int64_t)gk * p.lda + gm);
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int c = tid + 256 * i;
            if constexpr (B_KC) {
                const int row = c >> 2, t = c & 3;
                int gn = n0 + row; gn = gn < N ? gn : N - 1;
                const int gk = k0 + 8 * t;
                s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (gk < K) {
                    v = *reinterpret_cast<const s16x8_t*>(B + (int64_t)gn * p.ldb + gk);
                    if (gk + 8 > K) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (gk + e >= K) v[e] = 0;
                    }
                }
                rb[i] = v;
            } else {
                const int krow = c / (BN / 8), c8 = c % (BN / 8);
                const int gk = k0 + krow;
                int gn = n0 + 8 * c8; gn = gn + 8 <= p.ldb ? gn : 0;
                s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (gk < K) v = *reinterpret_cast<const s16x8_t*>(B + (int64_t)gk * p.ldb + gn);
                rb[i] = v;
            }
        }
    };
    auto store_tiles = [&](int buf) {
        unsigned char* sa = smem + buf * (A_BYTES + B_BYTES);
        unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int c = tid + 256 * i;
            if constexpr (A_KC) {
                const int row = c >> 2, t = c & 3;
                *reinterpret_cast<s16x8_t*>(sa + row * 64 + ((t ^ ((row >> 2) & 3)) << 4)) = ra[i];
            } else {
                const int krow = c / (BM / 8), c8 = c % (BM / 8);
                *reinterpret_cast<s16x8_t*>(sa + krow * RSA + c8 * 16) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int c = tid + 256 * i;
            if constexpr (B_KC) {
                const int row = c >> 2, t = c & 3;
                *reinterpret_cast<s16x8_t*>(sb + row * 64 + ((t ^ ((row >> 2) & 3)) << 4)) = rb[i];
            } else {
                const int krow = c / (BN / 8), c8 = c % (BN / 8);
                *reinterpret_cast<s16x8_t*>(sb + krow * RSB + c8 * 16) = rb[i];
            }
        }
    };

    // ---- fragment read (8 bf16 per lane) ----
    const int r16 = lane & 15, g = lane >> 4;
    auto read_frag = [&](const unsigned char* tile, bool kcontig, int base_row, int rs) -> bf16x8_t {
        s16x4_t lo, hi;
        if (kcontig) {
            const int row = base_row + r16;
            const int sw = (row >> 2) & 3;
            const unsigned char* rp = tile + row * 64 + (g & 1) * 8;
            lo = *reinterpret_cast<const s16x4_t*>(rp + ((((g >> 1)) ^ sw) << 4));
            hi = *reinterpret_cast<const s16x4_t*>(rp + (((2 + (g >> 1)) ^ sw) << 4));
        } else {
            const int q = r16 >> 2, pp = r16 & 3;
            const unsigned char* rp = tile + (4 * g + q) * rs + (base_row + 4 * pp) * 2;
            lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(rp));
            hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(rp + 16 * rs));
        }
        s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        if (t + 1 < nk) load_tiles((t + 1) * BK);
        const unsigned char* sa = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* sb = sa + A_BYTES;
        bf16x8_t af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = read_frag(sa, A_KC, wm0 + 16 * i, RSA);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[j] = read_frag(sb, B_KC, wn0 + 16 * j, RSB);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        if (t + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n..n+3], m = ..+(lane&15), n = ..+4*(lane>>4) ----
    const int Np4 = (N + 3) & ~3;
    const int64_t coff = zo * p.sC0 + zi * p.sC1;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm0 + 16 * i + r16;
        if (m >= M) continue;
        int orow = m;
        int brow = 0;
        if (p.rpg > 0) {
            brow = (m % p.rpg) + p.roff;
            orow = (m / p.rpg) * p.gstride + brow;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn0 + 16 * j + 4 * g;
            if (n >= Np4) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * p.alpha;
            if (p.bias) {
                const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
            }
            if (p.rowbias) {
                const float4 b4 = *reinterpret_cast<const float4*>(p.rowbias + (int64_t)brow * N + n);
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
            }
            const int64_t o = coff + (int64_t)orow * p.ldc + n;
            if (p.residual) {
                const float4 r4 = *reinterpret_cast<const float4*>(p.residual + o);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if (p.actin) {  // dgrad through the activation: dF = dG * act'(F)
                const ushort4 f4 = *reinterpret_cast<const ushort4*>(p.actin + (int64_t)m * p.ldact + n);
                v[0] *= act_bwd(bf2f(f4.x), p.act); v[1] *= act_bwd(bf2f(f4.y), p.act);
                v[2] *= act_bwd(bf2f(f4.z), p.act); v[3] *= act_bwd(bf2f(f4.w), p.act);
            }
            if (p.out_f32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + o) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                ushort4 o4 = make_ushort4(f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3]));
                *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(p.C) + o) = o4;
                if (p.C2) {  // second output: activation of the ROUNDED pre-activation (what backward re-reads)
                    ushort4 a4 = make_ushort4(f2bf(act_fwd(bf2f(o4.x), p.act)), f2bf(act_fwd(bf2f(o4.y), p.act)),
                                              f2bf(act_fwd(bf2f(o4.z), p.act)), f2bf(act_fwd(bf2f(o4.w), p.act)));
                    *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(p.C2) + o) = a4;
                }
            }
        }
    }
}

template <int LAYOUT>
static int launch_gemm(const GemmArgs& a, int batch, int tile, hipStream_t s) {
    if (tile == 128) {
        dim3 grid((a.N + 127) / 128, (a.M + 127) / 128, batch);
        hipLaunchKernelGGL((gemm_bf16_kernel<LAYOUT, 128, 128>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid((a.N + 63) / 64, (a.M + 63) / 64, batch);
        hipLaunchKernelGGL((gemm_bf16_kernel<LAYOUT, 64, 64>), grid, dim3(256), 0, s, a);
    }
    return check_launch("swat_gemm_bf16");
}

// ---- optional per-launch timing (bench.py's roofline leg): HIP events on the launch stream around every GEMM ----
struct GemmProf {
    bool on = false;
    int cap = 0, n = 0;
    hipEvent_t* ev = nullptr;  // 2 per launch
    int* kind = nullptr;       // layout * 2 + (tile == 128)
    double* flops = nullptr;
};
static GemmProf g_prof;

int gemm_dispatch(int layout, const GemmArgs& a, int batch, hipStream_t s) {
    // tile choice: 128x128 only when it still yields >= ~1.5 waves of blocks over 256 CUs
    const int64_t big = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) * batch;
    const int tile = (big >= 384 && a.M >= 128 && a.N >= 128) ? 128 : 64;
    const bool prof = g_prof.on && g_prof.n < g_prof.cap;
    if (prof) hipEventRecord(g_prof.ev[2 * g_prof.n], s);
    int rc;
    if (layout == L_NT) rc = launch_gemm<L_NT>(a, batch, tile, s);
    else if (layout == L_NN) rc = launch_gemm<L_NN>(a, batch, tile, s);
    else rc = launch_gemm<L_TN>(a, batch, tile, s);
    if (prof) {
        hipEventRecord(g_prof.ev[2 * g_prof.n + 1], s);
        g_prof.kind[g_prof.n] = layout * 2 + (tile == 128);
        g_prof.flops[g_prof.n] = 2.0 * a.M * a.N * a.K * batch;
        ++g_prof.n;
    }
    return rc;
}

}  // namespace swat

// Start timing every GEMM launch (up to max_launches) with hipEvents on its own stream; not graph-capturable.
SWAT_API int swat_gemm_prof_begin(int max_launches) {
    using swat::g_prof;
    if (g_prof.ev) {
        for (int i = 0; i < 2 * g_prof.cap; ++i) hipEventDestroy(g_prof.ev[i]);
        delete[] g_prof.ev; delete[] g_prof.kind; delete[] g_prof.flops;
    }
    g_prof.cap = max_launches; g_prof.n = 0;
    g_prof.ev = new hipEvent_t[2 * max_launches];
    g_prof.kind = new int[max_launches];
    g_prof.flops = new double[max_launches];
    for (int i = 0; i < 2 * max_launches; ++i) hipEventCreate(&g_prof.ev[i]);
    g_prof.on = true;
    return 0;
}

// Stop, synchronise the events and report per kind (layout*2 + is128; 6 kinds): total ms, launches, flops launched.
SWAT_API int swat_gemm_prof_end(double* ms_by_kind, int64_t* count_by_kind, double* flops_by_kind) {
    using swat::g_prof;
    g_prof.on = false;
    for (int k = 0; k < 6; ++k) { ms_by_kind[k] = 0; count_by_kind[k] = 0; flops_by_kind[k] = 0; }
    for (int i = 0; i < g_prof.n; ++i) {
        hipEventSynchronize(g_prof.ev[2 * i + 1]);
        float ms = 0.f;
        hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        ms_by_kind[g_prof.kind[i]] += ms;
        count_by_kind[g_prof.kind[i]] += 1;
        flops_by_kind[g_prof.kind[i]] += g_prof.flops[i];
    }
    const int n = g_prof.n;
    g_prof.n = 0;
    return n;
}

// Generic entry (tests, attention as batched GEMMs).  See include/swat_hip.h.
SWAT_API int swat_gemm_bf16(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb,
                            int ldc, int batch, int inner, int64_t sA0, int64_t sA1, int64_t sB0, int64_t sB1,
                            int64_t sC0, int64_t sC1, const float* bias, const float* residual, const void* actin,
                            int ldact, int act, void* C2, int out_f32, float alpha, swat_stream_t stream) {
    SWAT_REQUIRE(layout >= 0 && layout <= 2, "swat_gemm_bf16: bad layout %d", layout);
    SWAT_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && inner > 0, "swat_gemm_bf16: bad shape");
    SWAT_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "swat_gemm_bf16: lda/ldb must be multiples of 8, ldc of 4");
    SWAT_REQUIRE(ldc >= ((N + 3) & ~3), "swat_gemm_bf16: ldc must cover N rounded up to 4");
    swat::GemmArgs a{};
    a.A = (const unsigned short*)A; a.B = (const unsigned short*)B; a.C = C; a.C2 = C2;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.inner = inner; a.sA0 = sA0; a.sA1 = sA1; a.sB0 = sB0; a.sB1 = sB1; a.sC0 = sC0; a.sC1 = sC1;
    a.bias = bias; a.residual = residual; a.actin = (const unsigned short*)actin; a.ldact = ldact; a.act = act;
    a.out_f32 = out_f32; a.alpha = alpha; a.rpg = 0; a.gstride = 0; a.roff = 0; a.rowbias = nullptr;
    return swat::gemm_dispatch(layout, a, batch, (hipStream_t)stream);
}
