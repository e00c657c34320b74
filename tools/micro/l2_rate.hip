// tools/micro/l2_rate.hip — how fast can a CU pull GEMM operand panels out of L2?  (tuning aid, not part of the library)
//   mode 0: LDS-DMA (global_load_lds_dwordx4) of 64-row x 64-k bf16 pieces, 3-stage ring, 256 threads (what gemm_dma does)
//   mode 1: global_load_dwordx4 straight into registers in MFMA-operand order (lane = row, 16 B of k), no LDS
//   mode 2: mode 1 for the A panel + LDS-DMA for the B panel (the hybrid)
// Tile walk = a 1600 x 3072 x 768 NT GEMM with 64 x 64 tiles (1200 workgroups, 12 k-steps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE, int SYNC>
__global__ __launch_bounds__(256) void pull(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, int M,
                                            int N, int K, unsigned* sink, int reps) {
    extern __shared__ char lds[];
    const int tiles_n = N / 64;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (int k0 = 0; k0 < K; k0 += 64) {
            if (MODE == 0 || MODE == 2) {
                // each wave moves 16 rows x 128 B of each panel it stages: 2 pieces of 1 KB (8 rows x 128 B)
                char* dst = lds + ((k0 / 64) % 3) * 16384;
                for (int p = 0; p < 2; ++p) {
                    const int row = wave * 16 + p * 8 + (lane >> 3);
                    const int kb = (lane & 7) * 8;
                    if (MODE == 0) {
                        int ra = tm * 64 + row; if (ra >= M) ra = M - 1;
                        dma16(A + (size_t)ra * K + k0 + kb, dst + (wave * 2 + p) * 1024);
                    }
                    dma16(B + (size_t)(tn * 64 + row) * K + k0 + kb, dst + 8192 + (wave * 2 + p) * 1024);
                }
            }
            if (MODE == 1 || MODE == 2) {
                // v_mfma_f32_16x16x32_bf16 operand order: lane -> row (lane & 15), k elements [8*(lane>>4), +8): one load
                // instruction covers 16 rows x 64 B; a 32-row x 64-k slab = 4 loads
                u32x4 v[4], w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int ra = tm * 64 + wm * 32 + (q >> 1) * 16 + (lane & 15); if (ra >= M) ra = M - 1;
                    v[q] = *reinterpret_cast<const u32x4*>(A + (size_t)ra * K + k0 + (q & 1) * 32 + (lane >> 4) * 8);
                    if (MODE == 1)
                        w[q] = *reinterpret_cast<const u32x4*>(B + (size_t)(tn * 64 + wn * 32 + (q >> 1) * 16 + (lane & 15)) * K + k0 + (q & 1) * 32 + (lane >> 4) * 8);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc ^= v[q]; if (MODE == 1) acc ^= w[q]; }
            }
            if (MODE != 1 && SYNC == 0 && (k0 / 64) % 3 == 2) {
                __builtin_amdgcn_s_waitcnt(0);  // drain the ring every third stage (bounded LDS-DMA queue)
                __syncthreads();
            }
            if (MODE != 1 && SYNC == 1) {
                // what gemm_dma_kernel does: the stage issued two steps ago has landed (this step's and the previous
                // step's 4 pieces may still fly), then one workgroup barrier per k-step
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    if (MODE != 1) { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); acc.x ^= ((unsigned*)lds)[threadIdx.x]; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}


// generic tile walk: BM x BN tile (rows multiple of 8), NST stages, 4 waves, pieces dealt round-robin, full drain per step
// replaced by a conservative counted wait (pieces per wave differ by at most one)
template <int BM, int BN, int NST, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void pull_tile(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, int M,
                                                 int N, int K, unsigned* sink, int reps, int ld) {
    extern __shared__ char lds[];
    constexpr int PA = BM / 8, PB = BN / 8, P = PA + PB;      // 1 KB pieces per stage
    constexpr int STAGE = P * 1024;
    constexpr int MAXP = (P + NWV - 1) / NWV;
    const int tiles_n = N / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        const int nk = K / 64;
        for (int t = -(NST - 1); t < nk; ++t) {
            const int ti = t + NST - 1;   // stage to issue
            if (ti < nk) {
                char* dst = lds + (ti % NST) * STAGE;
#pragma unroll
                for (int i = 0; i < MAXP; ++i) {
                    const int pc = wave + NWV * i;
                    if (pc < P) {
                        const bool isA = pc < PA;
                        const int row = (isA ? pc : pc - PA) * 8 + (lane >> 3);
                        const int kb = (lane & 7) * 8;
                        int gr = isA ? tm * BM + row : tn * BN + row;
                        const int R = isA ? M : N;
                        if (gr >= R) gr = R - 1;
                        dma16((isA ? A : B) + (size_t)gr * ld + ti * 64 + kb, dst + pc * 1024);
                    }
                }
            }
            if (t >= 0) {
                // stage t must have landed: allow the (NST-1) younger stages' pieces of this wave to fly (upper bound MAXP each
                // would be unsafe; use the lower bound of pieces per wave so the wait is conservative)
                if (t + NST - 1 < nk) { if (NST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (P / NWV)) : "memory");
                                        else if (NST == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (P / NWV)) : "memory");
                                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P / NWV)) : "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    acc ^= ((unsigned*)lds)[threadIdx.x];
    if (acc == 0x12345678u) sink[0] = 1;
}
template <int BM, int BN, int NST, int NWV = 4>
static void bench_tile(const char* name, const unsigned short* A, const unsigned short* B, int M, int N, int K, unsigned* sink, int reps, int ld = 0) {
    if (!ld) ld = K;
    constexpr int STAGE = (BM / 8 + BN / 8) * 1024;
    const int grid = ((M + BM - 1) / BM) * (N / BN);
    hipFuncSetAttribute((const void*)pull_tile<BM, BN, NST, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((pull_tile<BM, BN, NST, NWV>), dim3(grid), dim3(64 * NWV), NST * STAGE, 0, A, B, M, N, K, sink, reps, ld);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)grid * (K / 64) * reps * STAGE;
    printf("%-28s grid %4d lds %3d KB: %6.1f us per pass, %5.1f MB per pass, %.2f TB/s\n", name, grid, NST * STAGE / 1024, best * 1e3 / reps,
           bytes / reps / 1e6, bytes / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 8;
    const int M = 1600, N = 3072, K = 768;
    unsigned short *A, *B; unsigned* sink;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&sink, 4);
    hipMemset(A, 1, (size_t)M * K * 2); hipMemset(B, 2, (size_t)N * K * 2);
    const int grid = ((M + 63) / 64) * (N / 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL((pull<0, 0>), dim3(grid), dim3(256), 49152, 0, A, B, M, N, K, sink, reps);
            if (mode == 1) hipLaunchKernelGGL((pull<1, 0>), dim3(grid), dim3(256), 0, 0, A, B, M, N, K, sink, reps);
            if (mode == 2) hipLaunchKernelGGL((pull<2, 0>), dim3(grid), dim3(256), 49152, 0, A, B, M, N, K, sink, reps);
            if (mode == 3) hipLaunchKernelGGL((pull<0, 1>), dim3(grid), dim3(256), 49152, 0, A, B, M, N, K, sink, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // bytes requested by workgroups: per tile and k-step 16 KB (mode 1: each of the 4 waves loads its own 32+32 rows = 32 KB)
        const double tile_bytes = (double)grid * (K / 64) * reps * (mode == 1 ? 32768.0 : (mode == 2 ? 8192.0 + 16384.0 : 16384.0));
        printf("mode %d: %.1f us per GEMM-equivalent pass, %.2f TB/s requested, %.1f GB/s per CU\n", mode, best * 1e3 / reps,
               tile_bytes / (best * 1e-3) / 1e12, tile_bytes / (best * 1e-3) / 1e9 / 256);
    }
    // deep-K products: K = 3072 with the natural row stride (6144 B) and with padded strides
    unsigned short *A2, *B2;
    hipMalloc(&A2, (size_t)1600 * 3328 * 2); hipMalloc(&B2, (size_t)3072 * 3328 * 2);
    hipMemset(A2, 1, (size_t)1600 * 3328 * 2); hipMemset(B2, 2, (size_t)3072 * 3328 * 2);
    bench_tile<64, 64, 3, 4>("N=768 K=3072 ld 3072", A2, B2, M, 768, 3072, sink, reps, 3072);
    bench_tile<64, 64, 3, 4>("N=768 K=3072 ld 3136 (+128 B)", A2, B2, M, 768, 3072, sink, reps, 3136);
    bench_tile<64, 64, 3, 4>("N=768 K=3072 ld 3104 (+64 B)", A2, B2, M, 768, 3072, sink, reps, 3104);
    bench_tile<64, 64, 3, 4>("N=768 K=3072 ld 3200 (+256 B)", A2, B2, M, 768, 3072, sink, reps, 3200);
    bench_tile<64, 64, 3, 4>("N=768 K=3072 ld 3328 (+512 B)", A2, B2, M, 768, 3072, sink, reps, 3328);
    bench_tile<64, 64, 3, 4>("N=3072 K=768 ld 768", A2, B2, M, 3072, 768, sink, reps, 768);
    bench_tile<64, 64, 3, 4>("N=3072 K=768 ld 832 (+128 B)", A2, B2, M, 3072, 768, sink, reps, 832);
    bench_tile<64, 64, 3, 4>("N=768 K=768 ld 768", A2, B2, M, 768, 768, sink, reps, 768);
    bench_tile<64, 64, 3, 4>("N=768 K=768 ld 832", A2, B2, M, 768, 768, sink, reps, 832);
    return 0;
}
