// tools/micro/mfma_rate.hip — ground truth for the consumer loop of gemm_pc.hip (tuning aid, not part of the library):
// cycles per bf16 MFMA on one SIMD with (a) nothing else, (b) LDS fragment reads interleaved, (c) scalar noise, for the
// 16x16x32 and 32x32x16 shapes, one or two waves per SIMD, every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int IMM>
__device__ __forceinline__ bf16x8 lds128(unsigned a) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(IMM));
    return v;
}

// MODE 0: 16x16x32, 16 accumulators, no reads.  MODE 1: + one ds_read_b128 per 2 MFMAs (results unused -> pinned).
// MODE 2: 32x32x16, 4 accumulators, no reads.   MODE 3: + one ds_read_b128 per MFMA.
// MODE 4: MODE 1 with counted waits + pins as in gemm_pc (fragments actually consumed)
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters, float* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0x3f803f80u + i;
    __syncthreads();
    const unsigned base = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem) +
                          ((lane & 15) * 128 + (((lane >> 4) ^ (((lane & 15) >> 1) & 7)) << 4));
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = lds128<0>(base + 2048 * i); b[i] = lds128<0>(base + 8192 + 2048 * i); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 acc[16]; f32x16 acc32[4];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[m & 3], a[m >> 2], acc[m], 0, 0, 0);
                if (MODE == 1 && (m & 1)) { bf16x8 t = lds128<0>(base + 2048 * (m >> 1)); asm volatile("" ::"v"(t)); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (MODE == 4) {
            // rows of 4 MFMAs; a[i] refreshed in place after its row, b refreshed behind rows 0/1 into the same regs (toy)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                asm volatile("" : "+v"(a[i]));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i * 4 + j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a[i] = lds128<0>(base + 2048 * i + (it & 1) * 16384);
                if (i == 3) { b[0] = lds128<0>(base + 8192); b[1] = lds128<2048>(base + 8192); b[2] = lds128<4096>(base + 8192); b[3] = lds128<6144>(base + 8192); }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                acc32[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[m & 3], a[m >> 1], acc32[m & 3], 0, 0, 0);
                if (MODE == 3) { bf16x8 t = lds128<0>(base + 2048 * m); asm volatile("" ::"v"(t)); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 4; ++i) s += acc32[i][0] + acc32[i][15];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, int per_iter_mfma, double flop_per_mfma) {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * sizeof(unsigned long long)); hipMalloc(&sink, 4);
    const int iters = 2000;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters, sink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters, sink); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * threads / 64); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (auto v : h) cyc += v; cyc /= h.size();
    const double per = cyc / iters / per_iter_mfma;
    const double tf = 256.0 * (threads / 64) * iters * per_iter_mfma * flop_per_mfma / (ms * 1e-3) / 1e12;
    printf("%-46s %d waves/CU: %6.1f cycles / MFMA per wave, %7.1f TFLOP/s chip, clock %.2f GHz\n", name, threads / 64, per, tf, cyc / (ms * 1e-3) / 1e9);
    hipFree(d); hipFree(sink);
}
int main() {
    for (int thr : {256, 512}) {
        run<0>("16x16x32 x16 acc, nothing else", thr, 16, 16384.0);
        run<1>("16x16x32 + 1 ds_read_b128 / 2 MFMA", thr, 16, 16384.0);
        run<4>("16x16x32 rows of 4, counted waits, in-place", thr, 16, 16384.0);
        run<2>("32x32x16 x4 acc, nothing else", thr, 8, 32768.0);
        run<3>("32x32x16 + 1 ds_read_b128 / MFMA", thr, 8, 32768.0);
    }
    return 0;
}
