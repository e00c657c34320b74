// tools/micro/store_rate.hip — cost of a GEMM epilogue's store burst (tuning aid): 1200 workgroups each write a 64 x 64
// bf16 tile of two [1600, 3072] outputs (19.7 MB in all), either in MFMA accumulator order (lane = row, 8 B of 4 n:
// 16 rows x 32-B segments per store instruction) or row-coalesced (8 lanes x 16 B = one 128-B line per row).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void store_tiles(unsigned short* __restrict__ C, unsigned short* __restrict__ C2, int N, int spin) {
    const int tiles_n = N / 64;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // optional delay that differs per block: de-synchronises the bursts (spin > 0)
    if (spin) { const long long t0 = clock64(); const long long w = (long long)spin * (blockIdx.x % 7); while (clock64() - t0 < w) {} }
    if (MODE == 0) {
        const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = tm * 64 + wm * 32 + i * 16 + (lane & 15);
                const int col = tn * 64 + wn * 32 + j * 16 + (lane >> 4) * 4;
                const u32x2 v = {(unsigned)row, (unsigned)col};
                *reinterpret_cast<u32x2*>(C + (size_t)row * N + col) = v;
                *reinterpret_cast<u32x2*>(C2 + (size_t)row * N + col) = v;
            }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = tm * 64 + wave * 16 + i * 8 + (lane >> 3);
            const int col = tn * 64 + (lane & 7) * 8;
            const u32x4 v = {(unsigned)row, (unsigned)col, 0u, 1u};
            *reinterpret_cast<u32x4*>(C + (size_t)row * N + col) = v;
            *reinterpret_cast<u32x4*>(C2 + (size_t)row * N + col) = v;
        }
    }
}

int main(int argc, char** argv) {
    const int M = 1600, N = 3072;
    unsigned short *C, *C2;
    hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&C2, (size_t)M * N * 2);
    const int grid = (M / 64) * (N / 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin = 0; spin <= 2000; spin += 2000)
        for (int mode = 0; mode < 2; ++mode) {
            float tot = 0;
            for (int it = 0; it < 12; ++it) {
                hipEventRecord(e0);
                for (int r = 0; r < 10; ++r) {
                    if (mode == 0) hipLaunchKernelGGL(store_tiles<0>, dim3(grid), dim3(256), 0, 0, C, C2, N, spin);
                    else hipLaunchKernelGGL(store_tiles<1>, dim3(grid), dim3(256), 0, 0, C, C2, N, spin);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) tot += ms;
            }
            printf("spin %4d mode %d (%s): %.2f us per launch (10 back-to-back), %.2f TB/s\n", spin, mode,
                   mode ? "row-coalesced 128-B lines" : "accumulator order 32-B segments", tot / 10 / 10 * 1e3,
                   2.0 * M * N * 2 / (tot / 10 / 10 * 1e-3) / 1e12);
        }
    return 0;
}
