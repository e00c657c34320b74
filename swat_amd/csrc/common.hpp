// swat_amd/csrc/common.hpp — shared helpers for the gfx950 kernels behind include/swat_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define SWAT_API extern "C" __attribute__((visibility("default")))

namespace swat {

// thread-local last-error text (swat_last_error)
void set_error(const char* fmt, ...);
int check_launch(const char* what);  // hipGetLastError -> 0 / negative code

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// bf16 <-> f32 on raw bits (round-to-nearest-even; NaN kept NaN)
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // quiet NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

}  // namespace swat

#define SWAT_REQUIRE(cond, ...)                       \
    do {                                              \
        if (!(cond)) {                                \
            swat::set_error(__VA_ARGS__);             \
            return -1;                                \
        }                                             \
    } while (0)
