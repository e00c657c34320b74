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
// fp32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32)
typedef __bf16 swat_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float swat_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const swat_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, swat_bf16x2_t));
}
__device__ __forceinline__ unsigned short f2bf(float f) { return (unsigned short)(pack_bf16x2(f, 0.0f) & 0xffffu); }

}  // namespace swat

// a HIP runtime call inside a function that returns the ABI's int status
#define SWAT_HIP(call)                                                       \
    do {                                                                     \
        const hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) {                                             \
            swat::set_error("%s: %s", #call, hipGetErrorString(e__));        \
            return -1;                                                       \
        }                                                                    \
    } while (0)

#define SWAT_REQUIRE(cond, ...)                       \
    do {                                              \
        if (!(cond)) {                                \
            swat::set_error(__VA_ARGS__);             \
            return -1;                                \
        }                                             \
    } while (0)
