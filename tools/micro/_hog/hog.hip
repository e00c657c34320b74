// tools/micro/_hog/hog.hip — a kernel that only OCCUPIES: 12 waves per workgroup (3 per SIMD), 144 KB of LDS, a chosen VGPR allocation,
// spinning on the clock for `us` microseconds.  Beside it (another stream) tools/micro/adamw_beside.py times swat_adamw: how fast does
// the optimizer stream when a weight-gradient launch (csrc/gemm_xl.hip: 3 x 152 registers per SIMD) leaves it 56 registers per SIMD
// (one AdamW wave), and how fast when it left 128 (two)?
#include <hip/hip_runtime.h>
template <int R>
__global__ __launch_bounds__(768) void hog_kernel(unsigned long long ticks, unsigned* sink) {
    __shared__ unsigned char smem[144 * 1024];
    if (R == 152) asm volatile("v_mov_b32 v151, 0" ::: "v151");
    if (R == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (R == 104) asm volatile("v_mov_b32 v103, 0" ::: "v103");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks == 1) sink[0] = smem[threadIdx.x];
}
extern "C" int launch_hog(int regs, int us, int blocks, void* stream) {
    const unsigned long long ticks = (unsigned long long)us * 100ull;
    if (regs == 152) hipLaunchKernelGGL(hog_kernel<152>, dim3(blocks), dim3(768), 0, (hipStream_t)stream, ticks, nullptr);
    else if (regs == 128) hipLaunchKernelGGL(hog_kernel<128>, dim3(blocks), dim3(768), 0, (hipStream_t)stream, ticks, nullptr);
    else hipLaunchKernelGGL(hog_kernel<104>, dim3(blocks), dim3(768), 0, (hipStream_t)stream, ticks, nullptr);
    return (int)hipGetLastError();
}
