// Shared host/device helpers for the kccot HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/kccot.h"

#define KCCOT_WAVE 64

namespace kccot {

void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) {
    char buf[384];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- wave-level reductions (64-wide wavefront; xor butterflies leave the result in every lane)
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (vmcnt(0)), which would put every outstanding global store / prefetch load on the critical
// path of a loop that synchronises through LDS alone; LDS visibility needs lgkmcnt(0) + s_barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// ---- DPP reductions inside aligned groups of LPR <= 16 lanes; every lane gets the result -----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// max(v, dpp(v)) in ONE instruction.  fmaxf() on a DPP move costs mov + canonicalise + max; the
// values here are never NaN-signalling, so the bare v_max_f32_dpp is exact.  The s_nop covers the
// VALU-write -> DPP-read hazard (2 wait states) that hipcc does not pad inside an asm statement.
#define KCCOT_DPP_MAX(V, CTRL)                                                                          \
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(V) : "v"(V))
template <int LPR>
__device__ __forceinline__ float seg_max(float v) {
    if (LPR >= 2) KCCOT_DPP_MAX(v, "quad_perm:[1,0,3,2]");
    if (LPR >= 4) KCCOT_DPP_MAX(v, "quad_perm:[2,3,0,1]");
    if (LPR >= 8) KCCOT_DPP_MAX(v, "row_half_mirror");
    if (LPR >= 16) KCCOT_DPP_MAX(v, "row_mirror");
    return v;
}
template <int LPR>
__device__ __forceinline__ float seg_sum(float v) {
    if (LPR >= 2) v += dpp_mov<0xB1>(v);
    if (LPR >= 4) v += dpp_mov<0x4E>(v);
    if (LPR >= 8) v += dpp_mov<0x141>(v);
    if (LPR >= 16) v += dpp_mov<0x140>(v);
    return v;
}

// DPP forms for latency-critical loops: four in-row steps (every lane of a 16-lane row ends with its row's
// result), then the four row results are read through SGPRs.  ~11 instructions and no LDS-crossbar round
// trips (a __shfl_xor butterfly is six dependent ds_bpermute_b32).  The result is wave-uniform.
__device__ __forceinline__ float wave_row_dpp_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
    return v;
}
__device__ __forceinline__ float wave_sum_fast(float v) {
    v = wave_row_dpp_sum(v);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (a + b) + (c + d);
}
__device__ __forceinline__ float wave_max_fast(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)));
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum with a fixed (deterministic) combination order.  `scratch` holds >= 16 floats
// of LDS.  Every thread of the block must call it; every thread receives the result.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < nw; ++w) r += scratch[w];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float r = scratch[0];
    for (int w = 1; w < nw; ++w) r = fmaxf(r, scratch[w]);
    return r;
}

}  // namespace kccot
