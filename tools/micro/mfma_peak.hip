// Sustained rate of v_mfma_f32_32x32x16_bf16 with nothing else running: every CU, 1 or 2 waves per SIMD, four
// independent accumulators per wave, no memory traffic.  Gives the ACHIEVABLE matrix peak of this part under its power
// limit (the nominal 2.5 PFLOP/s assumes 2.4 GHz), against which the bf16 kernels' executed rates are to be read.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void spin(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.0f + i); }
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 1024 * 512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfgs[4][2] = {{256, 256}, {256, 512}, {512, 512}, {3, 256}};
    for (int c = 0; c < 4; ++c) {
        const int grid = cfgs[c][0], block = cfgs[c][1];
        for (int iters = 2000; iters <= 200000; iters *= 10) {
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(spin, dim3(grid), dim3(block), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double nm = 24.0 * iters * grid * (block / 64);
            const double flops = nm * 32.0 * 32 * 16 * 2;
            const int wps = (grid * (block / 64) + 1023) / 1024;   // waves per SIMD (grid <= 2 x CUs)
            printf("grid %3d x %3d threads, %6d iterations: %9.3f ms  %.3f PFLOP/s  (%.1f ns per MFMA per SIMD at ~%d wave(s)/SIMD)\n",
                   grid, block, iters, ms, flops / ms / 1e12, ms * 1e6 / (24.0 * iters * (wps < 1 ? 1 : wps)), wps < 1 ? 1 : wps);
        }
    }
    return 0;
}
