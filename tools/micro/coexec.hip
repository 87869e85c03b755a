// Do a VALU-heavy wave and an MFMA-heavy wave that share a SIMD run side by side?  512-thread workgroups on every CU:
// waves 0-3 run V independent v_and/v_sub/v_perm-like VALU instructions per iteration, waves 4-7 run M MFMAs per
// iteration (data-like operands).  Times: VALU waves alone, MFMA waves alone, both.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ROLE>   // 1 = VALU waves only, 2 = MFMA waves only, 3 = both; 4 = both roles interleaved in EVERY wave (4 waves)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int t = threadIdx.x, wave = t >> 6;
    const bool valu_wave = wave < 4;
    float r = 0.f;
    if (ROLE == 4) {
        if (wave >= 4) return;
        unsigned x[16];
        for (int i = 0; i < 16; ++i) x[i] = t * 2654435761u + i * 40503u;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + 0.01f * ((t * 7 + i) & 63)); b[i] = (__bf16)(1.5f - 0.01f * ((t * 3 + i) & 31)); }
        f32x16 c0, c1, c2, c3;
        for (int q = 0; q < 16; ++q) { c0[q] = 0.f; c1[q] = 0.f; c2[q] = 0.f; c3[q] = 0.f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_and_b32 %0, 0xffff0000, %0\n\tv_add_u32 %0, %0, %1" : "+v"(x[(u * 4 + i) & 15]) : "v"(x[(u * 4 + i + 1) & 15]));
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_and_b32 %0, 0xffff0000, %0\n\tv_add_u32 %0, %0, %1" : "+v"(x[(u * 4 + i + 4) & 15]) : "v"(x[(u * 4 + i + 5) & 15]));
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        }
        for (int q = 0; q < 16; ++q) r += c0[q] + c1[q] + c2[q] + c3[q];
        for (int i = 0; i < 16; ++i) r += (float)x[i];
        out[blockIdx.x * 512 + t] = r;
        return;
    }
    if (valu_wave) {
        if (!(ROLE & 1)) return;
        unsigned x[16];
        for (int i = 0; i < 16; ++i) x[i] = t * 2654435761u + i * 40503u;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i)       // 48 x 2 = 96 VALU instructions per iteration (the MFMA waves do 24 MFMAs)
                    asm volatile("v_and_b32 %0, 0xffff0000, %0\n\tv_add_u32 %0, %0, %1" : "+v"(x[(u * 8 + i) & 15]) : "v"(x[(u * 8 + i + 1) & 15]));
        }
        for (int i = 0; i < 16; ++i) r += (float)x[i];
    } else {
        if (!(ROLE & 2)) return;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + 0.01f * ((t * 7 + i) & 63)); b[i] = (__bf16)(1.5f - 0.01f * ((t * 3 + i) & 31)); }
        f32x16 c0, c1, c2, c3;
        for (int q = 0; q < 16; ++q) { c0[q] = 0.f; c1[q] = 0.f; c2[q] = 0.f; c3[q] = 0.f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        }
        for (int q = 0; q < 16; ++q) r += c0[q] + c1[q] + c2[q] + c3[q];
    }
    out[blockIdx.x * 512 + t] = r;
}

template <int ROLE>
static float run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<ROLE>, dim3(256), dim3(512), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 512);
    const int iters = 20000;
    const float v = run<1>(out, iters), m = run<2>(out, iters), both = run<3>(out, iters), inter = run<4>(out, iters);
    printf("per iteration and SIMD: 96 VALU instructions (waves 0-3) | 24 MFMAs (waves 4-7)\n");
    printf("VALU waves alone %.3f ms   MFMA waves alone %.3f ms   both, wave-specialised %.3f ms   (sum %.3f, max %.3f)\n", v, m, both, v + m, v > m ? v : m);
    printf("one wave per SIMD doing 24 MFMAs + 96 VALU interleaved: %.3f ms\n", inter);
    return 0;
}
