// v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 in bare loops (operands in registers, one wave per SIMD, every
// CU busy): FLOP/s on zero and on random operand bits, with 4 and 8 INDEPENDENT accumulator sets per wave, and the clock the
// chip held inside the kernel (s_memtime / s_memrealtime).  Settles DESIGN.md section 9 item 5 (round 3 left a note that
// recommended the 16x16x32 shape from MI355X_MICROARCH.md's "DVFS give-back (7)" while an uncommitted scratch probe of the
// same evening had measured it at 0.58-0.72x -- this is that probe, written so that neither shape is issue- or
// dependency-bound: >= 4 accumulator sets per wave, equal FLOPs per loop trip, no VALU / memory instruction in the loop.
// The MFMAs are inline asm with the accumulators pinned in VGPRs: with the builtin, hipcc kept the 16x16 tiles (four
// registers each) partly in AGPRs and copied them in and out with v_accvgpr_read/write inside the loop -- THAT was the
// 0.58x of the scratch probe, a register-allocation artefact, not the instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cyc, real; };

// NACC accumulator tiles of 32x32 (16 registers each); one trip = NACC MFMAs = NACC * 32768 FLOP per wave
template <int NACC>
__global__ __launch_bounds__(256) void loop32(const uint4* __restrict__ ops, float* out, Stamp* st, int iters) {
    const uint4 ua = ops[threadIdx.x], ub = ops[256 + threadIdx.x];
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(&ua), b = *reinterpret_cast<const bf16x8*>(&ub);
    f32x16 c[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) c[n][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[n]) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) s += c[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t1 - t0, r1 - r0};
}

// 2 * NACC accumulator tiles of 16x16 (4 registers each): one trip = 2 NACC MFMAs of 16384 FLOP = the same FLOPs per wave
template <int NACC>
__global__ __launch_bounds__(256) void loop16(const uint4* __restrict__ ops, float* out, Stamp* st, int iters) {
    const uint4 ua = ops[threadIdx.x], ub = ops[256 + threadIdx.x];
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(&ua), b = *reinterpret_cast<const bf16x8*>(&ub);
    f32x4 c[2 * NACC];
    for (int n = 0; n < 2 * NACC; ++n)
        for (int r = 0; r < 4; ++r) c[n][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 2 * NACC; ++n) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[n]) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int n = 0; n < 2 * NACC; ++n)
        for (int r = 0; r < 4; ++r) s += c[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t1 - t0, r1 - r0};
}

static int cmp_d(const void* x, const void* y) { const double a = *(const double*)x, b = *(const double*)y; return (a > b) - (a < b); }

template <typename K>
static void run(const char* name, K kern, int nacc, const uint4* ops, float* out, Stamp* st, const char* data) {
    const int grid = 256, iters = 400000;           // one workgroup of 4 waves per CU = one wave per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {             // ~100 ms each: long enough for the clock to settle
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, ops, out, st, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    Stamp h[256];
    hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
    double ghz[256];
    for (int i = 0; i < grid; ++i) ghz[i] = (double)h[i].cyc / ((double)h[i].real * 10.0);     // realtime ticks at 100 MHz
    qsort(ghz, grid, sizeof(double), cmp_d);
    const double flops = (double)iters * nacc * 32768.0 * grid * 4;
    const double cyc_per_32k = (double)h[0].cyc / ((double)iters * nacc);
    printf("%-6s operands, %-9s %d accumulator sets: %8.2f ms  %.3f PFLOP/s  in-kernel clock %.2f GHz (median), %.1f cycles per 32768 FLOP per SIMD\n",
           data, name, nacc, ms, flops / ms / 1e12, ghz[grid / 2], cyc_per_32k);
}

int main() {
    uint4 h[512];
    uint4* ops[2];
    float* out;
    Stamp* st;
    hipMalloc(&out, sizeof(float) * 256 * 256);
    hipMalloc(&st, sizeof(Stamp) * 256);
    srand(12345);
    for (int d = 0; d < 2; ++d) {
        for (int i = 0; i < 512; ++i) {
            uint32_t w[4];
            for (int k = 0; k < 4; ++k) {
                // random bf16 pairs in [1, 2) with random mantissas and signs: data-like switching activity, no inf / nan
                const uint32_t lo = 0x3F80u | (rand() & 0x7F) | ((rand() & 1) << 15), hi = 0x3F80u | (rand() & 0x7F) | ((rand() & 1) << 15);
                w[k] = d ? (lo | (hi << 16)) : 0u;
            }
            h[i] = uint4{w[0], w[1], w[2], w[3]};
        }
        hipMalloc(&ops[d], sizeof h);
        hipMemcpy(ops[d], h, sizeof h, hipMemcpyHostToDevice);
    }
    for (int round = 0; round < 2; ++round)
        for (int d = 0; d < 2; ++d) {
            const char* data = d ? "random" : "zero";
            run("32x32x16", loop32<4>, 4, ops[d], out, st, data);
            run("16x16x32", loop16<4>, 4, ops[d], out, st, data);
            run("32x32x16", loop32<8>, 8, ops[d], out, st, data);
            run("16x16x32", loop16<8>, 8, ops[d], out, st, data);
        }
    return 0;
}
