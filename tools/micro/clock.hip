// Effective shader clock seen by a tiny launch (3 workgroups, like the Sinkhorn kernels): N independent full-rate
// VALU instructions per wave, one wave per SIMD -> time = N * 4 cycles / f.  Also the quarter-rate v_exp_f32 and a
// dependent chain, to price the solver's half-step in real cycles.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/clock.hip -o /tmp/clock && /tmp/clock
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(1024) void spin(float* out, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(1.0f));            // 16 independent adds
            if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                              // 16 independent exps
            if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(1.0f));            // dependent chain
            if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[i & 14]) : "v"(*(double*)&a[(i + 2) & 14]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* what, int grid, int block, int iters) {
    float* out;
    hipMalloc(&out, sizeof(float) * grid * block);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin<MODE>, dim3(grid), dim3(block), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = 16.0 * iters;
        const int waves_per_simd = (block / 64 + 3) / 4;
        if (rep == 2)
            printf("%-34s grid %4d x %4d: %8.1f us, %.2f ns per instruction slot per SIMD (waves/SIMD %d) -> f = %.2f GHz if %d cycles each\n",
                   what, grid, block, ms * 1e3, ms * 1e6 / (instr_per_wave * waves_per_simd), waves_per_simd,
                   (MODE == 1 ? 16.0 : 4.0) / (ms * 1e6 / (instr_per_wave * waves_per_simd)), MODE == 1 ? 16 : 4);
    }
    hipFree(out);
}

int main() {
    const int it = 20000;
    run<0>("independent v_add_f32", 3, 256, it);
    run<0>("independent v_add_f32", 3, 512, it);
    run<0>("independent v_add_f32", 256, 512, it);
    run<3>("independent v_pk_add_f32", 3, 256, it);
    run<1>("independent v_exp_f32", 3, 256, it);
    run<1>("independent v_exp_f32", 3, 512, it);
    run<1>("independent v_exp_f32", 256, 512, it);
    run<2>("dependent v_add_f32 chain", 3, 256, it);
    run<2>("dependent v_add_f32 chain", 3, 512, it);
    return 0;
}
