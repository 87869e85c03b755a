// Micro-benchmark for the 256-row pair-tile Gram kernel's staging loads: 256-thread workgroups (one per CU) stream
// 512 rows x one K-chunk of an fp32 stack, 16 float4 loads in flight per thread, with row pieces of 64 / 128 / 256
// bytes per row and wave instruction (4 / 8 / 16 lanes per row).  SHARE workgroups read the same rows of the same
// K-chunk (the pairs that share a panel); they sit on one XCD (ids equal mod 8), so SHARE - 1 of them are served by L2.
// Question: does a 64-byte piece (a 16-k stage) cost L2 / TA efficiency against 128 bytes (32 k)?
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/rowpiece.hip -o gpurun_out/rowpiece
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int LPR>
__global__ __launch_bounds__(256) void stream(const float* __restrict__ s, int64_t K, int64_t chunk, int share, float* __restrict__ out) {
    constexpr int RPI = 256 / LPR;            // rows per block-wide load instruction
    constexpr int COLS = LPR * 4;             // floats of a row piece
    constexpr int NL = 16;                    // loads in flight per thread
    constexpr int ROWS_PER_GROUP = NL * RPI;  // 1024 / 512 / 256 rows per group of 16 loads
    const int t = threadIdx.x, lr = t / LPR, lc = (t % LPR) * 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int64_t kbeg = (int64_t)(xcd + 8 * (slot / share)) * chunk;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t k0 = kbeg; k0 < kbeg + chunk; k0 += COLS) {
#pragma unroll 1
        for (int r0 = 0; r0 < 512; r0 += (ROWS_PER_GROUP > 512 ? 512 : ROWS_PER_GROUP)) {
            float4 v[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                int row = r0 + j * RPI + lr;
                int64_t k = k0 + lc;
                if (ROWS_PER_GROUP > 512) { k += (int64_t)(row / 512) * COLS; row &= 511; }   // LPR = 4: two 16-k steps per group
                v[j] = *reinterpret_cast<const float4*>(s + (int64_t)row * K + k);
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        }
        if (ROWS_PER_GROUP > 512) k0 += COLS;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int LPR>
static void run(const float* s, int64_t K, int share, float* out) {
    const int nchunk = 8 * (256 / 8 / share > 0 ? 256 / 8 / share : 1) * 4;     // ~4 rounds of 256 workgroups
    const int64_t chunk = K / nchunk / 64 * 64;
    const int nwg = nchunk * share;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream<LPR>), dim3(nwg), dim3(256), 0, 0, s, K, chunk, share, out);
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream<LPR>), dim3(nwg), dim3(256), 0, 0, s, K, chunk, share, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / reps * 1e3, bytes = (double)nwg * 512 * chunk * 4;
    printf("piece %4d B  share %2d  %5d WGs  %9.1f us  %6.2f TB/s into the CUs (%.2f TB/s distinct)\n", LPR * 16, share, nwg, us,
           bytes / us / 1e6, bytes / share / us / 1e6);
}

int main() {
    const int64_t K = 1 << 20;                       // 512 rows x 1 Mi floats = 2 GiB
    float *s, *out;
    CHECK(hipMalloc(&s, 512 * K * 4)); CHECK(hipMalloc(&out, 1 << 16));
    CHECK(hipMemset(s, 0, 512 * K * 4));
    for (int share : {1, 4, 8}) {
        run<4>(s, K, share, out);
        run<8>(s, K, share, out);
        run<16>(s, K, share, out);
    }
    return 0;
}
