// Micro-benchmark: how fast can 240 workgroups stream a [128 rows x K] fp32 stack (two tensors of 64
// rows, rows K floats apart) when each wave-load instruction touches
//   A: four 256-byte row pieces (the Gram kernel's staging pattern: 16 lanes per row),
//   B: two 512-byte row pieces, C: one 1-KB row piece,
// with one or two stages of loads in flight?  No LDS, no MFMA: loads are summed into a register.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_pattern.hip -o gpurun_out/hbm_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LANES_PER_ROW, int DEPTH>
__global__ __launch_bounds__(256) void stream(const float* __restrict__ a, const float* __restrict__ b, int64_t K,
                                              int64_t chunk, float* __restrict__ out) {
    // A block-wide load instruction covers ROWS_PER_INSTR rows x PIECE_COLS columns; a stage = 8 of them per
    // thread (32 KB per workgroup in flight per stage, as in the Gram kernel).
    constexpr int ROWS_PER_INSTR = 256 / LANES_PER_ROW;
    constexpr int NLOAD = 8;
    constexpr int PIECE_COLS = LANES_PER_ROW * 4;                      // 64, 128, 256
    constexpr int ROWS_PER_STAGE = NLOAD * ROWS_PER_INSTR;             // 128, 64, 32
    constexpr int SPC = 128 / ROWS_PER_STAGE;                          // stages per column step: 1, 2, 4
    const int t = threadIdx.x;
    const int lr = t / LANES_PER_ROW, lc = (t % LANES_PER_ROW) * 4;
    const int64_t kbeg = (int64_t)blockIdx.x * chunk, kend = kbeg + chunk < K ? kbeg + chunk : K;
    const int nstage = (int)((kend - kbeg) / PIECE_COLS) * SPC;
    float4 acc = make_float4(0, 0, 0, 0);
    float4 v[DEPTH][NLOAD];
    auto issue = [&](int d, int sidx) {
        const int64_t k0 = kbeg + (int64_t)(sidx / SPC) * PIECE_COLS;
        const int rowbase = (sidx % SPC) * ROWS_PER_STAGE;
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) {
            const int row = rowbase + j * ROWS_PER_INSTR + lr;
            const float* p = (row < 64 ? a + (int64_t)row * K : b + (int64_t)(row - 64) * K) + k0 + lc;
            v[d][j] = *reinterpret_cast<const float4*>(p);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (d < nstage) issue(d, d);
    for (int s0 = 0; s0 < nstage; s0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (s0 + d < nstage) {
#pragma unroll
                for (int j = 0; j < NLOAD; ++j) { acc.x += v[d][j].x; acc.y += v[d][j].y; acc.z += v[d][j].z; acc.w += v[d][j].w; }
                if (s0 + d + DEPTH < nstage) issue(d, s0 + d + DEPTH);
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int LPR, int DEPTH>
static void run(const char* name, const float* a, const float* b, int64_t K, float* out) {
    const int nwg = 240;
    const int64_t chunk = ((K / 256 + nwg - 1) / nwg) * 256;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((stream<LPR, DEPTH>), dim3(nwg), dim3(256), 0, 0, a, b, K, chunk, out);
    CHECK(hipEventRecord(e0));
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream<LPR, DEPTH>), dim3(nwg), dim3(256), 0, 0, a, b, K, chunk, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / reps * 1e3, bytes = 2.0 * 64 * K * 4;
    printf("%-34s %7.2f us  %6.2f TB/s\n", name, us, bytes / us / 1e6);
}

int main() {
    const int64_t K = 122880;
    float *a, *b, *out;
    CHECK(hipMalloc(&a, 64 * K * 4)); CHECK(hipMalloc(&b, 64 * K * 4)); CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(a, 0, 64 * K * 4)); CHECK(hipMemset(b, 0, 64 * K * 4));
    run<16, 1>("4 x 256 B per wave instr, depth 1", a, b, K, out);
    run<16, 2>("4 x 256 B per wave instr, depth 2", a, b, K, out);
    run<32, 1>("2 x 512 B per wave instr, depth 1", a, b, K, out);
    run<32, 2>("2 x 512 B per wave instr, depth 2", a, b, K, out);
    run<64, 1>("1 x 1 KB per wave instr, depth 1", a, b, K, out);
    run<64, 2>("1 x 1 KB per wave instr, depth 2", a, b, K, out);
    return 0;
}
