// LayerNormalization over the channels of an NCHW tensor (kccot_channel_layernorm_{fwd,bwd}_f32, include/kccot.h).
// The stock route -- permute to channels-last, copy, a row kernel over N*H*W rows of C = 32..256 elements, permute
// back -- was 10 % of the training iteration.  Here a thread owns one (n, pixel): consecutive threads touch consecutive
// addresses at every channel (stride H*W between channels), the channel loop runs three times forward (mean; centred
// variance; normalise -- the second and third read come from L2) and twice backward.  Parameter gradients: a second
// launch, one workgroup per (channel, chunk of samples), partial sums out, summed in chunk order by the caller.
#include "common.h"

namespace kccot {

__global__ __launch_bounds__(256) void chan_ln_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, int64_t npix, int C, int64_t HW, float eps,
                                                   float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= npix) return;
    const int64_t n = g / HW, p = g - n * HW;
    const float* xp = x + n * C * HW + p;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += xp[(int64_t)c * HW];
    const float m = s / (float)C;
    float v = 0.f;
    for (int c = 0; c < C; ++c) { const float d = xp[(int64_t)c * HW] - m; v = fmaf(d, d, v); }
    const float r = 1.0f / sqrtf(v / (float)C + eps);
    float* yp = y + n * C * HW + p;
    for (int c = 0; c < C; ++c) yp[(int64_t)c * HW] = (xp[(int64_t)c * HW] - m) * r * gamma[c] + beta[c];
    mean[g] = m;
    rstd[g] = r;
}

__global__ __launch_bounds__(256) void chan_ln_bwd_dx(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, int64_t npix, int C, int64_t HW,
                                                      float* __restrict__ dx) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= npix) return;
    const int64_t n = g / HW, p = g - n * HW, base = n * C * HW + p;
    const float m = mean[g], r = rstd[g];
    float a = 0.f, b = 0.f;      // a = sum_c dy g xhat, b = sum_c dy g
    for (int c = 0; c < C; ++c) {
        const float dg = dy[base + (int64_t)c * HW] * gamma[c];
        a = fmaf(dg, (x[base + (int64_t)c * HW] - m) * r, a);
        b += dg;
    }
    a /= (float)C; b /= (float)C;
    for (int c = 0; c < C; ++c) {
        const float dg = dy[base + (int64_t)c * HW] * gamma[c];
        const float xh = (x[base + (int64_t)c * HW] - m) * r;
        dx[base + (int64_t)c * HW] = r * (dg - b - xh * a);
    }
}

constexpr int LN_CHUNK_SAMPLES_TARGET = 64 * 1024;   // pixels per (channel, chunk) workgroup

static int ln_chunks(int N, int C, int HW) {
    // enough (channel, chunk) workgroups to fill the device, whole samples per chunk
    int64_t per = LN_CHUNK_SAMPLES_TARGET / HW;
    if (per < 1) per = 1;
    int64_t ch = (N + per - 1) / per;
    while (ch * C < 1024 && ch < N) { per = (per + 1) / 2; ch = (N + per - 1) / per; }
    return (int)ch;
}

// partials[chunk][0][c] = sum dy * xhat, partials[chunk][1][c] = sum dy over the chunk's samples
__global__ __launch_bounds__(256) void chan_ln_bwd_params(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          int N, int C, int64_t HW, int nchunk, float* __restrict__ partials) {
    __shared__ float red[16];
    const int c = blockIdx.x, chunk = blockIdx.y;
    const int per = (N + nchunk - 1) / nchunk;
    const int n0 = chunk * per, n1 = (n0 + per < N) ? n0 + per : N;
    float sg = 0.f, sb = 0.f;
    for (int n = n0; n < n1; ++n) {
        const float* dyp = dy + ((int64_t)n * C + c) * HW;
        const float* xp = x + ((int64_t)n * C + c) * HW;
        const float* mp = mean + (int64_t)n * HW;
        const float* rp = rstd + (int64_t)n * HW;
        for (int64_t p = threadIdx.x; p < HW; p += 256) {
            const float d = dyp[p];
            sg = fmaf(d, (xp[p] - mp[p]) * rp[p], sg);
            sb += d;
        }
    }
    const float tg = block_sum(sg, red);
    const float tb = block_sum(sb, red);
    if (threadIdx.x == 0) {
        partials[((int64_t)chunk * 2 + 0) * C + c] = tg;
        partials[((int64_t)chunk * 2 + 1) * C + c] = tb;
    }
}

}  // namespace kccot
using namespace kccot;

extern "C" int kccot_channel_layernorm_chunks(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return ln_chunks(N, C, HW);
}

extern "C" int kccot_channel_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, int N, int C, int HW,
                                               float eps, float* y, float* mean, float* rstd, kccot_stream_t stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return fail(KCCOT_EINVAL, "channel_layernorm_fwd: null pointer");
    if (N <= 0 || C <= 0 || HW <= 0 || !(eps >= 0.f)) return fail(KCCOT_EINVAL, "channel_layernorm_fwd: bad shape N=%d C=%d HW=%d", N, C, HW);
    const int64_t npix = (int64_t)N * HW;
    hipLaunchKernelGGL(chan_ln_fwd, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, npix, C,
                       (int64_t)HW, eps, y, mean, rstd);
    return launch_status("chan_ln_fwd");
}

extern "C" int kccot_channel_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                                               const float* rstd, int N, int C, int HW, float* dx, float* partials,
                                               kccot_stream_t stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !partials) return fail(KCCOT_EINVAL, "channel_layernorm_bwd: null pointer");
    if (N <= 0 || C <= 0 || HW <= 0) return fail(KCCOT_EINVAL, "channel_layernorm_bwd: bad shape N=%d C=%d HW=%d", N, C, HW);
    const int64_t npix = (int64_t)N * HW;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chan_ln_bwd_dx, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, dy, x, gamma, mean, rstd, npix, C,
                       (int64_t)HW, dx);
    int rc = launch_status("chan_ln_bwd_dx");
    if (rc) return rc;
    const int nchunk = ln_chunks(N, C, HW);
    hipLaunchKernelGGL(chan_ln_bwd_params, dim3(C, nchunk), dim3(256), 0, st, dy, x, mean, rstd, N, C, (int64_t)HW, nchunk, partials);
    return launch_status("chan_ln_bwd_params");
}
