// The drop-in boundary WITHOUT Python or torch: a native program that links nothing but the HIP runtime and libkccot.so,
// allocates with hipMalloc, calls compute_sinkhorn_loss's one-call entry points (kccot_sinkhorn_loss_{fwd,bwd}_f32) on its
// own stream and checks the cost matrices against a double-precision host evaluation of gan_utils.py:14-17,34-38 and the
// gradient against a finite difference of the library's own loss.  Built with hipcc and run by
// tests/test_gpu_parity.py::test_c_abi_from_a_native_program (-m gpu).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kccot.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
#define KCHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "kccot error %d (%s) at %s:%d\n", rc_, kccot_last_error(), __FILE__, __LINE__); return 1; } } while (0)

static double lcg(unsigned long long& s) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(s >> 11) / 9007199254740992.0; }

template <class T> static T* dev(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main() {
    const int B = 16, T = 6, J = 8, L = 100;
    const int64_t K = 6 * 64;                        // [B, H=4, T=6, W=4, C=4] read as [B, K]
    const float sc = 1.0f / 15.0f, eps = 1.0f;
    unsigned long long seed = 12345;
    std::vector<float> real(B * K), fake(B * K), f[4];
    for (auto& v : real) v = (float)lcg(seed);
    for (size_t i = 0; i < fake.size(); ++i) { double y = real[i] + 0.1 * (lcg(seed) - 0.5); fake[i] = (float)(y < 0 ? 0 : (y > 1 ? 1 : y)); }
    for (auto& t : f) { t.resize(B * T * J); for (auto& v : t) v = (float)lcg(seed); }   // h_fake, h_real, m_real, m_fake
    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));
    float *d_real = dev(real), *d_fake = dev(fake), *d_f[4];
    for (int i = 0; i < 4; ++i) d_f[i] = dev(f[i]);
    if (!d_real || !d_fake || !d_f[0] || !d_f[1] || !d_f[2] || !d_f[3]) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    float *C3, *uh, *vh, *cost3, *loss, *gloss, *dfake;
    int32_t *nits, *ticket;
    void* ws;
    const size_t wsb = kccot_sinkhorn_loss_workspace_bytes(B, K);
    HIPCHECK(hipMalloc(&C3, 3 * B * B * 4)); HIPCHECK(hipMalloc(&uh, 3 * L * B * 4)); HIPCHECK(hipMalloc(&vh, 3 * L * B * 4));
    HIPCHECK(hipMalloc(&cost3, 3 * 4)); HIPCHECK(hipMalloc(&loss, 4)); HIPCHECK(hipMalloc(&gloss, 4));
    HIPCHECK(hipMalloc(&dfake, B * K * 4)); HIPCHECK(hipMalloc(&nits, 6 * 4)); HIPCHECK(hipMalloc(&ticket, 4));
    HIPCHECK(hipMalloc(&ws, wsb ? wsb : 16));
    HIPCHECK(hipMemset(ticket, 0, 4));
    const float one = 1.0f;
    HIPCHECK(hipMemcpy(gloss, &one, 4, hipMemcpyHostToDevice));
    auto forward = [&](const float* fk, float* out_loss) -> int {
        KCHECK(kccot_sinkhorn_loss_fwd_f32(d_real, fk, B, K, sc, d_f[0], d_f[1], d_f[2], d_f[3], T, J, eps, L, 100, 1e-2f, 0u, C3, uh, vh,
                                           cost3, nits, loss, ticket, ws, wsb, st));
        HIPCHECK(hipStreamSynchronize(st));
        HIPCHECK(hipMemcpy(out_loss, loss, 4, hipMemcpyDeviceToHost));
        return 0;
    };
    float l0 = 0.f;
    if (forward(d_fake, &l0)) return 1;
    std::vector<float> hC(3 * B * B);
    HIPCHECK(hipMemcpy(hC.data(), C3, hC.size() * 4, hipMemcpyDeviceToHost));
    // gan_utils.py:221-223: xy = (real, fake, h_fake, m_real), xx = (real, real, h_real, m_real), yy = (fake, fake, h_fake, m_fake)
    const std::vector<float>* X[3] = {&real, &real, &fake};
    const std::vector<float>* Y[3] = {&fake, &real, &fake};
    const int hi[3] = {0, 1, 0}, mi[3] = {2, 2, 3};
    double worst = 0.0, cmax = 0.0;
    for (int p = 0; p < 3; ++p)
        for (int i = 0; i < B; ++i)
            for (int j = 0; j < B; ++j) {
                double d = 0.0, c = 0.0;
                for (int64_t k = 0; k < K; ++k) { const double t = (double)(*X[p])[i * K + k] - (double)(*Y[p])[j * K + k]; d += t * t; }
                for (int t = 0; t + 1 < T; ++t)
                    for (int k = 0; k < J; ++k)
                        c += (double)f[hi[p]][(i * T + t) * J + k] * ((double)f[mi[p]][(j * T + t + 1) * J + k] - (double)f[mi[p]][(j * T + t) * J + k]);
                const double ref = sc * (d + c), got = hC[(p * B + i) * B + j];
                worst = fmax(worst, fabs(got - ref));
                cmax = fmax(cmax, fabs(ref));
            }
    if (!(worst <= 1e-5 * cmax)) { fprintf(stderr, "cost matrices off by %g of max %g\n", worst, cmax); return 1; }
    std::vector<int32_t> hn(6);
    HIPCHECK(hipMemcpy(hn.data(), nits, 24, hipMemcpyDeviceToHost));
    if (hn[0] != 100 || hn[1] != 100 || hn[2] != 100 || !std::isfinite(l0)) { fprintf(stderr, "solve: nits %d %d %d loss %g\n", hn[0], hn[1], hn[2], l0); return 1; }
    // backward, then a central difference of the library's own forward along the gradient direction
    KCHECK(kccot_sinkhorn_loss_bwd_f32(gloss, d_real, d_fake, B, K, sc, d_f[0], d_f[1], d_f[2], d_f[3], T, J, eps, L, C3, uh, vh, nits, dfake,
                                       nullptr, nullptr, nullptr, nullptr, ws, wsb, st));
    HIPCHECK(hipStreamSynchronize(st));
    std::vector<float> g(B * K);
    HIPCHECK(hipMemcpy(g.data(), dfake, g.size() * 4, hipMemcpyDeviceToHost));
    double g2 = 0.0;
    for (float v : g) g2 += (double)v * v;
    if (!(g2 > 0.0) || !std::isfinite(g2)) { fprintf(stderr, "gradient norm %g\n", g2); return 1; }
    const double h = 0.05 / sqrt(g2);                               // a step of 0.05 in the L2 norm of the video batch
    std::vector<float> fp(fake), fm(fake);
    for (size_t i = 0; i < g.size(); ++i) { fp[i] = (float)(fake[i] + h * g[i]); fm[i] = (float)(fake[i] - h * g[i]); }
    float *d_fp = dev(fp), *d_fm = dev(fm), lp = 0.f, lm = 0.f;
    if (!d_fp || !d_fm || forward(d_fp, &lp) || forward(d_fm, &lm)) return 1;
    const double fd = ((double)lp - (double)lm) / (2.0 * h);          // should equal |g|^2
    if (!(fabs(fd - g2) <= 0.05 * g2)) { fprintf(stderr, "directional derivative %g vs |g|^2 %g\n", fd, g2); return 1; }
    printf("abi_gpu_smoke ok: loss %.6f, max cost error %.2e of %.3g, directional derivative %.5g vs %.5g\n", l0, worst, cmax, fd, g2);
    return 0;
}
