// Gaussian kernel smoothing of [B,H,T,W,C] video tensors (replaces KernelSmoothing of the
// reference, data_utils.py:478-582): normalised (2r+1)-tap Gaussian along T (temporal_convolution)
// or along T, H and W (gaussian_convolution3D -- its 7x7x7 kernel, data_utils.py:493-501, is the
// outer product of the 1-D kernel), REFLECT borders (data_utils.py:512-513,562-565), then division
// by the maximum of the whole smoothed tensor (data_utils.py:520,573,581).
//
// The reference reaches the conv through four physical transposes and a padded copy; here every
// axis is convolved in place on the native layout: along any axis consecutive threads touch
// consecutive addresses (W*C is innermost), the 2r neighbour lines come from L1/L2.
// This is the first, pass-per-axis version: HBM traffic is one read + one write of the tensor
// per axis plus the max/scale pass (algorithmic minimum: one read + one write in total).
#include "common.h"
#include "options.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <type_traits>
#include <stdio.h>
#include <stdint.h>

namespace kccot {

constexpr int SM_MAXR = 7;

struct Taps { float w[2 * SM_MAXR + 1]; int r; };

// data_utils.py:483-491: kernel = exp(-0.5/sigma^2 * x^2) (fp32), normalised by its fp32 sum
static Taps make_taps(float sigma, int r) {
    Taps tp{};
    tp.r = r;
    const float coef = (float)(-0.5 / ((double)sigma * (double)sigma));
    float sum = 0.f;
    for (int d = -r; d <= r; ++d) { tp.w[d + r] = expf(coef * (float)(d * d)); sum += tp.w[d + r]; }
    for (int d = 0; d <= 2 * r; ++d) tp.w[d] /= sum;
    return tp;
}

__device__ __forceinline__ int reflect(int p, int len) { return p < 0 ? -p : (p >= len ? 2 * (len - 1) - p : p); }

// out[e] = sum_d w[d] * in[e with its axis position moved to reflect(p+d)]           (adjoint = 0)
// out[e] = sum over all (t,d) whose reflected source is p of w[d] * in[.. t ..]      (adjoint = 1)
// If blockmax != null also writes the maximum of the block's outputs (forward, last axis).
__global__ __launch_bounds__(256) void conv_axis(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                 int len, int64_t stride, Taps tp, int adjoint,
                                                 float* __restrict__ blockmax) {
    __shared__ float red[16];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    const bool ok = e < n;
    if (ok) {
        const int p = (int)((e / stride) % len);
        const float* base = in + (e - (int64_t)p * stride);
        const int r = tp.r;
        if (!adjoint) {
            for (int d = -r; d <= r; ++d) acc = fmaf(tp.w[d + r], base[(int64_t)reflect(p + d, len) * stride], acc);
        } else {
            for (int d = -r; d <= r; ++d) {
                const float w = tp.w[d + r];
                int t = p - d;
                if (t >= 0 && t < len) acc = fmaf(w, base[(int64_t)t * stride], acc);
                t = -p - d;
                if (p >= 1 && t >= 0 && t < len) acc = fmaf(w, base[(int64_t)t * stride], acc);
                t = 2 * (len - 1) - p - d;
                if (p <= len - 2 && t >= 0 && t < len) acc = fmaf(w, base[(int64_t)t * stride], acc);
            }
        }
        out[e] = acc;
    }
    if (blockmax) {
        const float m = block_max(ok ? acc : -FLT_MAX, red);
        if (threadIdx.x == 0) blockmax[blockIdx.x] = m;
    }
}

__global__ __launch_bounds__(256) void copy_with_blockmax(const float* __restrict__ in, float* __restrict__ out,
                                                          int64_t n, float* __restrict__ blockmax) {
    __shared__ float red[16];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = e < n;
    const float v = ok ? in[e] : -FLT_MAX;
    if (ok && out != in) out[e] = v;
    const float m = block_max(v, red);
    if (threadIdx.x == 0) blockmax[blockIdx.x] = m;
}

__global__ __launch_bounds__(1024) void reduce_blockmax(const float* __restrict__ blockmax, int64_t nb,
                                                        float* __restrict__ max_out) {
    __shared__ float red[16];
    float m = -FLT_MAX;
    for (int64_t i = threadIdx.x; i < nb; i += blockDim.x) m = fmaxf(m, blockmax[i]);
    m = block_max(m, red);
    if (threadIdx.x == 0) max_out[0] = m;
}

__global__ __launch_bounds__(256) void divide_by(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                 const float* __restrict__ mx) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = in[e] / mx[0];
}

// backward of out = s / max(s):  ds = gout/m - [out == 1] * (sum gout*out) / (m * #ties)
// stage 1: per-block partial (dot, ties) ; stage 2: combine ; stage 3: elementwise
__global__ __launch_bounds__(256) void maxnorm_bwd_partial(const float* __restrict__ gout, const float* __restrict__ out,
                                                           int64_t n, float* __restrict__ pdot, float* __restrict__ pcnt) {
    __shared__ float red[16];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = e < n;
    const float o = ok ? out[e] : 0.f;
    const float d = block_sum(ok ? gout[e] * o : 0.f, red);
    const float c = block_sum((ok && o == 1.0f) ? 1.f : 0.f, red);
    if (threadIdx.x == 0) { pdot[blockIdx.x] = d; pcnt[blockIdx.x] = c; }
}

__global__ __launch_bounds__(1024) void maxnorm_bwd_combine(const float* __restrict__ pdot, const float* __restrict__ pcnt,
                                                            int64_t nb, float* __restrict__ res) {
    __shared__ float red[16];
    double d = 0.0;
    float c = 0.f;
    for (int64_t i = threadIdx.x; i < nb; i += blockDim.x) { d += (double)pdot[i]; c += pcnt[i]; }
    const float ds = block_sum((float)d, red);
    const float cs = block_sum(c, red);
    if (threadIdx.x == 0) { res[0] = ds; res[1] = cs; }
}

__global__ __launch_bounds__(256) void maxnorm_bwd_apply(const float* __restrict__ gout, const float* __restrict__ out,
                                                         int64_t n, const float* __restrict__ mx,
                                                         const float* __restrict__ res, float* __restrict__ ds) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float m = mx[0];
    float v = gout[e] / m;
    if (out[e] == 1.0f && res[1] > 0.f) v -= res[0] / (m * res[1]);
    ds[e] = v;
}

// ------------------------------------------------------------------------------------------
// Fused plane kernel: T-, W- and H-convolution in ONE pass over the tensor.
//
// A workgroup owns one sample b and a segment of HSEG image rows h.  It streams the (T x W*C)
// planes of that segment (7.7 KB at 30 x 64 x 1) through LDS: the T- and W-stencils run inside the
// plane (no halo: both axes lie fully inside it), and the H-stencil is a sliding window of the
// last 2R+1 planes kept in registers (each thread owns PPT fixed positions of the plane), so every
// input element is read once per segment (+R halo planes either side, served by L2) and every
// output element is written once, coalesced.  The global maximum costs no extra pass over HBM
// either: pass 1 runs the same kernel WITHOUT writing (block maxima only), pass 2 recomputes and
// writes s / max -- the 31 MB input is still in the 256 MB Infinity Cache.
//
// The same kernel is the backward (ADJ): the adjoint of "REFLECT-pad then correlate" along an axis
// of length L is again a gather over the +-R window, with position-dependent weights
//     din[p] = sum_{|k|<=R, 0<=p+k<L} g[p+k] * ( w[k] + [p>=1] wq(-2p-k) + [p<=L-2] wq(2(L-1)-2p-k) )
// (wq = w inside +-R, 0 outside), and the load stage forms ds = gout/m - [out==1]*corr on the fly.
// ------------------------------------------------------------------------------------------
constexpr int SP_PPT = 4;

struct PlaneArgs {
    const float* in;       // forward: input video; adjoint: gout
    const float* out_fwd;  // adjoint only: the forward's normalised output (arg-max detection)
    float* out;            // forward pass 2: normalised output; adjoint: din; null in pass 1
    float* blockmax;       // forward pass 1: per-workgroup maxima
    const float* mx;       // device scalar: tensor maximum (pass 2 / adjoint)
    const float* res;      // adjoint: {sum gout*out, #ties}
    int B, H, T, W, C, hseg;
    unsigned axes;
    Taps tp;
};

template <int R>
__device__ __forceinline__ float wq(const Taps& tp, int d) { return (d >= -R && d <= R) ? tp.w[d + R] : 0.f; }

// weight of source position p+k for output position p (axis length L); returns false if the
// source does not exist (adjoint) -- forward always returns true with src = reflect(p+k)
template <int R, bool ADJ>
__device__ __forceinline__ bool tap(const Taps& tp, int p, int k, int L, int& src, float& w) {
    if (!ADJ) { src = reflect(p + k, L); w = tp.w[k + R]; return true; }
    src = p + k;
    if (src < 0 || src >= L) return false;
    w = tp.w[k + R] + (p >= 1 ? wq<R>(tp, -2 * p - k) : 0.f) + (p <= L - 2 ? wq<R>(tp, 2 * (L - 1) - 2 * p - k) : 0.f);
    return true;
}

template <int R, bool ADJ, int NT>
__global__ __launch_bounds__(NT) void smooth_plane(PlaneArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sp_lds[];
    __shared__ float red[16];
    const int T = a.T, WC = a.W * a.C, P = T * WC;       // plane size
    float* bufA = sp_lds;
    float* bufB = sp_lds + P;
    const int b = blockIdx.y, h0 = blockIdx.x * a.hseg;
    const int h1 = min(h0 + a.hseg, a.H);
    const bool doT = a.axes & KCCOT_SMOOTH_T, doH = a.axes & KCCOT_SMOOTH_H, doW = a.axes & KCCOT_SMOOTH_W;
    const int nth = blockDim.x, t = threadIdx.x;
    const int64_t plane_stride = (int64_t)P, sample_stride = (int64_t)a.H * P;
    const float* inb = a.in + (int64_t)b * sample_stride;
    const float* ofb = ADJ ? a.out_fwd + (int64_t)b * sample_stride : nullptr;
    float* outb = a.out ? a.out + (int64_t)b * sample_stride : nullptr;
    float m = 1.f, corr = 0.f;
    if (ADJ) { m = a.mx[0]; corr = a.res[1] > 0.f ? a.res[0] / (m * a.res[1]) : 0.f; }
    else if (outb) m = a.mx[0];

    // positions owned by this thread: pos = t + nth*i  (consecutive threads, consecutive addresses)
    int prow[SP_PPT], pcol[SP_PPT];   // t index, w*C+c index
    bool pok[SP_PPT], tin[SP_PPT], win_[SP_PPT];
#pragma unroll
    for (int i = 0; i < SP_PPT; ++i) {
        const int pos = t + nth * i;
        pok[i] = pos < P;
        prow[i] = pok[i] ? pos / WC : 0;
        pcol[i] = pok[i] ? pos % WC : 0;
        const int pwi = pcol[i] / a.C;
        // interior positions: no reflected / folded tap, the stencil is 2R+1 fixed-offset reads
        tin[i] = prow[i] > R && prow[i] < T - 1 - R;
        win_[i] = pwi > R && pwi < a.W - 1 - R;
    }
    float win[SP_PPT][2 * R + 1];
#pragma unroll
    for (int i = 0; i < SP_PPT; ++i)
#pragma unroll
        for (int j = 0; j <= 2 * R; ++j) win[i][j] = 0.f;

    float vmax = -FLT_MAX;
    const int hlo = doH ? h0 - R : h0, hhi = doH ? h1 + R : h1;   // planes to stream
    // plane hp (forward: reflected index; adjoint: zero plane outside [0,H)), fetched one plane ahead
    float nin[SP_PPT], nof[SP_PPT];
    auto fetch = [&](int hp) {
        int hsrc = hp;
        bool exists = hp < hhi;
        if (!ADJ) hsrc = reflect(hp, a.H); else exists = exists && hp >= 0 && hp < a.H;
#pragma unroll
        for (int i = 0; i < SP_PPT; ++i) {
            nin[i] = 0.f; nof[i] = 0.f;
            if (pok[i] && exists) {
                const int64_t off = (int64_t)hsrc * plane_stride + t + nth * i;
                nin[i] = inb[off];
                if (ADJ) nof[i] = ofb[off];
            }
        }
    };
    fetch(hlo);
    for (int hp = hlo; hp < hhi; ++hp) {
        float v[SP_PPT];
#pragma unroll
        for (int i = 0; i < SP_PPT; ++i) v[i] = ADJ ? (nin[i] / m - (nof[i] == 1.0f ? corr : 0.f)) : nin[i];
        fetch(hp + 1);
        // ---- T stencil inside the plane
        if (doT) {
            lds_barrier();   // previous plane's readers of bufA are done
#pragma unroll
            for (int i = 0; i < SP_PPT; ++i) if (pok[i]) bufA[t + nth * i] = v[i];
            lds_barrier();
#pragma unroll
            for (int i = 0; i < SP_PPT; ++i) {
                float acc = 0.f;
                if (pok[i]) {
                    if (tin[i]) {
                        const float* c0 = bufA + t + nth * i;
#pragma unroll
                        for (int k = -R; k <= R; ++k) acc = fmaf(a.tp.w[k + R], c0[k * WC], acc);
                    } else {
#pragma unroll 1
                        for (int k = -R; k <= R; ++k) {
                            int src; float w;
                            if (tap<R, ADJ>(a.tp, prow[i], k, T, src, w)) acc = fmaf(w, bufA[src * WC + pcol[i]], acc);
                        }
                    }
                }
                v[i] = acc;
            }
        }
        // ---- W stencil inside the plane (stride C along the contiguous W*C axis)
        if (doW) {
            lds_barrier();
#pragma unroll
            for (int i = 0; i < SP_PPT; ++i) if (pok[i]) bufB[t + nth * i] = v[i];
            lds_barrier();
#pragma unroll
            for (int i = 0; i < SP_PPT; ++i) {
                float acc = 0.f;
                if (pok[i]) {
                    if (win_[i]) {
                        const float* c0 = bufB + t + nth * i;
#pragma unroll
                        for (int k = -R; k <= R; ++k) acc = fmaf(a.tp.w[k + R], c0[k * a.C], acc);
                    } else {
                        const int wpos = pcol[i] / a.C, c = pcol[i] - wpos * a.C;
                        const float* rowp = bufB + prow[i] * WC + c;
#pragma unroll 1
                        for (int k = -R; k <= R; ++k) {
                            int src; float w;
                            if (tap<R, ADJ>(a.tp, wpos, k, a.W, src, w)) acc = fmaf(w, rowp[src * a.C], acc);
                        }
                    }
                }
                v[i] = acc;
            }
        }
        // ---- H stencil: sliding register window over planes
        int hout = hp;   // output plane completed by this input plane
        if (doH) {
#pragma unroll
            for (int i = 0; i < SP_PPT; ++i) {
#pragma unroll
                for (int j = 0; j < 2 * R; ++j) win[i][j] = win[i][j + 1];
                win[i][2 * R] = v[i];
            }
            hout = hp - R;
            if (hout < h0) continue;   // window not full yet (uniform across the block)
            float wt[2 * R + 1];
            bool wok[2 * R + 1];
#pragma unroll
            for (int j = 0; j <= 2 * R; ++j) {
                int src;
                wok[j] = tap<R, ADJ>(a.tp, hout, j - R, a.H, src, wt[j]);
            }
#pragma unroll
            for (int i = 0; i < SP_PPT; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j <= 2 * R; ++j)
                    if (wok[j]) acc = fmaf(wt[j], win[i][j], acc);
                v[i] = acc;
            }
        }
        // ---- emit plane hout
#pragma unroll
        for (int i = 0; i < SP_PPT; ++i) {
            if (!pok[i]) continue;
            if (!ADJ) vmax = fmaxf(vmax, v[i]);
            if (outb) outb[(int64_t)hout * plane_stride + t + nth * i] = ADJ ? v[i] : v[i] / m;
        }
    }
    if (!ADJ && a.blockmax) {
        const float bm = block_max(vmax, red);
        if (t == 0) a.blockmax[blockIdx.y * gridDim.x + blockIdx.x] = bm;
    }
}

// ---- smoothing along a strided axis as a pure stream ----------------------------------------------
// Along T (stride W*C) and H (stride T*W*C) a thread can own a short contiguous piece (VW floats) of
// one line of the axis, issue the loads of all L <= LMAX steps of it at once (L x 4 VW bytes in
// flight per thread: a stream, not a latency chain) and walk the axis with a (2R+1)-deep register
// window -- no LDS, no barrier, consecutive lanes on consecutive addresses at every step.
// REFLECT borders: the head of the window is filled mirrored; near the end the incoming value
// x[2(L-1)-(p+R)] is already inside the window (slot 2(L-1-p)).  Same fma order as the other kernels.
// Everything is straight-line code on vector types (selects, no branches around the register arrays),
// so the line stays in VGPRs; steps p >= L of the unrolled walk run on clamped copies and are dropped.
//   WALK_MAX   forward, per-workgroup maxima only          WALK_WRITE  forward, writes s / max
//   WALK_RAW   forward, writes s                            WALK_ADJ    adjoint of WALK_RAW
//   WALK_ADJX  adjoint preceded by the max-normalisation adjoint  x = g / max - corr * [out == 1]
// Adjoint weights: w_k(p) = w[k] + [p >= 1] w[-2p-k] + [p <= L-2] w[2(L-1)-2p-k] (taps outside -R..R
// are zero), sources outside [0,L) do not exist; the head fold is known at compile time, the tail
// fold depends on d = L-1-p and is chosen with scalar selects.
enum { WALK_MAX = 0, WALK_WRITE = 1, WALK_RAW = 2, WALK_ADJ = 3, WALK_ADJX = 4, WALK_ADJS = 5, WALK_RAW_TW = 6 };
// WALK_ADJS (round 3): the first adjoint stage WITHOUT the arg-max correction, x = g / max, which ALSO gathers what that
// correction needs while it streams gout and the forward output anyway: one TieRec per workgroup = its sum of g * out,
// its number of elements with out == 1 and the positions of the first TIE_PER_WG of them.  The adjoint is linear, so
// A^T (g / max - corr [out == 1]) = A^T (g / max) - corr A^T [out == 1]:  the second term touches the (2R+1)^axes
// neighbours of each arg-max element and is subtracted afterwards by maxnorm_bwd_fixup -- the separate pass over gout
// and out that computed the two sums first (two tensor reads of the five of the temporal backward) is gone.  No global
// atomics and nothing to zero in front of the launch: every workgroup writes its own record.
constexpr int SMOOTH_MAX_TIES = 32;    // arg-max elements the sparse fix-up handles; more -> the dense fallback (on the device)
constexpr int TIE_PER_WG = 4;
struct TieRec { double dot; long long idx[TIE_PER_WG]; int ties; int pad; };

struct TieNote {                       // LDS side of one workgroup's record (workgroups of 256 threads)
    double part[4]; long long idx[TIE_PER_WG]; int n;
    __device__ __forceinline__ void clear() { if (threadIdx.x == 0) n = 0; __syncthreads(); }
    __device__ __forceinline__ void add(long long e) { const int slot = atomicAdd(&n, 1); if (slot < TIE_PER_WG) idx[slot] = e; }
    // `dot` = the thread's sum of g * out, kept in fp64: the sum cancels (g has both signs), and the correction it feeds
    // is the largest entry of the gradient
    __device__ __forceinline__ void store(TieRec* rec, double dot) {
        dot = wave_sum_d(dot);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = dot;
        __syncthreads();
        if (threadIdx.x == 0) {
            rec->dot = (part[0] + part[1]) + (part[2] + part[3]); rec->ties = n; rec->pad = 0;
            for (int i = 0; i < TIE_PER_WG; ++i) rec->idx[i] = idx[i];
        }
    }
};

// WALK_RAW_TW (C == 1, four consecutive w per thread, W/4 a power of two <= 64): the T walk with the W stencil of
// smooth_w1 applied to every T-smoothed piece before it is stored -- the left / right neighbour pieces of a row are the
// neighbouring LANES' registers (one row = W/4 consecutive lanes of a wave), fetched with DPP wave shifts; REFLECT at the
// row ends as in smooth_w1.  Same fma order as the two kernels it replaces (bit-identical), one read + one write of the
// tensor less.
__device__ __forceinline__ float from_prev_lane(float v) {   // lane i <- lane i - 1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_next_lane(float v) {   // lane i <- lane i + 1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// (A cooperative single-launch form of the last stage -- lines kept in registers across a device-wide barrier, tensor read
// once and written once -- was built in round 2, measured SLOWER than the two passes (22.7 + 4.6 us against 8.3 + 12.5 us
// at configs[1], profiles/r03a_prof_smooth_coop_vs_two_pass.txt) and removed in round 3.)
struct WalkArgs {
    const float* in;       // forward: input; adjoint: incoming gradient
    const float* out_fwd;  // WALK_ADJX: the forward's normalised output
    float* out;
    float* blockmax;       // WALK_MAX
    const float* mx;       // WALK_WRITE / WALK_ADJX: device scalar, the tensor maximum
    const float* res;      // WALK_ADJX: {sum gout*out, #ties}
    float* mx_out;         // WALK_WRITE with nblk > 0: the tensor maximum is written here
    TieRec* ties;          // WALK_ADJS: one record per workgroup
    const int* run_if;     // WALK_ADJ / WALK_ADJX: non-null -> the launch does nothing unless *run_if != 0 (the backward's
                           // dense chain, decided on the device by maxnorm_bwd_fixup)
    int nblk;              // WALK_WRITE: > 0 = reduce blockmax[0, nblk) (the preceding WALK_MAX launch's) instead of reading mx
    int L;                 // axis length
    int64_t S;             // axis stride in floats
    int64_t inner;         // S / VW pieces per axis-stride block
    int64_t ncols;         // pieces in total = numel / L / VW
    Taps tp;
};

template <int VW> struct WalkVec { typedef float type __attribute__((ext_vector_type(VW))); };
template <> struct WalkVec<1> { typedef float type; };

template <int VW, typename V> __device__ __forceinline__ float& vat(V& v, int c) {
    if constexpr (VW == 1) return v; else return reinterpret_cast<float*>(&v)[c];
}

template <int R, int LMAX, int VW, int MODE>
__global__ __launch_bounds__(256) void smooth_walk(WalkArgs a) {
    typedef typename WalkVec<VW>::type V;
    constexpr bool ADJ = MODE == WALK_ADJ || MODE == WALK_ADJX || MODE == WALK_ADJS;
    __shared__ float red[16];
    if ((MODE == WALK_ADJ || MODE == WALK_ADJX) && a.run_if && *a.run_if == 0) return;
    const int L = a.L;
    const int64_t S = a.S;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = gid < a.ncols;
    const int64_t g = ok ? gid : 0;
    const int64_t off = (g / a.inner) * L * S + (g % a.inner) * VW;
    const float* src = a.in + off;
    float m = 1.f, corr = 0.f;
    if ((MODE == WALK_WRITE && a.nblk <= 0) || MODE == WALK_ADJX || MODE == WALK_ADJS) m = a.mx[0];
    if (MODE == WALK_ADJX) corr = a.res[1] > 0.f ? a.res[0] / (m * a.res[1]) : 0.f;
    double sdot = 0.0;                  // WALK_ADJS
    __shared__ TieNote note;
    if (MODE == WALK_ADJS) note.clear();
    V x[LMAX];
#pragma unroll
    for (int p = 0; p < LMAX; ++p) {    // addresses clamped, never predicated: all loads issue back to back
        const int64_t o = (int64_t)(p < L ? p : L - 1) * S;
        x[p] = *reinterpret_cast<const V*>(src + o);
        if (MODE == WALK_ADJX) {
            V of = *reinterpret_cast<const V*>(a.out_fwd + off + o);
#pragma unroll
            for (int c = 0; c < VW; ++c) vat<VW>(x[p], c) = vat<VW>(x[p], c) / m - (vat<VW>(of, c) == 1.0f ? corr : 0.f);
        }
        if (MODE == WALK_ADJS) {
            V of = *reinterpret_cast<const V*>(a.out_fwd + off + o);
#pragma unroll
            for (int c = 0; c < VW; ++c) {
                const float gg = vat<VW>(x[p], c), oo = vat<VW>(of, c);
                if (ok && p < L) {
                    sdot = fma((double)gg, (double)oo, sdot);
                    if (oo == 1.0f) note.add((long long)(off + o + c));
                }
                vat<VW>(x[p], c) = gg / m;
            }
        }
    }
    if (MODE == WALK_WRITE && a.nblk > 0) {
        // every workgroup reduces the preceding launch's block maxima itself, under its own line loads: the separate
        // single-workgroup reduction launch between the two passes (4.7 us at configs[1]) is gone
        float v = -FLT_MAX;
        for (int i = threadIdx.x; i < a.nblk; i += 256) v = fmaxf(v, a.blockmax[i]);
        m = block_max(v, red);
        if (blockIdx.x == 0 && threadIdx.x == 0) a.mx_out[0] = m;
    }
    V zero;
#pragma unroll
    for (int c = 0; c < VW; ++c) vat<VW>(zero, c) = 0.f;
    V win[2 * R + 1];                   // win[j] = x at position p - R + j
#pragma unroll
    for (int j = 0; j <= 2 * R; ++j) {
        if (!ADJ) win[j] = x[j < R ? R - j : j - R];                       // mirrored head
        else win[j] = (j < R) ? zero : ((j - R < L) ? x[j - R] : zero);    // nothing before 0 / after L-1
    }
    float vmax = -FLT_MAX;
#pragma unroll
    for (int p = 0; p < LMAX; ++p) {
        const bool live = p < L;        // uniform
        const int d = L - 1 - p;        // distance to the last position (uniform)
        V acc = zero;
#pragma unroll
        for (int k = -R; k <= R; ++k) {
            float w = a.tp.w[k + R];
            if (ADJ) {
                const int hf = -2 * p - k;                       // head fold: compile-time
                if (p >= 1 && hf >= -R && hf <= R) w += a.tp.w[hf + R];
                float tf = 0.f;                                  // tail fold: 2d - k in [-R, R], d >= 1
#pragma unroll
                for (int dd = 1; dd <= R; ++dd)
                    if (2 * dd - k >= -R && 2 * dd - k <= R) tf = (d == dd) ? a.tp.w[2 * dd - k + R] : tf;
                w += tf;
            }
            if constexpr (VW == 1) acc = fmaf(w, win[k + R], acc);
            else {
                V wv;
#pragma unroll
                for (int c = 0; c < VW; ++c) vat<VW>(wv, c) = w;
                acc = __builtin_elementwise_fma(wv, win[k + R], acc);
            }
        }
        if constexpr (MODE == WALK_RAW_TW && VW == 4) {
            const int pc = (int)(g % a.inner);
            const bool first = pc == 0, last = pc == (int)a.inner - 1;
            float v[12];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[4 + c] = vat<VW>(acc, c);
                v[c] = from_prev_lane(v[4 + c]);
                v[8 + c] = from_next_lane(v[4 + c]);
            }
            // mirrored values: x[-j] = x[j] (j = 1..4), x[W-1+j] = x[W-1-j]
            const float m4 = v[8], m4b = v[3];               // x[4] resp. x[W-5] as seen from the first / last piece
            v[3] = first ? v[5] : v[3]; v[2] = first ? v[6] : v[2]; v[1] = first ? v[7] : v[1]; v[0] = first ? m4 : v[0];
            v[8] = last ? v[6] : v[8]; v[9] = last ? v[5] : v[9]; v[10] = last ? v[4] : v[10]; v[11] = last ? m4b : v[11];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float o = 0.f;
#pragma unroll
                for (int k = -R; k <= R; ++k) o = fmaf(a.tp.w[k + R], v[4 + j + k], o);
                vat<VW>(acc, j) = o;
            }
        }
        if (MODE == WALK_MAX) {
            float mx = vat<VW>(acc, 0);
#pragma unroll
            for (int c = 1; c < VW; ++c) mx = fmaxf(mx, vat<VW>(acc, c));
            vmax = live ? fmaxf(vmax, mx) : vmax;
        } else {
            if (MODE == WALK_WRITE) {
#pragma unroll
                for (int c = 0; c < VW; ++c) vat<VW>(acc, c) = vat<VW>(acc, c) / m;
            }
            if (ok && live) *reinterpret_cast<V*>(a.out + off + (int64_t)p * S) = acc;
        }
        // advance to p + 1: the incoming position is p + 1 + R
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) win[j] = win[j + 1];
        V inc;
        if (!ADJ) {
            // past the end the mirror image x[2(L-1) - (p+1+R)] is in slot 2(d-1), d - 1 < R
            inc = win[0];
            inc = d - 1 >= 1 ? win[2] : inc;
            inc = d - 1 >= 2 ? win[4] : inc;
            if (R > 3) inc = d - 1 >= 3 ? win[6] : inc;
        } else {
            inc = zero;
        }
        if (p + 1 + R < LMAX) inc = (p + 1 + R < L) ? x[p + 1 + R < LMAX ? p + 1 + R : 0] : inc;
        win[2 * R] = inc;
    }
    if (MODE == WALK_MAX && a.blockmax) {
        const float bm = block_max(ok ? vmax : -FLT_MAX, red);
        if (threadIdx.x == 0) a.blockmax[blockIdx.x] = bm;
    }
    if (MODE == WALK_ADJS) note.store(a.ties + blockIdx.x, sdot);
}

// ---- W axis, C == 1: the axis is the contiguous one ------------------------------------------------
// A thread owns one 16-byte piece (4 consecutive w) of a row and reads its own, its left and its
// right piece straight from global memory (the neighbours' loads are the same lines 16 bytes off:
// L1 serves them, HBM sees every byte once); the 4 + 2R values it needs are then in registers.
// REFLECT at the row ends needs no extra load: x[-j] = x[j] and x[W-1+j] = x[W-1-j] lie in the own
// piece (j <= 3) or in the right / left neighbour (j = 4).  Raw sums only (the W stage of the 3-D
// smoothing is never the last one).  Adjoint: sources outside the row are zero and the weights of
// the first / last R+1 positions carry the fold-ins; they come from a per-position table in LDS so
// that every lane runs the same code.
template <int R, bool ADJ>
__global__ __launch_bounds__(256) void smooth_w1(const float* __restrict__ in, float* __restrict__ out, int64_t npieces,
                                                 int W, Taps tp) {
    __shared__ float wt[ADJ ? 128 * (2 * R + 1) : 1];
    if (ADJ) {
        for (int e = threadIdx.x; e < W * (2 * R + 1); e += 256) {
            const int p = e / (2 * R + 1), k = e % (2 * R + 1) - R;
            int src; float w;
            wt[e] = tap<R, true>(tp, p, k, W, src, w) ? w : 0.f;
        }
        __syncthreads();
    }
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npieces) return;
    const int W4 = W >> 2;
    const int pc = (int)(gid % W4), w0 = pc * 4;
    const float* row = in + (gid - pc) * 4;
    const float4 own = *reinterpret_cast<const float4*>(row + w0);
    const float4 lf = *reinterpret_cast<const float4*>(row + (pc > 0 ? w0 - 4 : w0));
    const float4 rt = *reinterpret_cast<const float4*>(row + (pc < W4 - 1 ? w0 + 4 : w0));
    // v[i] = x[w0 - 4 + i], i = 0..11
    float v[12] = {lf.x, lf.y, lf.z, lf.w, own.x, own.y, own.z, own.w, rt.x, rt.y, rt.z, rt.w};
    const bool first = pc == 0, last = pc == W4 - 1;
    if (!ADJ) {
        // mirrored values: x[-j] = x[j] (j = 1..4), x[W-1+j] = x[W-1-j]
        const float m4 = rt.x, m4b = lf.w;               // x[4] resp. x[W-5] as seen from the first / last piece
        v[3] = first ? own.y : v[3]; v[2] = first ? own.z : v[2]; v[1] = first ? own.w : v[1]; v[0] = first ? m4 : v[0];
        v[8] = last ? own.z : v[8]; v[9] = last ? own.y : v[9]; v[10] = last ? own.x : v[10]; v[11] = last ? m4b : v[11];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = first ? 0.f : v[i]; v[8 + i] = last ? 0.f : v[8 + i]; }
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int k = -R; k <= R; ++k) {
            const float w = ADJ ? wt[(w0 + j) * (2 * R + 1) + k + R] : tp.w[k + R];
            acc = fmaf(w, v[4 + j + k], acc);
        }
        o[j] = acc;
    }
    *reinterpret_cast<float4*>(out + gid * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---- generic forms: any axis length, any channel count -----------------------------------------------------------
// smooth_walk keeps a whole line of the axis in registers (L <= 64) and smooth_w1 / WALK_RAW_TW need C == 1; everything
// else used to fall back to the per-element conv_axis chain (configs[3] shape, C = 3: 3-D forward 2.36 ms).  These two
// kernels cover what is left at streaming speed, with the same fma order as the specialised ones.
//
// smooth_roll: a strided axis of ANY length as a rolling walk.  A thread owns VW contiguous floats of one line, keeps the
// (2R+1)-deep window in registers and fetches the incoming element x[reflect(p + R + 1)] U = 8 steps ahead.  REFLECT is
// done on the ADDRESS (the mirrored element is re-read; it was loaded a few steps earlier by the same thread: L1 / L2),
// so the loop has no special cases and every load is unconditional.  Adjoint: sources outside [0, L) are zero and the
// position-dependent weights (fold-ins of the mirrored taps near both ends) come from a table in LDS.
template <int R, int VW, int MODE>
__global__ __launch_bounds__(256) void smooth_roll(WalkArgs a) {
    typedef typename WalkVec<VW>::type V;
    constexpr bool ADJ = MODE == WALK_ADJ || MODE == WALK_ADJX || MODE == WALK_ADJS;
    constexpr int U = 8, NW = 2 * R + 1;
    if ((MODE == WALK_ADJ || MODE == WALK_ADJX) && a.run_if && *a.run_if == 0) return;
    extern __shared__ float roll_wt[];      // ADJ: L x NW weights
    __shared__ float red[16];
    const int L = a.L;
    const int64_t S = a.S;
    if (ADJ) {
        for (int e = threadIdx.x; e < L * NW; e += 256) {
            const int p = e / NW, k = e % NW - R;
            int src; float w;
            roll_wt[e] = tap<R, true>(a.tp, p, k, L, src, w) ? w : 0.f;
        }
        __syncthreads();
    }
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = gid < a.ncols;
    const int64_t g = ok ? gid : 0;
    const int64_t off = (g / a.inner) * L * S + (g % a.inner) * VW;
    const float* src = a.in + off;
    float m = 1.f, corr = 0.f;
    if ((MODE == WALK_WRITE && a.nblk <= 0) || MODE == WALK_ADJX || MODE == WALK_ADJS) m = a.mx[0];
    if (MODE == WALK_ADJX) corr = a.res[1] > 0.f ? a.res[0] / (m * a.res[1]) : 0.f;
    double sdot = 0.0;                  // WALK_ADJS
    __shared__ TieNote note;
    if (MODE == WALK_ADJS) note.clear();
    V zero;
#pragma unroll
    for (int c = 0; c < VW; ++c) vat<VW>(zero, c) = 0.f;
    auto fetch = [&](int q) -> V {
        int qq = q;
        if (!ADJ) { qq = qq < 0 ? -qq : qq; qq = qq >= L ? 2 * (L - 1) - qq : qq; }
        const bool inside = q >= 0 && q < L;
        qq = qq < 0 ? 0 : (qq > L - 1 ? L - 1 : qq);            // prefetch past the last needed element: any valid address
        V v = *reinterpret_cast<const V*>(src + (int64_t)qq * S);
        if (MODE == WALK_ADJX) {
            const V of = *reinterpret_cast<const V*>(a.out_fwd + off + (int64_t)qq * S);
#pragma unroll
            for (int c = 0; c < VW; ++c) vat<VW>(v, c) = vat<VW>(v, c) / m - (vat<VW>(const_cast<V&>(of), c) == 1.0f ? corr : 0.f);
        }
        if (MODE == WALK_ADJS) {        // every position inside [0, L) is fetched exactly once (no reflection in the adjoint)
            const V of = *reinterpret_cast<const V*>(a.out_fwd + off + (int64_t)qq * S);
#pragma unroll
            for (int c = 0; c < VW; ++c) {
                const float gg = vat<VW>(v, c), oo = vat<VW>(const_cast<V&>(of), c);
                if (ok && inside) {
                    sdot = fma((double)gg, (double)oo, sdot);
                    if (oo == 1.0f) note.add((long long)(off + (int64_t)qq * S + c));
                }
                vat<VW>(v, c) = gg / m;
            }
        }
        if (ADJ) v = inside ? v : zero;
        return v;
    };
    V win[NW], nxt[U];
#pragma unroll
    for (int j = 0; j < NW; ++j) win[j] = fetch(j - R);
#pragma unroll
    for (int j = 0; j < U; ++j) nxt[j] = fetch(R + 1 + j);
    if (MODE == WALK_WRITE && a.nblk > 0) {     // see smooth_walk: the preceding WALK_MAX launch's block maxima
        float v = -FLT_MAX;
        for (int i = threadIdx.x; i < a.nblk; i += 256) v = fmaxf(v, a.blockmax[i]);
        m = block_max(v, red);
        if (blockIdx.x == 0 && threadIdx.x == 0) a.mx_out[0] = m;
    }
    float vmax = -FLT_MAX;
    for (int p0 = 0; p0 < L; p0 += U) {
        V cur[U];
#pragma unroll
        for (int j = 0; j < U; ++j) cur[j] = nxt[j];
#pragma unroll
        for (int j = 0; j < U; ++j) nxt[j] = fetch(p0 + U + R + 1 + j);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int p = p0 + j;
            const bool live = p < L;                // uniform
            const float* wrow = roll_wt + (live ? p : L - 1) * NW;
            V acc = zero;
#pragma unroll
            for (int k = -R; k <= R; ++k) {
                const float w = ADJ ? wrow[k + R] : a.tp.w[k + R];
                if constexpr (VW == 1) acc = fmaf(w, win[k + R], acc);
                else {
                    V wv;
#pragma unroll
                    for (int c = 0; c < VW; ++c) vat<VW>(wv, c) = w;
                    acc = __builtin_elementwise_fma(wv, win[k + R], acc);
                }
            }
            if (MODE == WALK_MAX) {
                float mx = vat<VW>(acc, 0);
#pragma unroll
                for (int c = 1; c < VW; ++c) mx = fmaxf(mx, vat<VW>(acc, c));
                vmax = live ? fmaxf(vmax, mx) : vmax;
            } else {
                if (MODE == WALK_WRITE) {
#pragma unroll
                    for (int c = 0; c < VW; ++c) vat<VW>(acc, c) = vat<VW>(acc, c) / m;
                }
                if (ok && live) *reinterpret_cast<V*>(a.out + off + (int64_t)p * S) = acc;
            }
#pragma unroll
            for (int i = 0; i < 2 * R; ++i) win[i] = win[i + 1];
            win[2 * R] = cur[j];
        }
    }
    if (MODE == WALK_MAX && a.blockmax) {
        const float bm = block_max(ok ? vmax : -FLT_MAX, red);
        if (threadIdx.x == 0) a.blockmax[blockIdx.x] = bm;
    }
    if (MODE == WALK_ADJS) note.store(a.ties + blockIdx.x, sdot);
}

// smooth_wrow: the contiguous axis W with C interleaved channels.  A workgroup stages `rpw` rows of W*C floats in LDS
// (coalesced 16-byte loads when the span allows), then thread t computes outputs t, t + 256, ... of the span: consecutive
// lanes read consecutive LDS words for every tap (conflict-free) and store consecutive floats.  CC = channel count as a
// compile-time constant (0 = runtime).  Raw sums (W is never the last stage of the 3-D smoothing); ADJ as in smooth_w1.
// (A form with the REFLECTed halo materialised in LDS -- no reflection arithmetic in the tap loop -- was measured: faster
// at C = 1, 59 vs 64 us for the whole 3-D call at configs[1], where smooth_w1 serves anyway, and slower at C = 3, 576 vs
// 548 us at the configs[3] shape: the extra barrier and LDS cost more than the selects.)
template <int R, bool ADJ, int CC>
__global__ __launch_bounds__(256) void smooth_wrow(const float* __restrict__ in, float* __restrict__ out, int64_t nrows, int W,
                                                   int Crt, int rpw, Taps tp) {
    extern __shared__ __attribute__((aligned(16))) float wrow_lds[];
    constexpr int NW = 2 * R + 1;
    const int C = CC ? CC : Crt;
    const int WC = W * C, span = rpw * WC;
    float* wt = wrow_lds + ((span + 3) & ~3);
    if (ADJ) {
        for (int e = threadIdx.x; e < W * NW; e += 256) {
            const int p = e / NW, k = e % NW - R;
            int src; float w;
            wt[e] = tap<R, true>(tp, p, k, W, src, w) ? w : 0.f;
        }
    }
    const int64_t row0 = (int64_t)blockIdx.x * rpw;
    const int rows = (int)((nrows - row0) < rpw ? (nrows - row0) : rpw);
    const int cnt = rows * WC;
    const float* src = in + row0 * WC;
    float* dst = out + row0 * WC;
    if ((span & 3) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
        for (int e = threadIdx.x * 4; e < cnt; e += 1024) {
            // cnt % 4 == 0 when the workgroup holds all its rpw rows; the last workgroup may hold fewer
            if (e + 4 <= cnt) *reinterpret_cast<float4*>(wrow_lds + e) = *reinterpret_cast<const float4*>(src + e);
            else for (int i = e; i < cnt; ++i) wrow_lds[i] = src[i];
        }
    } else {
        for (int e = threadIdx.x; e < cnt; e += 256) wrow_lds[e] = src[e];
    }
    __syncthreads();
    int r = 0, f = threadIdx.x;                       // output e = r * WC + f
    while (f >= WC) { f -= WC; ++r; }
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const float* row = wrow_lds + r * WC;
        const int w = f / C, c = f - w * C;
        float acc = 0.f;
#pragma unroll
        for (int k = -R; k <= R; ++k) {
            int ws = w + k;
            if (!ADJ) {
                ws = ws < 0 ? -ws : ws;
                ws = ws >= W ? 2 * (W - 1) - ws : ws;
                acc = fmaf(tp.w[k + R], row[ws * C + c], acc);
            } else {
                const float wg = wt[w * NW + k + R];      // 0 where the source does not exist
                ws = ws < 0 ? 0 : (ws > W - 1 ? W - 1 : ws);
                acc = fmaf(wg, row[ws * C + c], acc);
            }
        }
        dst[e] = acc;
        f += 256;
        while (f >= WC) { f -= WC; ++r; }
    }
}

// smooth_tw_plane: the T stage and the W stage of the 3-D smoothing in ONE pass over the tensor, any channel count
// 1..4 (W*C a multiple of 4).  The rows of one (b, h) are consecutive in memory (t-major), so a workgroup that stages the
// whole T x W*C plane of a (b, h) in LDS has both stencils' neighbours at hand: forward = T stencil from buffer A (plane
// + R REFLECTed rows above and below) into buffer B (rows + a REFLECTed column halo either side), then the W stencil from
// B to global memory; adjoint = W^T from A (zero column halo, border weights from a table) into B (zero row halo), then
// T^T to global.  With the halos materialised every tap is a plain LDS read at a fixed offset, and everything moves as
// float4: a thread item is four consecutive floats of a row -- the T stencil is 2R+1 ds_read_b128 at row offsets, the W
// stencil reads the 4 + 2 R C floats it needs as 2 ceil(R C / 4) + 1 aligned float4s and indexes them statically (C is a
// template parameter).  Same fma order per output as smooth_walk / smooth_roll followed by smooth_w1 / smooth_wrow
// (forward: bit-identical).  One read and one write of the tensor instead of two each.
// (The first version of this kernel walked single floats through three LDS loops and was VALU-bound: forward +5 %,
// adjoint -55 % against the separate stages at the configs[3] shape.)
template <int R, bool ADJ, int CC>
__global__ __launch_bounds__(256) void smooth_tw_plane(const float* __restrict__ in, float* __restrict__ out, int T, int W,
                                                       Taps tp) {
    extern __shared__ __attribute__((aligned(16))) float twp[];
    constexpr int NW = 2 * R + 1, C = CC;
    constexpr int NQ = (R * C + 3) / 4, HP = 4 * NQ;        // halo columns, padded to whole float4s
    const int WC = W * C, Q = WC >> 2, pitch = WC + 2 * HP, P = T * WC, P4 = T * Q;
    // forward: A = (T + 2R) x WC, B = T x pitch;  adjoint: A = T x pitch, B = (T + 2R) x WC
    const int sizeA = ADJ ? T * pitch : (T + 2 * R) * WC;
    const int sizeB = ADJ ? (T + 2 * R) * WC : T * pitch;
    float* A = twp;
    float* Bf = twp + sizeA;
    float* wtT = Bf + sizeB;
    float* wtW = wtT + T * NW;
    const int tid = threadIdx.x;
    if (ADJ) {
        for (int e = tid; e < T * NW; e += 256) {
            int src; float w;
            wtT[e] = tap<R, true>(tp, e / NW, e % NW - R, T, src, w) ? w : 0.f;
        }
        for (int e = tid; e < W * NW; e += 256) {
            int src; float w;
            wtW[e] = tap<R, true>(tp, e / NW, e % NW - R, W, src, w) ? w : 0.f;
        }
    }
    const float* src = in + (int64_t)blockIdx.x * P;
    float* dst = out + (int64_t)blockIdx.x * P;
    int t0 = 0, q0 = tid;                                   // this thread's first item (t0, q0); items step by 256
    while (q0 >= Q) { q0 -= Q; ++t0; }
    // ---- stage 0: the plane into A (interior), coalesced
    {
        int t = t0, q = q0;
        for (int i = tid; i < P4; i += 256) {
            const float4 x = reinterpret_cast<const float4*>(src)[i];
            *reinterpret_cast<float4*>(ADJ ? A + t * pitch + HP + 4 * q : A + (t + R) * WC + 4 * q) = x;
            q += 256;
            while (q >= Q) { q -= Q; ++t; }
        }
    }
    __syncthreads();
    // ---- halo of A: forward = REFLECTed rows t = -R..-1 and T..T+R-1; adjoint = zero columns
    if (!ADJ) {
        for (int i = tid; i < 2 * R * Q; i += 256) {
            const int hr = i / Q, q = i - hr * Q;                   // halo row 0..2R-1
            const int t = hr < R ? hr - R : T + (hr - R);           // position
            const int ts = t < 0 ? -t : 2 * (T - 1) - t;            // reflect
            *reinterpret_cast<float4*>(A + (t + R) * WC + 4 * q) = *reinterpret_cast<const float4*>(A + (ts + R) * WC + 4 * q);
        }
    } else {
        for (int i = tid; i < T * 2 * NQ; i += 256) {
            const int t = i / (2 * NQ), j = i - t * 2 * NQ;
            *reinterpret_cast<float4*>(A + t * pitch + (j < NQ ? 4 * j : WC + 4 * j)) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    // the W stencil of four consecutive outputs from a row with halo: x points at the first of them
    auto wconv = [&](const float* x, int f, float4& o) {
        float v[4 * (2 * NQ + 1)];
#pragma unroll
        for (int j = 0; j < 2 * NQ + 1; ++j) {
            const float4 p = *reinterpret_cast<const float4*>(x + 4 * (j - NQ));
            v[4 * j] = p.x; v[4 * j + 1] = p.y; v[4 * j + 2] = p.z; v[4 * j + 3] = p.w;
        }
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = 0.f;
            const float* wr = wtW + ((f + j) / C) * NW;     // adjoint only
#pragma unroll
            for (int k = -R; k <= R; ++k) acc = fmaf(ADJ ? wr[k + R] : tp.w[k + R], v[HP + j + k * C], acc);
            r[j] = acc;
        }
        o = make_float4(r[0], r[1], r[2], r[3]);
    };
    // the T stencil of a float4 column item: x points at row t of a (T + 2R)-row buffer without column halo
    auto tconv = [&](const float* x, int t, float4& o) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = -R; k <= R; ++k) {
            const float w = ADJ ? wtT[t * NW + k + R] : tp.w[k + R];
            const float4 p = *reinterpret_cast<const float4*>(x + k * WC);
            acc.x = fmaf(w, p.x, acc.x); acc.y = fmaf(w, p.y, acc.y); acc.z = fmaf(w, p.z, acc.z); acc.w = fmaf(w, p.w, acc.w);
        }
        o = acc;
    };
    // ---- stage 1: first stencil, A -> B interior
    {
        int t = t0, q = q0;
        for (int i = tid; i < P4; i += 256) {
            float4 o;
            if (!ADJ) {
                tconv(A + (t + R) * WC + 4 * q, t, o);
                *reinterpret_cast<float4*>(Bf + t * pitch + HP + 4 * q) = o;
            } else {
                wconv(A + t * pitch + HP + 4 * q, 4 * q, o);
                *reinterpret_cast<float4*>(Bf + (t + R) * WC + 4 * q) = o;
            }
            q += 256;
            while (q >= Q) { q -= Q; ++t; }
        }
    }
    __syncthreads();
    // ---- halo of B: forward = REFLECTed columns; adjoint = zero rows
    if (!ADJ) {
        for (int e = tid; e < T * 2 * R * C; e += 256) {
            const int t = e / (2 * R * C), j = e - t * 2 * R * C;
            const int hw = j / C, c = j - hw * C;                   // halo w index 0..2R-1, channel
            const int w = hw < R ? hw - R : W + (hw - R);
            const int ws = w < 0 ? -w : 2 * (W - 1) - w;
            Bf[t * pitch + HP + w * C + c] = Bf[t * pitch + HP + ws * C + c];
        }
    } else {
        for (int i = tid; i < 2 * R * Q; i += 256) {
            const int hr = i / Q, q = i - hr * Q;
            *reinterpret_cast<float4*>(Bf + (hr < R ? hr : T + hr) * WC + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    // ---- stage 2: second stencil, B -> global
    {
        int t = t0, q = q0;
        for (int i = tid; i < P4; i += 256) {
            float4 o;
            if (!ADJ) wconv(Bf + t * pitch + HP + 4 * q, 4 * q, o);
            else tconv(Bf + (t + R) * WC + 4 * q, t, o);
            reinterpret_cast<float4*>(dst)[i] = o;
            q += 256;
            while (q >= Q) { q -= Q; ++t; }
        }
    }
}

static size_t tw_plane_lds(int T, int W, int C, int radius, bool adjoint) {
    const size_t WC = (size_t)W * C, pitch = WC + 2 * 4 * (((size_t)radius * C + 3) / 4);
    const size_t a = adjoint ? T * pitch : (T + 2 * (size_t)radius) * WC;
    const size_t b = adjoint ? (T + 2 * (size_t)radius) * WC : T * pitch;
    return (a + b + (adjoint ? (size_t)(T + W) * (2 * radius + 1) : 0)) * sizeof(float);
}

static bool tw_plane_eligible(int T, int W, int C, int radius, bool adjoint, const void* p0, const void* p1) {
    if (!opt(OPT_SMOOTH_FUSED_TW)) return false;
    // forward only: the adjoint form's border-weight table reads made it SLOWER than the separate W^T and T^T stages
    // (configs[3] shape: 741 vs 682 us for the whole 3-D adjoint; forward: 456 vs 545 us)
    if (adjoint) return false;
    return (radius == 3 || radius == 4) && C >= 1 && C <= 4 && ((W * C) & 3) == 0 && T > radius && W > radius &&
           (((uintptr_t)p0 | (uintptr_t)p1) & 15) == 0 && tw_plane_lds(T, W, C, radius, adjoint) <= 156 * 1024;
}

static int launch_tw_plane(const float* in, float* out, int64_t n, int T, int W, int C, int radius, bool adjoint, const Taps& tp,
                           hipStream_t st) {
    const size_t lds = tw_plane_lds(T, W, C, radius, adjoint);
    const dim3 grid((unsigned)(n / ((int64_t)T * W * C)));
    // (the dynamic-LDS limit is raised on every launch that needs it: a per-process flag would be wrong for a second device)
#define KCCOT_TWP(RR, CCC)                                                                                             \
    do {                                                                                                               \
        if (lds > 64 * 1024 &&                                                                                         \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&smooth_tw_plane<RR, false, CCC>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                 \
            return fail(KCCOT_EUNSUPPORTED, "smooth_tw_plane: cannot raise the dynamic LDS limit");                    \
        hipLaunchKernelGGL((smooth_tw_plane<RR, false, CCC>), grid, dim3(256), lds, st, in, out, T, W, tp);            \
    } while (0)
#define KCCOT_TWP_C(RR)                                                                 \
    switch (C) { case 1: KCCOT_TWP(RR, 1); break; case 2: KCCOT_TWP(RR, 2); break;      \
                 case 3: KCCOT_TWP(RR, 3); break; default: KCCOT_TWP(RR, 4); break; }
    if (adjoint) return fail(KCCOT_EUNSUPPORTED, "smooth_tw_plane: the adjoint runs as separate W and T stages");
    if (radius == 3) { KCCOT_TWP_C(3) } else { KCCOT_TWP_C(4) }
#undef KCCOT_TWP_C
#undef KCCOT_TWP
    return launch_status("smooth_tw_plane");
}

// ---- smooth_fused3: T, W and H stage of the 3-D smoothing in ONE pass (round 4) ---------------------------------------
// The chain above moves the tensor five times (T+W: read + write, H maxima: read, H write: read + write) and seven times where
// the (b, h) plane does not fit smooth_tw_plane's LDS budget (configs[4]: T x W*C = 48 x 384).  Here a workgroup owns a column
// tile (wt columns of W, all of T) of one sample and WALKS along H: every plane piece goes global -> registers (fetched one
// plane ahead) -> LDS A (rows REFLECTed into a row halo while they are written; the R columns either side come from the
// neighbouring tile, or REFLECTed at the tensor border, as 4-byte loads) -> T stencil -> LDS B -> W stencil -> a (2R+1)-deep
// REGISTER window per owned float4 -> H stencil -> out.  Two LDS barriers per plane, no intermediate tensor.  The planes before
// h0 and behind h1 are the REFLECTed (or neighbouring) ones, loaded and smoothed again: (hseg + 2R) / hseg of the reads, and
// (wt + 2R) / wt from the column halo -- 1.05 x 1.19 at configs[4], where nothing is segmented along H.
// The tensor maximum costs one more READ of the input (mode 0: same arithmetic, per-workgroup maxima, no store), the writing
// pass (mode 1) recomputes s and stores s / max, reducing the maxima itself: three tensor moves + halo instead of five / seven.
// Same fma order per output as the chain (k = -R .. R ascending on every axis; T, then W, then H): bit-identical results.
constexpr int F3_NH = 3;            // halo floats a thread may own per plane
constexpr int F3A_NH = 3;           // the same for smooth_fused3_adj (they live in its register window)
struct Fused3Args {
    const float* in;
    float* out;            // null in mode 0
    float* blockmax;       // mode 0: one maximum per workgroup;  mode 1 with nblk > 0: read
    const float* mx;       // mode 1 with nblk == 0: the tensor maximum
    float* mx_out;         // mode 1 with nblk > 0: the tensor maximum is stored here
    int nblk, mode;        // mode: 0 maxima only, 1 write s / max, 2 write s
    int H, T, W, wt, ntw, hseg, nseg;
    Taps tp;
};

// Registers decide the occupancy, and the occupancy the speed: the SQ counters of the first version said VALU 36 % busy at 1.5 waves
// per SIMD (180 registers x 384 threads = ONE workgroup per CU) -- bound by LDS round trips and barriers.  Holding that body to 168
// registers with amdgpu_waves_per_eu made it SLOWER (4.61 -> 6.95 ms at configs[4]: the spill reloads share vmcnt with the plane
// prefetch and expose its latency every step); buffer descriptors instead of 64-bit per-lane addresses freed the registers
// without a spill (158), and two workgroups fit.
template <int R, int CC, int NI>
__global__ __launch_bounds__(512) void smooth_fused3(Fused3Args a) {
    typedef WalkVec<4>::type V4;        // (arrays of HIP's float4 struct stayed in scratch memory; ext-vectors do not)
    extern __shared__ __attribute__((aligned(16))) float f3lds[];
    __shared__ float red[16];
    constexpr int C = CC, NW = 2 * R + 1, RC = R * C, NQ = (RC + 3) / 4, HP = 4 * NQ;
    const int NT = blockDim.x, tid = threadIdx.x;
    const int T = a.T, WC = a.W * C, wtc = a.wt * C, Q = wtc >> 2, pitch = wtc + 2 * HP, QA = pitch >> 2;
    const float inv_qa = 1.0f / (float)QA;
    float* A = f3lds;                             // (T + 2R) x pitch: the plane piece with row and column halo
    float* Bf = f3lds + (T + 2 * R) * pitch;      // T x pitch: T-smoothed
    // four floats behind B: the target of writes a thread has no item for (one slot for all: same-address LDS writes do not
    // serialise -- a slot per thread measured no faster and its 8 KB pushed the configs[2..3] tiles over the LDS budget)
    const int dummy = (2 * T + 2 * R) * pitch;
    // workgroups are dealt to the eight XCDs round-robin: give each XCD a CONTIGUOUS run of tiles, so that the column tiles of a
    // plane and the H segments of a sample -- which read each other's halo -- share an L2
    int blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
    const int tw = blk % a.ntw; blk /= a.ntw;
    const int seg = blk % a.nseg, b = blk / a.nseg;
    const int w0 = tw * a.wt, h0 = seg * a.hseg, h1 = min(h0 + a.hseg, a.H);
    const int64_t P = (int64_t)T * WC;
    // the sample through buffer descriptors: one 32-bit byte offset per lane (+ the plane's in an SGPR) instead of 64-bit
    // per-lane addresses -- a dozen registers and their adds; offsets are relative to the SAMPLE (the left halo lies in
    // front of the tile), which the plan keeps below 2 GB
    typedef unsigned int U4 __attribute__((ext_vector_type(4)));
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + (int64_t)b * a.H * P), 0, 0xFFFFFFFFu, 0x00020000);
    const auto rout = __builtin_amdgcn_make_buffer_rsrc((a.out ? a.out : const_cast<float*>(a.in)) + (int64_t)b * a.H * P, 0, 0xFFFFFFFFu, 0x00020000);
    const int plane_bytes = (int)(P * 4);

    // mirror row of row t in the (T + 2R)-row buffer, or -1: rows 1..R also serve -1..-R, rows T-1-R..T-2 serve T..T-1+R
    auto mirror = [&](int t) { return (t >= 1 && t <= R) ? R - t : ((t >= T - 1 - R && t <= T - 2) ? 2 * (T - 1) - t + R : -1); };
    // owned float4 items of the interior (fixed for the whole walk)
    int goff[NI], amir[NI], boff[NI];         // (the item's slot in A is boff + R * pitch)
    bool ok[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        int i = tid + NT * n;
        ok[n] = i < T * Q;
        i = ok[n] ? i : T * Q - 1;
        const int t = i / Q, q = i - t * Q;
        goff[n] = (t * WC + w0 * C + 4 * q) * 4;          // bytes within a plane of the sample
        const int mr = mirror(t);
        amir[n] = (ok[n] && mr >= 0) ? mr * pitch + HP + 4 * q : dummy;
        boff[n] = t * pitch + HP + 4 * q;
    }
    // owned halo floats: R*C either side of every row
    int hgo[F3_NH], hao[F3_NH], hmi[F3_NH];
    const int nhalo = T * 2 * RC;
#pragma unroll
    for (int n = 0; n < F3_NH; ++n) {
        int e = tid + NT * n;
        const bool hok = e < nhalo;
        e = hok ? e : nhalo - 1;
        const int t = e / (2 * RC), j = e - t * 2 * RC;
        const int side = j >= RC, jj = j - side * RC;
        const int w = (side ? w0 + a.wt : w0 - R) + jj / C, c = jj % C;
        hgo[n] = (t * WC + reflect(w, a.W) * C + c) * 4;
        const int col = side ? HP + wtc + jj : HP - RC + jj;
        hao[n] = hok ? (t + R) * pitch + col : dummy;
        const int mr = mirror(t);
        hmi[n] = (hok && mr >= 0) ? mr * pitch + col : dummy;
    }
    float m = 1.f;
    if (a.mode == 1) {
        if (a.nblk > 0) {           // the maxima of the preceding mode-0 launch, reduced by every workgroup itself
            float v = -FLT_MAX;
            for (int i = tid; i < a.nblk; i += NT) v = fmaxf(v, a.blockmax[i]);
            m = block_max(v, red);
            if (blockIdx.x == 0 && tid == 0) a.mx_out[0] = m;
        } else {
            m = a.mx[0];
        }
    }
    // 1 / m refined once (what the IEEE division's sequence starts from); the short division is taken for ordinary maxima only
    float rcp_m = __builtin_amdgcn_rcpf(m);
    rcp_m = fmaf(fmaf(-m, rcp_m, 1.f), rcp_m, rcp_m);
    const bool fast_div = fabsf(m) > 0x1p-40f && fabsf(m) < 0x1p40f;
    V4 win[NI][NW];
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int j = 0; j < NW; ++j) win[n][j] = V4{0.f, 0.f, 0.f, 0.f};
    V4 x[NI];
    float hx[F3_NH];
    float vmax = -FLT_MAX;
    const int hlo = h0 - R, hhi = h1 + R;
#ifdef KCCOT_DIAG
    const int abl = a.mode >> 8;        // timing ablations of tools/micro/f3_ablate.sh (results wrong)
    a.mode &= 255;
#else
    constexpr int abl = 0;
#endif
    {
        const int so = reflect(hlo, a.H) * plane_bytes;
#pragma unroll
        for (int n = 0; n < NI; ++n) x[n] = __builtin_bit_cast(V4, (U4)__builtin_amdgcn_raw_buffer_load_b128(rin, goff[n], so, 0));
#pragma unroll
        for (int n = 0; n < F3_NH; ++n) hx[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, hgo[n], so, 0));
    }
    // The walk, unrolled by the depth of the window so that the window's slots are static registers: the plane of trip s of a
    // round lands in slot s, the H stencil reads the slots in the order s + 1, .., s + NW (mod NW) = oldest .. newest.
    auto step = [&](auto slot, const int hp) {
        {
            constexpr int s = decltype(slot)::value;
            // ---- registers -> A (the previous plane's T stage finished in front of that plane's second barrier); items a
            // thread does not own go to a dummy slot behind the buffers: no branches
            if (!(abl & 128)) {
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                *reinterpret_cast<V4*>(f3lds + (ok[n] ? boff[n] + R * pitch : dummy)) = x[n];
                *reinterpret_cast<V4*>(f3lds + amir[n]) = x[n];
            }
#pragma unroll
            for (int n = 0; n < F3_NH; ++n) {
                f3lds[hao[n]] = hx[n];
                f3lds[hmi[n]] = hx[n];
            }
            }
            if (!(abl & 1)) {   // the next plane's piece, in flight across this plane's stencils
                const int so = reflect(min(hp + 1, hhi - 1), a.H) * plane_bytes;
#pragma unroll
                for (int n = 0; n < NI; ++n) x[n] = __builtin_bit_cast(V4, (U4)__builtin_amdgcn_raw_buffer_load_b128(rin, goff[n], so, 0));
#pragma unroll
                for (int n = 0; n < F3_NH; ++n) hx[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, hgo[n], so, 0));
            }
            if (!(abl & 512)) lds_barrier();
            // ---- T stencil: A -> B over the whole pitch (interior + column halo).  A thread takes a strip of LT rows of one float4
            // column: LT + 2R rows read once, LT outputs -- 2.5 LDS reads per output instead of 7 (the walk is bound by LDS
            // bandwidth before VALU issue: ~18 b128 accesses per output float4 against ~60 VALU instructions).  Same fma order.
            if (!(abl & 2)) {
                constexpr int LT = 4;
                const int nstrip = (T + LT - 1) / LT;
                for (int i = tid; i < nstrip * QA; i += NT) {
                    // (strip = i / QA by a float multiply, exact for these sizes: stepping by repeated subtraction as
                    // smooth_tw_plane does costs NT / QA trips per item here, QA being a dozen -- that was 2/3 of the first version)
                    const int st = (int)(((float)i + 0.5f) * inv_qa), q = i - st * QA, t0 = st * LT;
                    const int c0 = t0 * pitch + 4 * q;                      // row t0 - R of the plane = row t0 of A
                    const int last = (T + 2 * R - 1 - t0) * pitch;          // the strip may overhang the plane: rows clamped
                    V4 r[LT + 2 * R];
#pragma unroll
                    for (int j = 0; j < LT + 2 * R; ++j) r[j] = *reinterpret_cast<const V4*>(A + c0 + min(j * pitch, last));
#pragma unroll
                    for (int jo = 0; jo < LT; ++jo) {
                        V4 acc = V4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = -R; k <= R; ++k) {
                            const float w = a.tp.w[k + R];
                            acc = __builtin_elementwise_fma(V4{w, w, w, w}, r[jo + k + R], acc);
                        }
                        *reinterpret_cast<V4*>(f3lds + (t0 + jo < T ? (T + 2 * R + t0 + jo) * pitch + 4 * q : dummy)) = acc;
                    }
                }
            }
            if (!(abl & 512)) lds_barrier();
            // ---- W stencil: B -> the window's slot s
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                float v[4 * (2 * NQ + 1)];
#pragma unroll
                for (int j = 0; j < 2 * NQ + 1; ++j) {
                    const V4 p = (abl & 4) ? x[n] : *reinterpret_cast<const V4*>(Bf + boff[n] + 4 * (j - NQ));
                    v[4 * j] = p[0]; v[4 * j + 1] = p[1]; v[4 * j + 2] = p[2]; v[4 * j + 3] = p[3];
                }
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = -R; k <= R; ++k) acc = fmaf(a.tp.w[k + R], v[HP + j + k * C], acc);
                    r[j] = acc;
                }
                win[n][s] = V4{r[0], r[1], r[2], r[3]};
            }
            const int hout = hp - R;
            if (hout >= h0 && !(abl & 256)) {               // else the window is not full yet (uniform)
                // ---- H stencil over the register window, emit plane hout
#pragma unroll
                for (int n = 0; n < NI; ++n) {
                    V4 acc = V4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const float w = a.tp.w[j];
                        acc = __builtin_elementwise_fma(V4{w, w, w, w}, win[n][(s + 1 + j) % NW], acc);
                    }
                    if (a.mode == 0) {
                        const float mx4 = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
                        vmax = ok[n] ? fmaxf(vmax, mx4) : vmax;
                    } else {
                        if (a.mode == 1 && !(abl & 8)) {
                            if (fast_div) {
                                // s / m with the loop-invariant divisor: the instruction sequence of the IEEE division (refined
                                // reciprocal, quotient, two residual corrections) without its scaling steps, which are the
                                // identity unless the quotient is below 2^-100 (there: within 2^-149 absolute)
                                const V4 nm = V4{-m, -m, -m, -m}, rr = V4{rcp_m, rcp_m, rcp_m, rcp_m};
                                V4 q = acc * rr;
                                q = __builtin_elementwise_fma(__builtin_elementwise_fma(nm, q, acc), rr, q);
                                q = __builtin_elementwise_fma(__builtin_elementwise_fma(nm, q, acc), rr, q);
                                acc = q;
                            } else {
                                acc[0] = acc[0] / m; acc[1] = acc[1] / m; acc[2] = acc[2] / m; acc[3] = acc[3] / m;
                            }
                        }
                        if (ok[n] && (!(abl & 16) || acc[0] == 123.f))
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(U4, acc), rout, goff[n], hout * plane_bytes, 0);
                    }
                }
            }
        }
    };
#define KCCOT_F3_STEP(S) if constexpr (S < NW) { if (hp0 + S < hhi) step(std::integral_constant<int, S>{}, hp0 + S); }
    for (int hp0 = hlo; hp0 < hhi; hp0 += NW) {
        KCCOT_F3_STEP(0) KCCOT_F3_STEP(1) KCCOT_F3_STEP(2) KCCOT_F3_STEP(3) KCCOT_F3_STEP(4)
        KCCOT_F3_STEP(5) KCCOT_F3_STEP(6) KCCOT_F3_STEP(7) KCCOT_F3_STEP(8)
    }
#undef KCCOT_F3_STEP
    if (a.mode == 0) {
        const float bm = block_max(vmax, red);
        if (tid == 0) a.blockmax[blockIdx.x] = bm;
    }
}


struct Fused3Plan { int wt, hseg, ni, nt; size_t lds; int64_t grid; bool ok; };

// tile choice: the cheapest (column halo) x (plane halo) overhead among the power-of-two cuts of W and H that still fills the
// CUs; NI = float4 items per thread <= 4 at <= 512 threads
static Fused3Plan fused3_plan(int B, int H, int T, int W, int C, int radius, const void* p0, const void* p1, bool adj = false) {
    Fused3Plan best{};
    const int mode = opt(OPT_SMOOTH_FUSED3);
    if (!mode || !(radius == 3 || radius == 4) || !(C == 1 || C == 3)) return best;
    if (adj && radius != 3) return best;      // (the nine-deep window of radius 4 does not fit the register file next to the halo's)
    // where it wins (same-box A/B at the BASELINE frame shapes and batch sizes in between, profiles/r4_ab_smooth_fused3.txt): three
    // channels from ~20 M elements on (configs[2..4]: -21 / -29 / -29 %); one channel never -- there the chain's T and W stages are
    // one register-only launch (WALK_RAW_TW) and the tensors are small (configs[1]: 66 against 41 us).  2 = wherever it can run.
    // The adjoint (smooth_fused3_adj) replaces seven moves, not five, and wins from ~3.5 M elements on at either channel count
    // (profiles/r4_ab_smooth_fused3.txt: configs[1] 65.6 -> 54.2 us, configs[2..4] -49 / -47 / -33 %).
    const int64_t numel = (int64_t)B * H * T * W * C;
    if (mode == 1 && (adj ? numel < 3500000 : !(C == 3 && numel >= 20000000))) return best;
    if (T < 2 * radius + 2 || H < radius + 2 || W < radius + 2 || ((W * C) & 3) || (((uintptr_t)p0 | (uintptr_t)p1) & 15)) return best;
    if ((int64_t)H * T * W * C * 4 >= ((int64_t)1 << 31)) return best;         // byte offsets within a sample are 32-bit
    double best_cost = 1e30;
    int force_wt = 0, force_hs = 0;
#ifdef KCCOT_DIAG
    if (const char* e = getenv("KCCOT_F3_PLAN")) sscanf(e, "%d,%d", &force_wt, &force_hs);     // tile experiments (diag twin only)
#endif
    for (int wt = W; wt >= 8; wt >>= 1) {
        if (W % wt || ((wt * C) & 3)) break;
        if (force_wt && wt != force_wt) continue;
        const int items = T * (wt * C / 4);
        int ni = 0, nt = 0;
        for (int n = 1; n <= (adj ? 3 : (radius == 4 ? 3 : 4)) && !ni; ++n) {      // (more items would spill)
            const int th = ((items + n - 1) / n + 63) / 64 * 64;
            if (th <= 512) { ni = n; nt = th < 128 ? 128 : th; }
            if (adj && ni == 3) nt = 512;       // (two waves per SIMD instead of 1.5, a quarter of the lanes idle: configs[4] 4.67 -> 4.38 ms)
#ifdef KCCOT_DIAG
            if (ni && getenv("KCCOT_F3_NT")) nt = atoi(getenv("KCCOT_F3_NT"));     // occupancy experiments (diag twin only)
#endif
        }
        if (!ni || T * 2 * radius * C > (adj ? F3A_NH : F3_NH) * nt) continue;
        const int hp = 4 * ((radius * C + 3) / 4), pitch = wt * C + 2 * hp;
        const size_t lds = adj ? ((size_t)T * pitch + (size_t)(T + 4 * radius) * wt * C + 4) * sizeof(float)
                               : ((size_t)(2 * T + 2 * radius) * pitch + 4) * sizeof(float);
        if (lds > 64 * 1024) continue;
        for (int hs = H; hs >= 8; hs = (hs + 1) / 2) {
            const int64_t grid = (int64_t)B * ((H + hs - 1) / hs) * (W / wt);
            const double waves = (double)grid * nt / (256.0 * 512.0);      // in units of 512 threads per CU (measured: one such
                                                                              // round with less halo beats two with more)
            const double over = (1.0 + 2.0 * radius / wt) * (1.0 + 2.0 * radius / hs);
            // below one full round the chip idles: price that as if the work were spread over the workgroups there are
            double cost = over * (waves >= 1.0 ? 1.0 : 1.0 / waves);
            if (force_hs) cost = hs == force_hs ? 0.0 : 1e29;
            if (cost < best_cost && grid <= 0x7fffffff) { best_cost = cost; best = Fused3Plan{wt, hs, ni, nt, lds, grid, true}; }
            if (hs == 8) break;
        }
    }
    return best;
}

static int launch_fused3(const Fused3Plan& pl, Fused3Args fa, int radius, int C, hipStream_t st) {
#define KCCOT_F3(RR, CCC, NN) hipLaunchKernelGGL((smooth_fused3<RR, CCC, NN>), dim3((unsigned)pl.grid), dim3(pl.nt), pl.lds, st, fa)
#define KCCOT_F3_N(RR, CCC)                                                                           \
    switch (pl.ni) { case 1: KCCOT_F3(RR, CCC, 1); break; case 2: KCCOT_F3(RR, CCC, 2); break;        \
                     case 3: KCCOT_F3(RR, CCC, 3); break; default: KCCOT_F3(RR, CCC, 4); break; }
    if (radius == 3) { if (C == 1) { KCCOT_F3_N(3, 1) } else { KCCOT_F3_N(3, 3) } }
    else { if (C == 1) { KCCOT_F3_N(4, 1) } else { KCCOT_F3_N(4, 3) } }
#undef KCCOT_F3_N
#undef KCCOT_F3
    return launch_status("smooth_fused3");
}

// ---- smooth_fused3_adj: the adjoint of the three stages in ONE pass (round 4) ------------------------------------------------
// Backward of gaussian_convolution3D where the statistics fold applies (smooth_bwd_fold): the chain reads gout and the forward
// output, writes H^T, reads it, writes W^T, reads it, writes T^T -- seven tensor moves; here x = gout / max goes straight into a
// (2R+1)-deep register window while the workgroup walks along H (interior float4 items AND the R columns either side, which the
// W stage needs H-smoothed), H^T with the border-folded weights of the output plane (uniform per step, planes outside the tensor
// are zero), then through LDS: W^T over the rows of the plane piece, T^T over its columns, din stored once: read gout, read out
// (owned planes only, for the two sums of the normalisation's adjoint), write din = three moves + halo.
// Borders: the adjoint of "REFLECT-pad, then correlate" is "correlate the zero-extended line, then fold the pad back":
//   din[q] = y[q] + [1 <= q <= R] y[-q] + [L-1-R <= q <= L-2] y[2(L-1)-q],   y[p] = sum_k w[k] x[p+k], x = 0 outside [0, L)
// W: the tiles at the tensor border add the folded columns in a short pass of their own (R*C floats per row and side);
// T: the items of rows 1..R and L-1-R..L-2 run the stencil a second time at their mirror row of the zero-extended buffer.
// Algebraically the chain's position-dependent weights; the sums are associated differently at the folded positions (the
// chain adds the weights first), so agreement with the chain is to rounding (1e-6 of max|din|), not bit for bit.
// The statistics (sum of gout * out in fp64, positions of out == 1) are gathered from the owned elements while they stream by,
// one TieRec per workgroup, as smooth_walk's WALK_ADJS does; maxnorm_bwd_fixup consumes them unchanged.
struct TieNote8 {                   // TieNote for workgroups of up to 512 threads
    double part[8]; long long idx[TIE_PER_WG]; int n;
    __device__ __forceinline__ void clear() { if (threadIdx.x == 0) n = 0; __syncthreads(); }
    __device__ __forceinline__ void add(long long e) { const int slot = atomicAdd(&n, 1); if (slot < TIE_PER_WG) idx[slot] = e; }
    __device__ __forceinline__ void store(TieRec* rec, double dot) {
        dot = wave_sum_d(dot);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = dot;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int nw = (blockDim.x + 63) >> 6;
            double d = 0.0;
            for (int w = 0; w < nw; ++w) d += part[w];
            rec->dot = d; rec->ties = n; rec->pad = 0;
            for (int i = 0; i < TIE_PER_WG; ++i) rec->idx[i] = idx[i];
        }
    }
};
struct Fused3AdjArgs {
    const float* gout;
    const float* out_fwd;
    float* din;
    const float* mx;
    TieRec* ties;          // one record per workgroup (not XM)
    const float* res;      // XM: {sum gout * out, #(out == 1)} over the whole batch
    const int* run_if;     // XM: non-null -> the launch does nothing unless *run_if != 0 (the dense fallback behind the fix-up)
    int H, T, W, wt, ntw, hseg, nseg;
    Taps tp;
};

// XM: the batch-sharded caller's form (KCCOT_SMOOTH_EXTERNAL_STATS): the two sums are known (all-reduced over the ranks), so the
// normalisation's adjoint is applied while the planes are loaded, x = gout / max - corr [out == 1] -- for the halo columns too --
// and nothing is gathered.
template <int R, int CC, int NI, bool XM>
__global__ __launch_bounds__(512) void smooth_fused3_adj(Fused3AdjArgs a) {
    typedef WalkVec<4>::type V4;
    extern __shared__ __attribute__((aligned(16))) float f3lds[];
    __shared__ TieNote8 note;
    constexpr int C = CC, NW = 2 * R + 1, RC = R * C, NQ = (RC + 3) / 4, HP = 4 * NQ, NH = F3A_NH;
    const int NT = blockDim.x, tid = threadIdx.x;
    const int T = a.T, WC = a.W * C, wtc = a.wt * C, Q = wtc >> 2, pitch = wtc + 2 * HP;
    float* A = f3lds;                             // T x pitch: the H^T-smoothed plane piece with its column halo
    float* Bz = f3lds + T * pitch;                // (T + 4R) x wtc: W^T-smoothed, 2R zero rows above and below
    const int dummy = T * pitch + (T + 4 * R) * wtc;      // four floats behind: the target of writes a thread has no item for
    // workgroups are dealt to the eight XCDs round-robin: give each XCD a CONTIGUOUS run of tiles, so that the column tiles of a
    // plane and the H segments of a sample -- which read each other's halo -- share an L2
    int blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
    const int tw = blk % a.ntw; blk /= a.ntw;
    const int seg = blk % a.nseg, b = blk / a.nseg;
    const int w0 = tw * a.wt, h0 = seg * a.hseg, h1 = min(h0 + a.hseg, a.H);
    const bool left_edge = w0 == 0, right_edge = w0 + a.wt == a.W;
    const int64_t P = (int64_t)T * WC;
    const int64_t base = (int64_t)b * a.H * P;         // the sample, through buffer descriptors (see smooth_fused3)
    typedef unsigned int U4 __attribute__((ext_vector_type(4)));
    const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gout + base), 0, 0xFFFFFFFFu, 0x00020000);
    const auto ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.out_fwd + base), 0, 0xFFFFFFFFu, 0x00020000);
    const auto rd = __builtin_amdgcn_make_buffer_rsrc(a.din + base, 0, 0xFFFFFFFFu, 0x00020000);
    const int plane_bytes = (int)(P * 4);
    if (XM && a.run_if && *a.run_if == 0) return;
    const float m = a.mx[0];
    float rcp_m = __builtin_amdgcn_rcpf(m);
    rcp_m = fmaf(fmaf(-m, rcp_m, 1.f), rcp_m, rcp_m);
    const bool fast_div = fabsf(m) > 0x1p-40f && fabsf(m) < 0x1p40f;
    for (int i = tid; i < 2 * R * wtc; i += NT) { Bz[i] = 0.f; Bz[(T + 2 * R) * wtc + i] = 0.f; }
    if (!XM) note.clear();
    const float corr = XM ? (a.res[1] > 0.f ? a.res[0] / (m * a.res[1]) : 0.f) : 0.f;

    int goff[NI], ard[NI], brd[NI], bmir[NI];     // ard / brd: the item's position in A / Bz (writes of items a thread does not
                                                  // own are redirected to the dummy slot where they happen)
    bool ok[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        int i = tid + NT * n;
        ok[n] = i < T * Q;
        i = ok[n] ? i : T * Q - 1;
        const int t = i / Q, q = i - t * Q;
        goff[n] = (t * WC + w0 * C + 4 * q) * 4;          // bytes within a plane of the sample
        ard[n] = t * pitch + HP + 4 * q;
        brd[n] = (t + 2 * R) * wtc + 4 * q;
        const int mt = (t >= 1 && t <= R) ? -t : ((t >= T - 1 - R && t <= T - 2) ? 2 * (T - 1) - t : -4 * R);
        bmir[n] = (ok[n] && mt > -4 * R) ? (mt + 2 * R) * wtc + 4 * q : -1;
    }
    int hgo[NH], hao[NH];
    float hz[NH];
    const int nhalo = T * 2 * RC;
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        int e = tid + NT * n;
        const bool hok = e < nhalo;
        e = hok ? e : nhalo - 1;
        const int t = e / (2 * RC), j = e - t * 2 * RC;
        const int side = j >= RC, jj = j - side * RC;
        const int w = (side ? w0 + a.wt : w0 - R) + jj / C, c = jj % C;
        const bool inside = w >= 0 && w < a.W;                  // columns outside the tensor do not exist: zero
        hgo[n] = (t * WC + (inside ? w : (w < 0 ? 0 : a.W - 1)) * C + c) * 4;
        hao[n] = hok ? t * pitch + (side ? HP + wtc + jj : HP - RC + jj) : dummy;
        hz[n] = (hok && inside) ? 1.f : 0.f;
    }
    auto div_m = [&](V4 g) -> V4 {           // g / m: the IEEE sequence with the loop-invariant parts hoisted (see smooth_fused3)
        if (fast_div) {
            const V4 nm = V4{-m, -m, -m, -m}, rr = V4{rcp_m, rcp_m, rcp_m, rcp_m};
            V4 q = g * rr;
            q = __builtin_elementwise_fma(__builtin_elementwise_fma(nm, q, g), rr, q);
            q = __builtin_elementwise_fma(__builtin_elementwise_fma(nm, q, g), rr, q);
            return q;
        }
        return V4{g[0] / m, g[1] / m, g[2] / m, g[3] / m};
    };
    V4 win[NI][NW];
    float hwin[NH][NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
#pragma unroll
        for (int n = 0; n < NI; ++n) win[n][j] = V4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NH; ++n) hwin[n][j] = 0.f;
    }
    V4 x[NI], of[NI];
    float hx[NH], hof[NH];
    double sdot = 0.0;
    const int hlo = h0 - R, hhi = h1 + R;
    auto fetch = [&](int hp) {
        const int hc = min(max(hp, 0), a.H - 1);
        const int so = hc * plane_bytes;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            x[n] = __builtin_bit_cast(V4, (U4)__builtin_amdgcn_raw_buffer_load_b128(rg, goff[n], so, 0));
            of[n] = __builtin_bit_cast(V4, (U4)__builtin_amdgcn_raw_buffer_load_b128(ro, goff[n], so, 0));
        }
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            hx[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, hgo[n], so, 0));
            if (XM) hof[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ro, hgo[n], so, 0));
        }
    };
    fetch(hlo);
    auto step = [&](auto slot, const int hp) {
        constexpr int s = decltype(slot)::value;
        const bool exists = hp >= 0 && hp < a.H, owned = hp >= h0 && hp < h1;       // uniform
        // ---- x = gout / max into the window's slot s; the two sums from the owned elements
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const V4 g = x[n];
            V4 xs = div_m(g);
            if (XM) {
#pragma unroll
                for (int c = 0; c < 4; ++c) xs[c] = xs[c] - (of[n][c] == 1.0f ? corr : 0.f);
            }
            win[n][s] = exists ? xs : V4{0.f, 0.f, 0.f, 0.f};
            if (!XM && owned && ok[n]) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    sdot = fma((double)g[c], (double)of[n][c], sdot);
                    if (of[n][c] == 1.0f) note.add(base + (int64_t)hp * P + (goff[n] >> 2) + c);
                }
            }
        }
        {
            V4 q = div_m(V4{hx[0], hx[1], hx[2], 0.f});          // NH == 3
            if (XM) {
#pragma unroll
                for (int n = 0; n < NH; ++n) q[n] = q[n] - (hof[n] == 1.0f ? corr : 0.f);
            }
#pragma unroll
            for (int n = 0; n < NH; ++n) hwin[n][s] = exists ? q[n] * hz[n] : 0.f;
        }
        fetch(min(hp + 1, hhi - 1));
        const int hout = hp - R;
        if (hout < h0) return;              // window not full yet (uniform)
        // ---- H^T over the window: the weights of output plane hout (border folds included; a source outside [0, H) is a zero slot)
        float wh[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            int src; float w;
            wh[j] = tap<R, true>(a.tp, hout, j - R, a.H, src, w) ? w : 0.f;
        }
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            V4 acc = V4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NW; ++j) acc = __builtin_elementwise_fma(V4{wh[j], wh[j], wh[j], wh[j]}, win[n][(s + 1 + j) % NW], acc);
            *reinterpret_cast<V4*>(f3lds + (ok[n] ? ard[n] : dummy)) = acc;
        }
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NW; ++j) acc = fmaf(wh[j], hwin[n][(s + 1 + j) % NW], acc);
            f3lds[hao[n]] = acc;
        }
        lds_barrier();
        // ---- W^T: A -> B (interior columns)
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            float v[4 * (2 * NQ + 1)];
#pragma unroll
            for (int j = 0; j < 2 * NQ + 1; ++j) {
                const V4 p = *reinterpret_cast<const V4*>(A + ard[n] + 4 * (j - NQ));
                v[4 * j] = p[0]; v[4 * j + 1] = p[1]; v[4 * j + 2] = p[2]; v[4 * j + 3] = p[3];
            }
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int k = -R; k <= R; ++k) acc = fmaf(a.tp.w[k + R], v[HP + j + k * C], acc);
                r[j] = acc;
            }
            *reinterpret_cast<V4*>(f3lds + (ok[n] ? T * pitch + brd[n] : dummy)) = V4{r[0], r[1], r[2], r[3]};
        }
        lds_barrier();
        if (left_edge || right_edge) {      // uniform: fold the pad columns back, B[q] += y[-q] resp. y[2(W-1)-q]
            const int per_side = T * RC, total = per_side * ((left_edge ? 1 : 0) + (right_edge ? 1 : 0));
            for (int e = tid; e < total; e += NT) {
                const bool right = !left_edge || e >= per_side;
                const int e2 = e - ((left_edge && right) ? per_side : 0);
                const int t = e2 / RC, jc = e2 - t * RC, j = jc / C + 1, c = jc - (j - 1) * C;      // folded column q = j (left) / wt-1-j (right)
                const float* row = A + t * pitch + HP + c;
                float sum = 0.f;
                for (int k = j; k <= R; ++k)        // y[-j] = sum_{k >= j} w[k] x[k-j];  right: the mirror image
                    sum = fmaf(a.tp.w[k + R], right ? row[(a.wt - 1 - (k - j)) * C] : row[(k - j) * C], sum);
                float* dst = Bz + (t + 2 * R) * wtc + (right ? a.wt - 1 - j : j) * C + c;
                *dst += sum;
            }
            lds_barrier();
        }
        // ---- T^T: B -> din; rows next to the border also take their mirror row of the zero-extended buffer
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            V4 acc = V4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = -R; k <= R; ++k) {
                const float w = a.tp.w[k + R];
                acc = __builtin_elementwise_fma(V4{w, w, w, w}, *reinterpret_cast<const V4*>(Bz + brd[n] + k * wtc), acc);
            }
            if (bmir[n] >= 0) {
                V4 acc2 = V4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = -R; k <= R; ++k) {
                    const float w = a.tp.w[k + R];
                    acc2 = __builtin_elementwise_fma(V4{w, w, w, w}, *reinterpret_cast<const V4*>(Bz + bmir[n] + k * wtc), acc2);
                }
                acc += acc2;
            }
            if (ok[n]) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(U4, acc), rd, goff[n], hout * plane_bytes, 0);
        }
    };
#define KCCOT_F3_STEP(S) if constexpr (S < NW) { if (hp0 + S < hhi) step(std::integral_constant<int, S>{}, hp0 + S); }
    for (int hp0 = hlo; hp0 < hhi; hp0 += NW) {
        KCCOT_F3_STEP(0) KCCOT_F3_STEP(1) KCCOT_F3_STEP(2) KCCOT_F3_STEP(3) KCCOT_F3_STEP(4)
        KCCOT_F3_STEP(5) KCCOT_F3_STEP(6) KCCOT_F3_STEP(7) KCCOT_F3_STEP(8)
    }
#undef KCCOT_F3_STEP
    if (!XM) note.store(a.ties + blockIdx.x, sdot);
}

static int launch_fused3_adj(const Fused3Plan& pl, Fused3AdjArgs fa, int C, bool xm, hipStream_t st) {
#define KCCOT_F3A(CCC, NN, XX) hipLaunchKernelGGL((smooth_fused3_adj<3, CCC, NN, XX>), dim3((unsigned)pl.grid), dim3(pl.nt), pl.lds, st, fa)
#define KCCOT_F3A_N(CCC, XX) switch (pl.ni) { case 1: KCCOT_F3A(CCC, 1, XX); break; case 2: KCCOT_F3A(CCC, 2, XX); break; default: KCCOT_F3A(CCC, 3, XX); break; }
    if (xm) { if (C == 1) { KCCOT_F3A_N(1, true) } else { KCCOT_F3A_N(3, true) } }
    else { if (C == 1) { KCCOT_F3A_N(1, false) } else { KCCOT_F3A_N(3, false) } }
#undef KCCOT_F3A_N
#undef KCCOT_F3A
    return launch_status("smooth_fused3_adj");
}

// rows per workgroup (~16 KB of LDS); 0 = the row does not fit (W * C > 16384)
static int wrow_rpw(int W, int C) {
    const int64_t WC = (int64_t)W * C;
    if (WC > 12288) return 0;
    const int rpw = (int)(4096 / WC);
    return rpw < 1 ? 1 : rpw;
}

static int launch_wrow(const float* in, float* out, int64_t n, int W, int C, int radius, bool adjoint, const Taps& tp,
                       hipStream_t st) {
    const int rpw = wrow_rpw(W, C);
    const int64_t nrows = n / ((int64_t)W * C);
    const dim3 grid((unsigned)((nrows + rpw - 1) / rpw));
    const size_t lds = ((size_t)((rpw * W * C + 3) & ~3) + (adjoint ? (size_t)W * (2 * radius + 1) : 0)) * sizeof(float);
#define KCCOT_WROW(RR, AA, CCC) hipLaunchKernelGGL((smooth_wrow<RR, AA, CCC>), grid, dim3(256), lds, st, in, out, nrows, W, C, rpw, tp)
#define KCCOT_WROW_C(RR, AA)                                                                      \
    switch (C) { case 1: KCCOT_WROW(RR, AA, 1); break; case 2: KCCOT_WROW(RR, AA, 2); break;      \
                 case 3: KCCOT_WROW(RR, AA, 3); break; case 4: KCCOT_WROW(RR, AA, 4); break;      \
                 default: KCCOT_WROW(RR, AA, 0); break; }
    if (radius == 3) { if (adjoint) { KCCOT_WROW_C(3, true) } else { KCCOT_WROW_C(3, false) } }
    else { if (adjoint) { KCCOT_WROW_C(4, true) } else { KCCOT_WROW_C(4, false) } }
#undef KCCOT_WROW_C
#undef KCCOT_WROW
    return launch_status("smooth_wrow");
}

static bool wrow_eligible(int W, int C, int radius) {
    return (radius == 3 || radius == 4) && wrow_rpw(W, C) > 0 && W > radius &&
           ((size_t)wrow_rpw(W, C) * W * C + (size_t)W * (2 * radius + 1) + 4) * sizeof(float) <= 64 * 1024;
}

static bool w1_eligible(int W, int C, int radius, const void* a, const void* b) {
    return C == 1 && (radius == 3 || radius == 4) && W % 4 == 0 && W >= 8 && W <= 128 &&
           (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0;
}

static int launch_w1(const float* in, float* out, int64_t n, int W, int radius, bool adjoint, const Taps& tp, hipStream_t st) {
    const int64_t np = n / 4;
    const dim3 grid((unsigned)((np + 255) / 256));
    if (radius == 3) {
        if (adjoint) hipLaunchKernelGGL((smooth_w1<3, true>), grid, dim3(256), 0, st, in, out, np, W, tp);
        else hipLaunchKernelGGL((smooth_w1<3, false>), grid, dim3(256), 0, st, in, out, np, W, tp);
    } else {
        if (adjoint) hipLaunchKernelGGL((smooth_w1<4, true>), grid, dim3(256), 0, st, in, out, np, W, tp);
        else hipLaunchKernelGGL((smooth_w1<4, false>), grid, dim3(256), 0, st, in, out, np, W, tp);
    }
    return launch_status("smooth_w1");
}

// vector width and unroll bound for an axis of length L whose lines are `S` floats apart; 0 = not eligible
struct WalkPlan { int vw, lmax; };
static WalkPlan walk_plan(int L, int64_t S, bool two_inputs, const void* p0, const void* p1, const void* p2) {
    WalkPlan pl{0, 0};
    if (L > 64) return pl;
    pl.lmax = L <= 32 ? 32 : 64;
    // registers: LMAX x VW floats for the line (twice that in flight while two tensors are being read)
    int vw = (pl.lmax == 32) ? (two_inputs ? 2 : 4) : (two_inputs ? 1 : 2);
    while (vw > 1 && (S % vw != 0 || (uintptr_t)p0 % (4 * vw) || (uintptr_t)p1 % (4 * vw) || (uintptr_t)p2 % (4 * vw))) vw >>= 1;
    pl.vw = vw;
    return pl;
}

template <int R, int MODE>
static void launch_walk_r(const WalkArgs& wa, WalkPlan pl, hipStream_t st) {
    const dim3 grid((unsigned)((wa.ncols + 255) / 256));
#define KCCOT_WALK(LM, VW) hipLaunchKernelGGL((smooth_walk<R, LM, VW, MODE>), grid, dim3(256), 0, st, wa)
    if (pl.lmax == 32) { if (pl.vw == 4) KCCOT_WALK(32, 4); else if (pl.vw == 2) KCCOT_WALK(32, 2); else KCCOT_WALK(32, 1); }
    else { if (pl.vw == 2) KCCOT_WALK(64, 2); else KCCOT_WALK(64, 1); }
#undef KCCOT_WALK
}

// one pass of the walk along an axis; `numel` = elements of the tensor
static int launch_walk(int mode, WalkArgs wa, int radius, int64_t numel, WalkPlan pl, hipStream_t st) {
    wa.inner = wa.S / pl.vw;
    wa.ncols = numel / wa.L / pl.vw;
#define KCCOT_WALK_MODE(M) do { if (radius == 3) launch_walk_r<3, M>(wa, pl, st); else launch_walk_r<4, M>(wa, pl, st); } while (0)
    switch (mode) {
        case WALK_MAX: KCCOT_WALK_MODE(WALK_MAX); break;
        case WALK_WRITE: KCCOT_WALK_MODE(WALK_WRITE); break;
        case WALK_RAW: KCCOT_WALK_MODE(WALK_RAW); break;
        case WALK_ADJ: KCCOT_WALK_MODE(WALK_ADJ); break;
        case WALK_RAW_TW: KCCOT_WALK_MODE(WALK_RAW_TW); break;
        case WALK_ADJS: KCCOT_WALK_MODE(WALK_ADJS); break;
        default: KCCOT_WALK_MODE(WALK_ADJX); break;
    }
#undef KCCOT_WALK_MODE
    return launch_status("smooth_walk");
}

template <int R, int MODE>
static void launch_roll_r(const WalkArgs& wa, int vw, hipStream_t st) {
    const dim3 grid((unsigned)((wa.ncols + 255) / 256));
    constexpr bool ADJ = MODE == WALK_ADJ || MODE == WALK_ADJX || MODE == WALK_ADJS;
    const size_t lds = ADJ ? (size_t)wa.L * (2 * R + 1) * sizeof(float) : 0;
    if (vw == 4) hipLaunchKernelGGL((smooth_roll<R, 4, MODE>), grid, dim3(256), lds, st, wa);
    else if (vw == 2) hipLaunchKernelGGL((smooth_roll<R, 2, MODE>), grid, dim3(256), lds, st, wa);
    else hipLaunchKernelGGL((smooth_roll<R, 1, MODE>), grid, dim3(256), lds, st, wa);
}

static int launch_roll(int mode, WalkArgs wa, int radius, int64_t numel, int vw, hipStream_t st) {
    wa.inner = wa.S / vw;
    wa.ncols = numel / wa.L / vw;
#define KCCOT_ROLL_MODE(M) do { if (radius == 3) launch_roll_r<3, M>(wa, vw, st); else launch_roll_r<4, M>(wa, vw, st); } while (0)
    switch (mode) {
        case WALK_MAX: KCCOT_ROLL_MODE(WALK_MAX); break;
        case WALK_WRITE: KCCOT_ROLL_MODE(WALK_WRITE); break;
        case WALK_RAW: KCCOT_ROLL_MODE(WALK_RAW); break;
        case WALK_ADJ: KCCOT_ROLL_MODE(WALK_ADJ); break;
        case WALK_ADJX: KCCOT_ROLL_MODE(WALK_ADJX); break;
        case WALK_ADJS: KCCOT_ROLL_MODE(WALK_ADJS); break;
        default: return fail(KCCOT_EINVAL, "smooth: mode %d has no rolling form", mode);
    }
#undef KCCOT_ROLL_MODE
    return launch_status("smooth_roll");
}

// How one strided axis is walked: AXIS_LINE = smooth_walk (the line in registers, L <= 64), AXIS_ROLL = smooth_roll (any L),
// AXIS_NONE = neither (radius other than 3 / 4, or an adjoint weight table that does not fit LDS): the caller falls back to
// the per-element chain.  KCCOT_SMOOTH_GENERIC=1 takes the rolling form wherever it exists (tests, A/B).
enum { AXIS_NONE = 0, AXIS_LINE = 1, AXIS_ROLL = 2 };
static bool smooth_generic() { return opt(OPT_SMOOTH_GENERIC) != 0; }   // option "smooth_generic" = 1: the any-shape kernels everywhere
struct AxisPlan { int kind; WalkPlan wp; int vw; };

static AxisPlan axis_plan(int L, int64_t S, bool two_inputs, bool adjoint, int radius, const void* p0, const void* p1,
                          const void* p2) {
    AxisPlan ap{AXIS_NONE, WalkPlan{0, 0}, 0};
    if (radius != 3 && radius != 4) return ap;
    ap.wp = walk_plan(L, S, two_inputs, p0, p1, p2);
    if (ap.wp.vw > 0 && !smooth_generic()) { ap.kind = AXIS_LINE; ap.vw = ap.wp.vw; return ap; }
    if (adjoint && (size_t)L * (2 * radius + 1) * sizeof(float) > 48 * 1024) return ap;
    int vw = two_inputs ? 2 : 4;
    while (vw > 1 && (S % vw != 0 || (uintptr_t)p0 % (4 * vw) || (uintptr_t)p1 % (4 * vw) || (uintptr_t)p2 % (4 * vw))) vw >>= 1;
    ap.kind = AXIS_ROLL; ap.vw = vw;
    return ap;
}

static int launch_axis(int mode, const WalkArgs& wa, int radius, int64_t numel, const AxisPlan& ap, hipStream_t st) {
    if (ap.kind == AXIS_LINE) return launch_walk(mode, wa, radius, numel, ap.wp, st);
    return launch_roll(mode, wa, radius, numel, ap.vw, st);
}

// sum(gout * out) and #(out == 1) with wide grid-stride loads (the per-element form above launches
// n/256 workgroups of one element per thread)
__global__ __launch_bounds__(256) void maxnorm_bwd_partial_v4(const float* __restrict__ gout, const float* __restrict__ out,
                                                              int64_t n4, float* __restrict__ pdot, float* __restrict__ pcnt) {
    __shared__ float red[16];
    float d = 0.f, c = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const float4 gg = reinterpret_cast<const float4*>(gout)[e];
        const float4 oo = reinterpret_cast<const float4*>(out)[e];
        d = fmaf(gg.x, oo.x, d); d = fmaf(gg.y, oo.y, d); d = fmaf(gg.z, oo.z, d); d = fmaf(gg.w, oo.w, d);
        c += (oo.x == 1.0f ? 1.f : 0.f) + (oo.y == 1.0f ? 1.f : 0.f) + (oo.z == 1.0f ? 1.f : 0.f) + (oo.w == 1.0f ? 1.f : 0.f);
    }
    const float ds = block_sum(d, red);
    const float cs = block_sum(c, red);
    if (threadIdx.x == 0) { pdot[blockIdx.x] = ds; pcnt[blockIdx.x] = cs; }
}

// ---- the sparse half of the folded backward (WALK_ADJS) ----------------------------------------------------------------
// One workgroup.  Adds the workgroups' records up (res[0] = sum g * out in fp64, res[1] = number of arg-max elements),
// and, when there are at most SMOOTH_MAX_TIES of them (one, for any real video), subtracts corr * A^T [out == 1] from din:
// per arg-max element the (2R+1)^axes REFLECT neighbours, one thread per tap combination, taps that fold onto the same
// position merged first (one writer per position), the elements in index order one after the other: deterministic.
// More arg-max elements than that (a saturated still image: a whole region ties) -> *dense = 1 and nothing is touched;
// the guarded dense chain behind this launch (the round-2 kernels with WalkArgs::run_if) then recomputes din the plain way.
struct FixupArgs {
    const TieRec* ties; int nrec;
    const float* mx; float* res; int* dense; float* din;
    int len[3]; long long stride[3]; int na;   // the smoothed axes
    Taps tp;
};

__global__ __launch_bounds__(1024) void maxnorm_bwd_fixup(FixupArgs a) {
    __shared__ double dsum[16];
    __shared__ int isum[16];
    __shared__ int nlist, overflow;
    __shared__ long long list[SMOOTH_MAX_TIES];
    if (threadIdx.x == 0) { nlist = 0; overflow = 0; }
    __syncthreads();
    const float m = a.mx[0];            // early: its latency hides under the scan
    double d = 0.0;
    int c = 0;
    for (int i = threadIdx.x; i < a.nrec; i += 1024) {
        const TieRec r = a.ties[i];
        d += r.dot;
        c += r.ties;
        if (r.ties > TIE_PER_WG) overflow = 1;
        for (int j = 0; j < r.ties && j < TIE_PER_WG; ++j) {
            const int slot = atomicAdd(&nlist, 1);
            if (slot < SMOOTH_MAX_TIES) list[slot] = r.idx[j];
        }
    }
    d = wave_sum_d(d);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) { dsum[threadIdx.x >> 6] = d; isum[threadIdx.x >> 6] = c; }
    __syncthreads();
    d = 0.0; c = 0;
    for (int w = 0; w < 16; ++w) { d += dsum[w]; c += isum[w]; }
    const bool dense = overflow != 0 || c > SMOOTH_MAX_TIES;
    if (threadIdx.x == 0) { a.res[0] = (float)d; a.res[1] = (float)c; *a.dense = dense ? 1 : 0; }
    if (dense || c == 0) return;
    if (threadIdx.x == 0)               // index order (the records arrive in any order)
        for (int i = 1; i < c; ++i) {
            const long long v = list[i];
            int j = i - 1;
            for (; j >= 0 && list[j] > v; --j) list[j + 1] = list[j];
            list[j + 1] = v;
        }
    __syncthreads();
    const float corr = (float)d / (m * (float)c);
    const int nt = 2 * a.tp.r + 1;
    int combos = 1;
    for (int x = 0; x < a.na; ++x) combos *= nt;
    for (int t = 0; t < c; ++t) {
        const long long e = list[t];
        for (int q = threadIdx.x; q < combos; q += 1024) {
            int rem = q;
            long long target = e;
            float w = 1.f;
            bool owner = true;
            for (int x = 0; x < a.na; ++x) {
                const int k = rem % nt;
                rem /= nt;
                const int p0 = (int)((e / a.stride[x]) % a.len[x]);
                const int pos = reflect(p0 + k - a.tp.r, a.len[x]);
                float wx = 0.f;
                for (int k2 = 0; k2 < nt; ++k2)
                    if (reflect(p0 + k2 - a.tp.r, a.len[x]) == pos) {
                        if (k2 < k) owner = false;
                        wx += a.tp.w[k2];
                    }
                w *= wx;
                target += (long long)(pos - p0) * a.stride[x];
            }
            if (owner) a.din[target] -= corr * w;
        }
        __syncthreads();
    }
}

// The W stage of the backward's dense chain (its H and T stages are the walks themselves, WalkArgs::run_if): a no-op
// unless *run_if != 0 -- decided on the device, without a host round trip.  A small fixed grid with a grid-stride loop: a
// launch that does nothing must cost nothing (one workgroup per 256 elements takes the dispatcher longer to start and
// retire than the folded kernels run).  W is the contiguous axis: neighbouring threads read neighbouring addresses.
__global__ __launch_bounds__(256) void conv_axis_adjoint_if(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                            int len, int64_t stride, Taps tp,
                                                            const int* __restrict__ run_if) {
    if (*run_if == 0) return;
    const int r = tp.r;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int p = (int)((e / stride) % len);
        const float* base = in + (e - (int64_t)p * stride);
        float acc = 0.f;
        for (int d = -r; d <= r; ++d) {
            const float w = tp.w[d + r];
            int t = p - d;
            if (t >= 0 && t < len) acc = fmaf(w, base[(int64_t)t * stride], acc);
            t = -p - d;
            if (p >= 1 && t >= 0 && t < len) acc = fmaf(w, base[(int64_t)t * stride], acc);
            t = 2 * (len - 1) - p - d;
            if (p <= len - 2 && t >= 0 && t < len) acc = fmaf(w, base[(int64_t)t * stride], acc);
        }
        out[e] = acc;
    }
}

// {1, 0, 0}: "maximum 1, no arg-max elements" for a plane kernel that runs a plain stencil.  A one-thread KERNEL, not
// hipMemsetD32Async: as nodes of a captured graph the library's memsets did not take effect for the kernel nodes behind
// them on this stack (ROCm 7.2) -- the first replay ran on fresh memory and was right, every later one saw the previous
// replay's words (here: garbage scalars -> NaN in the second replay of a captured 3-D backward; in sinkhorn_coop.hip: the
// previous replay's exchange tags).  Whether the memset node lacked its edge or its writes bypassed what the kernels' loads
// read was not isolated; the remedy covers both: a KERNEL writes the words (round 3, test_smoothing_replays_as_a_graph).
__global__ void set_unit_scalars(float* __restrict__ s) { s[0] = 1.f; s[1] = 0.f; s[2] = 0.f; }

static bool plane_eligible(int T, int W, int C, int radius, int naxes) {
    return naxes > 0 && (radius == 3 || radius == 4) && (int64_t)T * W * C <= 4096;
}

static int plane_hseg(int B, int H, bool halo) {
    // >= 2 workgroups per CU; with an H stencil each segment re-reads 2R halo planes, so segments
    // stay as long as that allows
    for (int hs = halo ? 16 : 8; hs > 2; hs >>= 1)
        if ((int64_t)B * ((H + hs - 1) / hs) >= 512) return hs;
    return 2;
}

static int launch_plane(const PlaneArgs& pa, int radius, bool adjoint, dim3 grid, hipStream_t st) {
    const int P = pa.T * pa.W * pa.C;
    const int threads = ((P + SP_PPT - 1) / SP_PPT + 63) / 64 * 64;
    const size_t lds = (size_t)2 * P * sizeof(float);
#define KCCOT_SP(RR, AA)                                                                                   \
    do {                                                                                                   \
        if (threads <= 512) hipLaunchKernelGGL((smooth_plane<RR, AA, 512>), grid, dim3(threads), lds, st, pa);   \
        else hipLaunchKernelGGL((smooth_plane<RR, AA, 1024>), grid, dim3(threads), lds, st, pa);           \
    } while (0)
    if (radius == 3) { if (adjoint) KCCOT_SP(3, true); else KCCOT_SP(3, false); }
    else { if (adjoint) KCCOT_SP(4, true); else KCCOT_SP(4, false); }
#undef KCCOT_SP
    return launch_status("smooth_plane");
}

struct Axis { int len; int64_t stride; };

static int collect_axes(int H, int T, int W, int C, unsigned flags, Axis* ax) {
    int na = 0;
    if (flags & KCCOT_SMOOTH_T) ax[na++] = Axis{T, (int64_t)W * C};
    if (flags & KCCOT_SMOOTH_H) ax[na++] = Axis{H, (int64_t)T * W * C};
    if (flags & KCCOT_SMOOTH_W) ax[na++] = Axis{W, (int64_t)C};
    return na;
}

}  // namespace kccot

using namespace kccot;

extern "C" size_t kccot_smooth_workspace_bytes(int B, int H, int T, int W, int C) {
    if (B <= 0 || H <= 0 || T <= 0 || W <= 0 || C <= 0) return 0;
    const size_t n = (size_t)B * H * T * W * C;
    const size_t nb = (n + 255) / 256;
    // one tensor-sized ping-pong buffer + two per-block reduction arrays + scalars + the backward's per-workgroup tie
    // records (a line kernel's workgroup covers >= 256 lines of >= 4 elements)
    return align_up(n * sizeof(float), 256) + 2 * align_up(nb * sizeof(float), 256) + 256 +
           align_up((n / 1024 + 2) * sizeof(TieRec), 256);
}

static int smooth_check(const char* who, const void* a, const void* b, int B, int H, int T, int W, int C, float sigma,
                        int radius, unsigned flags) {
    if (!a || !b) return fail(KCCOT_EINVAL, "%s: null pointer", who);
    if (B <= 0 || H <= 0 || T <= 0 || W <= 0 || C <= 0)
        return fail(KCCOT_EINVAL, "%s: bad shape [%d,%d,%d,%d,%d]", who, B, H, T, W, C);
    if (radius < 0 || radius > SM_MAXR) return fail(KCCOT_EUNSUPPORTED, "%s: radius %d > %d", who, radius, SM_MAXR);
    if (!(sigma > 0.f)) return fail(KCCOT_EINVAL, "%s: sigma must be > 0", who);
    // REFLECT padding needs pad < dim (tf.pad rejects it otherwise)
    if (((flags & KCCOT_SMOOTH_T) && radius >= T) || ((flags & KCCOT_SMOOTH_H) && radius >= H) ||
        ((flags & KCCOT_SMOOTH_W) && radius >= W))
        return fail(KCCOT_EINVAL, "%s: REFLECT padding needs radius < axis length", who);
    return 0;
}

extern "C" int kccot_smooth_fwd_f32(const float* in, int B, int H, int T, int W, int C, float sigma, int radius,
                                    unsigned flags, float* out, float* max_inout, void* ws, size_t ws_bytes,
                                    kccot_stream_t stream) {
    int rc = smooth_check("smooth_fwd", in, out, B, H, T, W, C, sigma, radius, flags);
    if (rc) return rc;
    if (!max_inout) return fail(KCCOT_EINVAL, "smooth_fwd: null max pointer");
    const size_t need = kccot_smooth_workspace_bytes(B, H, T, W, C);
    if (!ws || ws_bytes < need) return fail(KCCOT_EWORKSPACE, "smooth_fwd: workspace %zu < required %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * H * T * W * C;
    const int64_t nb = (n + 255) / 256;
    if (nb > 0x7fffffff) return fail(KCCOT_EUNSUPPORTED, "smooth_fwd: tensor too large");
    float* tmp = static_cast<float*>(ws);
    float* bmax = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)n * sizeof(float), 256));
    Axis ax[3];
    const int na = collect_axes(H, T, W, C, flags, ax);
    const bool ext = (flags & KCCOT_SMOOTH_EXTERNAL_MAX) != 0, nodiv = (flags & KCCOT_SMOOTH_NO_DIVIDE) != 0;
    if (ext && nodiv) return fail(KCCOT_EINVAL, "smooth_fwd: EXTERNAL_MAX and NO_DIVIDE are exclusive");
    if (na > 0 && in == out) return fail(KCCOT_EINVAL, "smooth_fwd: in-place convolution is not supported");
    const Taps tp = make_taps(sigma, radius);
    const unsigned axes = flags & (KCCOT_SMOOTH_T | KCCOT_SMOOTH_H | KCCOT_SMOOTH_W);
    const bool r34 = radius == 3 || radius == 4;
    float* one = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)n * sizeof(float), 256) +
                                          2 * align_up((size_t)nb * sizeof(float), 256));   // scalar slots: {1, 0, 0}
    // (NO_DIVIDE runs here too, since round 2: both phases of the batch-sharded protocol -- local maximum, then the
    // division by the all-reduced one -- must evaluate s with the SAME kernels, or the arg-max element comes out as
    // 0.99999994 instead of exactly 1 and the adjoint's `out == 1` tie detection finds nothing: found by the RCCL
    // world-size-1 test, where phase 1 took the per-axis chain and phase 2 the streamed walks)
    if (r34 && opt(OPT_SMOOTH_STREAM) && axes == (KCCOT_SMOOTH_T | KCCOT_SMOOTH_H | KCCOT_SMOOTH_W)) {
        const Fused3Plan fp = fused3_plan(B, H, T, W, C, radius, in, out);
        if (fp.ok) {
            Fused3Args fa{};
            fa.in = in; fa.H = H; fa.T = T; fa.W = W; fa.wt = fp.wt; fa.ntw = W / fp.wt; fa.hseg = fp.hseg;
            fa.nseg = (H + fp.hseg - 1) / fp.hseg; fa.tp = tp;
            if (!ext) {
                fa.mode = 0; fa.blockmax = bmax;
#ifdef KCCOT_DIAG
                if (const char* e = getenv("KCCOT_F3_ABLATE")) fa.mode |= (atoi(e) & (7 | 128 | 256 | 512)) << 8;
                if (!(getenv("KCCOT_F3_ABLATE") && (atoi(getenv("KCCOT_F3_ABLATE")) & 64)))     // bit 6: writing pass only
#endif
                if ((rc = launch_fused3(fp, fa, radius, C, st))) return rc;
                fa.mode = 0;
                if (!nodiv && fp.grid <= 4096) {
                    fa.nblk = (int)fp.grid;       // the writing pass reduces the maxima itself (and stores the maximum)
                } else {
                    hipLaunchKernelGGL(reduce_blockmax, dim3(1), dim3(1024), 0, st, (const float*)bmax, fp.grid, max_inout);
                    if ((rc = launch_status("reduce_blockmax"))) return rc;
                }
            }
            fa.mode = nodiv ? 2 : 1; fa.out = out; fa.mx = max_inout; fa.mx_out = max_inout;
#ifdef KCCOT_DIAG
            if (const char* e = getenv("KCCOT_F3_ABLATE")) {
                fa.mode |= atoi(e) << 8;
                if (atoi(e) & 32) return 0;         // bit 5: maxima pass only
            }
#endif
            return launch_fused3(fp, fa, radius, C, st);
        }
    }
    if (r34 && opt(OPT_SMOOTH_STREAM) && (axes == KCCOT_SMOOTH_T || axes == (KCCOT_SMOOTH_T | KCCOT_SMOOTH_H | KCCOT_SMOOTH_W))) {
        const int64_t WC = (int64_t)W * C;
        const bool three = axes != KCCOT_SMOOTH_T;
        // the last stage runs twice (maxima, then recompute + write s / max); earlier stages write raw sums
        AxisPlan pt = axis_plan(T, WC, false, false, radius, in, out, three ? tmp : out);
        // temporal-only call: 8-byte pieces (twice the threads) are 4-6 % faster than 16-byte ones on both passes at every
        // BASELINE shape (22.2 -> 20.8 us, 101 -> 97 us, 227 -> 216 us); the 3-D call keeps 16-byte pieces for the T+W fusion
        if (!three && pt.kind == AXIS_LINE && pt.wp.vw == 4) { pt.wp.vw = 2; pt.vw = 2; }
        const AxisPlan ph = three ? axis_plan(H, (int64_t)T * WC, false, false, radius, tmp, out, out) : AxisPlan{AXIS_LINE, WalkPlan{1, 32}, 1};
        const bool w1 = three && w1_eligible(W, C, radius, out, tmp) && !smooth_generic();
        const bool wplane = three && !w1 && plane_eligible(T, W, C, radius, 1) && !smooth_generic();
        const bool wrow = three && !w1 && !wplane && wrow_eligible(W, C, radius);
        if (pt.kind != AXIS_NONE && ph.kind != AXIS_NONE && (!three || w1 || wplane || wrow || tw_plane_eligible(T, W, C, radius, false, in, tmp))) {
            WalkArgs wa{};
            wa.tp = tp;
            const float* last_in = in;
            AxisPlan last = pt;
            wa.L = T; wa.S = WC;
            if (three) {
                // T: in -> out (raw);  W: out -> tmp (raw; the axis is contiguous: smooth_w1 / LDS rows);  H: tmp -> out
                // -- or T and W in one launch, in -> tmp (WALK_RAW_TW), when a row of W is W/4 lanes of a wave
                const int W4 = W >> 2;
                const bool tw = w1 && pt.kind == AXIS_LINE && pt.vw == 4 && W4 <= 64 && (W4 & (W4 - 1)) == 0 && opt(OPT_SMOOTH_FUSED_TW);
                // -- or, for any channel count, with the (b, h) plane staged in LDS (smooth_tw_plane)
                const bool twp = !tw && tw_plane_eligible(T, W, C, radius, false, in, tmp);
                if (twp) {
                    if ((rc = launch_tw_plane(in, tmp, n, T, W, C, radius, false, tp, st))) return rc;
                } else {
                    wa.in = in; wa.out = tw ? tmp : out;
                    if ((rc = launch_axis(tw ? WALK_RAW_TW : WALK_RAW, wa, radius, n, pt, st))) return rc;
                }
                if (tw || twp) {
                } else if (w1) {
                    if ((rc = launch_w1(out, tmp, n, W, radius, false, tp, st))) return rc;
                } else if (wrow) {
                    if ((rc = launch_wrow(out, tmp, n, W, C, radius, false, tp, st))) return rc;
                } else {
                    hipLaunchKernelGGL(set_unit_scalars, dim3(1), dim3(1), 0, st, one);
                    if ((rc = launch_status("set_unit_scalars"))) return rc;
                    PlaneArgs pa{};
                    pa.in = out; pa.out = tmp; pa.mx = one; pa.B = B; pa.H = H; pa.T = T; pa.W = W; pa.C = C;
                    pa.axes = KCCOT_SMOOTH_W; pa.tp = tp; pa.hseg = plane_hseg(B, H, false);
                    if ((rc = launch_plane(pa, radius, false, dim3((H + pa.hseg - 1) / pa.hseg, B), st))) return rc;
                }
                last_in = tmp; last = ph;
                wa.L = H; wa.S = (int64_t)T * WC;
            }
            wa.in = last_in;
            const int64_t last_wgs = (n / wa.L / last.vw + 255) / 256;
            wa.nblk = 0;
            if (!ext) {
                wa.out = nullptr; wa.blockmax = bmax;
                if ((rc = launch_axis(WALK_MAX, wa, radius, n, last, st))) return rc;
                if (!nodiv && last_wgs <= 4096) {
                    wa.nblk = (int)last_wgs;      // the writing pass reduces the block maxima itself (and stores the maximum)
                } else {
                    hipLaunchKernelGGL(reduce_blockmax, dim3(1), dim3(1024), 0, st, (const float*)bmax, last_wgs, max_inout);
                    if ((rc = launch_status("reduce_blockmax"))) return rc;
                }
            }
            wa.out = out; wa.mx = max_inout; wa.mx_out = max_inout;
            if (wa.nblk == 0) wa.blockmax = nullptr;
            return launch_axis(nodiv ? WALK_RAW : WALK_WRITE, wa, radius, n, last, st);   // NO_DIVIDE: the raw sums
        }
    }
    if (plane_eligible(T, W, C, radius, na) && !nodiv) {
        PlaneArgs pa{};
        pa.in = in; pa.B = B; pa.H = H; pa.T = T; pa.W = W; pa.C = C; pa.axes = flags; pa.tp = tp;
        pa.hseg = plane_hseg(B, H, (flags & KCCOT_SMOOTH_H) != 0);
        const dim3 grid((H + pa.hseg - 1) / pa.hseg, B);
        if (!ext) {   // pass 1: maxima only
            pa.out = nullptr; pa.blockmax = bmax;
            if ((rc = launch_plane(pa, radius, false, grid, st))) return rc;
            hipLaunchKernelGGL(reduce_blockmax, dim3(1), dim3(1024), 0, st, (const float*)bmax, (int64_t)grid.x * grid.y, max_inout);
            if ((rc = launch_status("reduce_blockmax"))) return rc;
        }
        pa.out = out; pa.blockmax = nullptr; pa.mx = max_inout;   // pass 2: recompute, write s / max
        return launch_plane(pa, radius, false, grid, st);
    }
    const float* src = in;
    for (int i = 0; i < na; ++i) {
        float* dst = ((na - 1 - i) % 2 == 0) ? out : tmp;
        const bool last = (i == na - 1);
        hipLaunchKernelGGL(conv_axis, dim3((unsigned)nb), dim3(256), 0, st, src, dst, n, ax[i].len, ax[i].stride, tp, 0,
                           (last && !ext) ? bmax : (float*)nullptr);
        if ((rc = launch_status("conv_axis"))) return rc;
        src = dst;
    }
    if (na == 0 && !ext) {   // max (and copy) only
        hipLaunchKernelGGL(copy_with_blockmax, dim3((unsigned)nb), dim3(256), 0, st, in, out, n, bmax);
        if ((rc = launch_status("copy_with_blockmax"))) return rc;
    }
    if (!ext) {
        hipLaunchKernelGGL(reduce_blockmax, dim3(1), dim3(1024), 0, st, (const float*)bmax, nb, max_inout);
        if ((rc = launch_status("reduce_blockmax"))) return rc;
    }
    if (!nodiv) {
        const float* dsrc = (na == 0 && ext) ? in : out;
        hipLaunchKernelGGL(divide_by, dim3((unsigned)nb), dim3(256), 0, st, dsrc, out, n, (const float*)max_inout);
        if ((rc = launch_status("divide_by"))) return rc;
    }
    return 0;
}

// stats_ext: null = the normalisation adjoint's two batch sums {sum(g * out), #(out == 1)} are computed and used
// here; with KCCOT_SMOOTH_STATS_ONLY they are written to stats_ext and nothing else happens; with
// KCCOT_SMOOTH_EXTERNAL_STATS they are READ from stats_ext (the batch-sharded caller has all-reduced(SUM) them).
static int smooth_bwd_impl(const float* gout, const float* out, const float* max_in, float* stats_ext, int B, int H, int T,
                           int W, int C, float sigma, int radius, unsigned flags, float* din, void* ws,
                           size_t ws_bytes, kccot_stream_t stream) {
    const bool stats_only = (flags & KCCOT_SMOOTH_STATS_ONLY) != 0, stats_in = (flags & KCCOT_SMOOTH_EXTERNAL_STATS) != 0;
    if (stats_only && stats_in) return fail(KCCOT_EINVAL, "smooth_bwd: STATS_ONLY and EXTERNAL_STATS are exclusive");
    if ((stats_only || stats_in) && !stats_ext) return fail(KCCOT_EINVAL, "smooth_bwd: null stats pointer");
    int rc = smooth_check("smooth_bwd", gout, stats_only ? const_cast<float*>(gout) : din, B, H, T, W, C, sigma, radius, flags);
    if (rc) return rc;
    if (!out || !max_in) return fail(KCCOT_EINVAL, "smooth_bwd: null pointer");
    const size_t need = kccot_smooth_workspace_bytes(B, H, T, W, C);
    if (!ws || ws_bytes < need) return fail(KCCOT_EWORKSPACE, "smooth_bwd: workspace %zu < required %zu", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * H * T * W * C;
    const int64_t nb = (n + 255) / 256;
    float* tmp = static_cast<float*>(ws);
    char* p = static_cast<char*>(ws) + align_up((size_t)n * sizeof(float), 256);
    float* pdot = reinterpret_cast<float*>(p);
    float* pcnt = reinterpret_cast<float*>(p + align_up((size_t)nb * sizeof(float), 256));
    float* res = reinterpret_cast<float*>(p + 2 * align_up((size_t)nb * sizeof(float), 256));
    Axis ax[3];
    const int na = collect_axes(H, T, W, C, flags, ax);
    const Taps tp = make_taps(sigma, radius);
    // ds goes where the adjoint chain wants its first source: (na passes) ... -> din
    float* ds = (na % 2 == 0) ? din : tmp;
    const bool wide = (n % 4 == 0) && ((uintptr_t)gout % 16 == 0) && ((uintptr_t)out % 16 == 0) && nb >= 2048;
    const int64_t nparts = wide ? 2048 : nb;
    float* scal = reinterpret_cast<float*>(p + 2 * align_up((size_t)nb * sizeof(float), 256));   // workspace scalars {dot, ties, .., .., 1, 0, 0, dense}
    TieRec* recs = reinterpret_cast<TieRec*>(p + 2 * align_up((size_t)nb * sizeof(float), 256) + 256);
    const unsigned axes = flags & (KCCOT_SMOOTH_T | KCCOT_SMOOTH_H | KCCOT_SMOOTH_W);
    // The streaming chains (temporal only / all three axes, radius 3 or 4).  Decided in front of the statistics: with the
    // option "smooth_bwd_fold" the chain's first stage gathers them itself (WALK_ADJS) and the pass below is skipped.
    const int64_t WC = (int64_t)W * C;
    const bool three = axes != KCCOT_SMOOTH_T;
    bool chain = (radius == 3 || radius == 4) && opt(OPT_SMOOTH_STREAM) && !stats_only &&
                 (axes == KCCOT_SMOOTH_T || axes == (KCCOT_SMOOTH_T | KCCOT_SMOOTH_H | KCCOT_SMOOTH_W));
    AxisPlan ph{AXIS_LINE, WalkPlan{1, 32}, 1}, pt{AXIS_NONE, WalkPlan{0, 0}, 0};
    bool w1 = false, wplane = false, wrow = false, twp = false;
    if (chain) {
        // adjoint stages in reverse order; the first one also applies the adjoint of the max-normalisation
        if (three) ph = axis_plan(H, (int64_t)T * WC, true, true, radius, gout, out, din);
        pt = axis_plan(T, WC, !three, true, radius, three ? (const void*)tmp : (const void*)gout, out, din);
        w1 = three && w1_eligible(W, C, radius, din, tmp) && !smooth_generic();
        wplane = three && !w1 && plane_eligible(T, W, C, radius, 1) && !smooth_generic();
        wrow = three && !w1 && !wplane && wrow_eligible(W, C, radius);
        twp = three && tw_plane_eligible(T, W, C, radius, true, tmp, din);
        chain = pt.kind != AXIS_NONE && ph.kind != AXIS_NONE && (!three || w1 || wplane || wrow || twp);
    }
    // Folding trades two tensor reads for two (temporal) / four (3-D) more launches of ~5 us each behind the chain (the
    // fix-up and the guarded dense chain, dependent launches on one stream).  Measured: temporal 17.7 -> 22.0 us at 2 M
    // elements, 33.9 -> 32.8 us at 7.9 M, 350 -> 254 us at 94 M; 3-D 37.8 -> 61.1 us at 2 M, 65.4 -> 79.0 us at 7.9 M,
    // 684 -> 635 us at 94 M.  Option value 1 folds from 4 M (temporal) / 32 M (3-D) elements on, 2 always (tests), 0 never.
    const int fold_opt = opt(OPT_SMOOTH_BWD_FOLD);
    // Round 4: with the three adjoint stages in ONE pass (smooth_fused3_adj, which gathers the sums itself) folding pays from
    // 3.5 M elements on for the 3-D call -- the plan decides.
    Fused3Plan fpa{};
    if (chain && three && (stats_in || fold_opt != 0) && ((uintptr_t)out & 15) == 0) fpa = fused3_plan(B, H, T, W, C, radius, gout, din, true);
    const bool fold = chain && !stats_in && (fold_opt == 2 || fpa.ok || (fold_opt == 1 && n >= (three ? (int64_t)1 << 25 : (int64_t)1 << 22)));
    if (stats_in && fpa.ok) {       // the batch-sharded caller: sums handed in, correction applied at the loads, nothing to fix up
        Fused3AdjArgs fa{};
        fa.gout = gout; fa.out_fwd = out; fa.din = din; fa.mx = max_in; fa.res = stats_ext;
        fa.H = H; fa.T = T; fa.W = W; fa.wt = fpa.wt; fa.ntw = W / fpa.wt; fa.hseg = fpa.hseg; fa.nseg = (H + fpa.hseg - 1) / fpa.hseg; fa.tp = tp;
        return launch_fused3_adj(fpa, fa, C, true, st);
    }
    if (stats_in) {
        res = stats_ext;                                       // the global sums: every kernel below reads res[0], res[1]
    } else if (!fold) {
        if (stats_only) res = stats_ext;
        if (wide) hipLaunchKernelGGL(maxnorm_bwd_partial_v4, dim3(2048), dim3(256), 0, st, gout, out, n / 4, pdot, pcnt);
        else hipLaunchKernelGGL(maxnorm_bwd_partial, dim3((unsigned)nb), dim3(256), 0, st, gout, out, n, pdot, pcnt);
        if ((rc = launch_status("maxnorm_bwd_partial"))) return rc;
        hipLaunchKernelGGL(maxnorm_bwd_combine, dim3(1), dim3(1024), 0, st, (const float*)pdot, (const float*)pcnt, nparts, res);
        if ((rc = launch_status("maxnorm_bwd_combine"))) return rc;
        if (stats_only) return 0;
    }
    if (chain) {
        {
            WalkArgs wa{};
            wa.tp = tp; wa.out_fwd = out; wa.mx = max_in; wa.res = res; wa.ties = recs;
            const int first = fold ? WALK_ADJS : WALK_ADJX;
            const AxisPlan& p1 = three ? ph : pt;               // the stage that reads gout and the forward output
            int nrec = (int)((n / (three ? H : T) / p1.vw + 255) / 256);
            // what follows the last stage when the statistics were folded into the first: the sparse fix-up, then the
            // dense chain that only runs when the fix-up found too many arg-max elements
            auto finish = [&](const Fused3Plan* fused = nullptr) -> int {
                if (!fold) return 0;
                int* dense = reinterpret_cast<int*>(scal + 7);
                FixupArgs fa{};
                fa.ties = recs; fa.nrec = nrec; fa.mx = max_in; fa.res = res; fa.dense = dense; fa.din = din; fa.na = na; fa.tp = tp;
                for (int i = 0; i < na; ++i) { fa.len[i] = ax[i].len; fa.stride[i] = ax[i].stride; }
                hipLaunchKernelGGL(maxnorm_bwd_fixup, dim3(1), dim3(1024), 0, st, fa);
                int rc2 = launch_status("maxnorm_bwd_fixup");
                if (rc2) return rc2;
                if (fused) {    // dense fallback of the fused walk: the same walk once more, guarded, with the sums the fix-up has just
                                // written and the correction applied at the loads (one guarded launch instead of three)
                    Fused3AdjArgs fx{};
                    fx.gout = gout; fx.out_fwd = out; fx.din = din; fx.mx = max_in; fx.res = res; fx.run_if = dense;
                    fx.H = H; fx.T = T; fx.W = W; fx.wt = fused->wt; fx.ntw = W / fused->wt; fx.hseg = fused->hseg;
                    fx.nseg = (H + fused->hseg - 1) / fused->hseg; fx.tp = tp;
                    return launch_fused3_adj(*fused, fx, C, true, st);
                }
                // the round-2 chain with the sums the fix-up has just written: H^T (+ normalisation adjoint) gout -> din,
                // W^T din -> tmp, T^T tmp -> din;  temporal only: T^T (+ normalisation adjoint) gout -> din
                WalkArgs wd{};
                wd.tp = tp; wd.out_fwd = out; wd.mx = max_in; wd.res = res; wd.run_if = dense;
                wd.in = gout; wd.out = din; wd.L = three ? H : T; wd.S = three ? (int64_t)T * WC : WC;
                if ((rc2 = launch_axis(WALK_ADJX, wd, radius, n, p1, st))) return rc2;
                if (!three) return 0;
                hipLaunchKernelGGL(conv_axis_adjoint_if, dim3((unsigned)std::min<int64_t>(nb, 2048)), dim3(256), 0, st,
                                   (const float*)din, tmp, n, W, (int64_t)C, tp, (const int*)dense);
                if ((rc2 = launch_status("conv_axis_adjoint_if"))) return rc2;
                wd.in = tmp; wd.out = din; wd.L = T; wd.S = WC;
                if ((rc2 = launch_axis(WALK_ADJ, wd, radius, n, pt, st))) return rc2;
                return 0;
            };
            if (!three) {
                wa.in = gout; wa.out = din; wa.L = T; wa.S = WC;
                if ((rc = launch_axis(first, wa, radius, n, pt, st))) return rc;
                return finish();
            }
            {               // all three adjoint stages in one pass (smooth_fused3_adj), the statistics gathered on the way
                const Fused3Plan& fp = fpa;
                if (fp.ok) {
                    Fused3AdjArgs fa{};
                    fa.gout = gout; fa.out_fwd = out; fa.din = din; fa.mx = max_in; fa.ties = recs;
                    fa.H = H; fa.T = T; fa.W = W; fa.wt = fp.wt; fa.ntw = W / fp.wt; fa.hseg = fp.hseg;
                    fa.nseg = (H + fp.hseg - 1) / fp.hseg; fa.tp = tp;
                    if ((rc = launch_fused3_adj(fp, fa, C, false, st))) return rc;
                    nrec = (int)fp.grid;
                    return finish(&fp);
                }
            }
            // H^T (+ normalisation adjoint): gout -> din;  W^T: din -> tmp;  T^T: tmp -> din
            wa.in = gout; wa.out = twp ? tmp : din; wa.L = H; wa.S = (int64_t)T * WC;
            if ((rc = launch_axis(first, wa, radius, n, ph, st))) return rc;
            if (twp) {                                              // W^T and T^T in one pass: tmp -> din
                if ((rc = launch_tw_plane(tmp, din, n, T, W, C, radius, true, tp, st))) return rc;
                return finish();
            }
            if (w1) {
                if ((rc = launch_w1(din, tmp, n, W, radius, true, tp, st))) return rc;
            } else if (wrow) {
                if ((rc = launch_wrow(din, tmp, n, W, C, radius, true, tp, st))) return rc;
            } else {
                float* one = scal + 4;                          // scalar slots behind {dot, ties}: {1, 0, 0}
                hipLaunchKernelGGL(set_unit_scalars, dim3(1), dim3(1), 0, st, one);
                if ((rc = launch_status("set_unit_scalars"))) return rc;
                PlaneArgs pa{};
                pa.in = din; pa.out_fwd = out; pa.out = tmp; pa.mx = one; pa.res = one + 1;   // max = 1, no ties: plain W^T
                pa.B = B; pa.H = H; pa.T = T; pa.W = W; pa.C = C; pa.axes = KCCOT_SMOOTH_W; pa.tp = tp;
                pa.hseg = plane_hseg(B, H, false);
                if ((rc = launch_plane(pa, radius, true, dim3((H + pa.hseg - 1) / pa.hseg, B), st))) return rc;
            }
            wa.in = tmp; wa.out = din; wa.L = T; wa.S = WC;
            if ((rc = launch_axis(WALK_ADJ, wa, radius, n, pt, st))) return rc;
            return finish();
        }
    }
    hipLaunchKernelGGL(maxnorm_bwd_apply, dim3((unsigned)nb), dim3(256), 0, st, gout, out, n, max_in, (const float*)res, ds);
    if ((rc = launch_status("maxnorm_bwd_apply"))) return rc;
    const float* src = ds;
    for (int i = na - 1; i >= 0; --i) {
        float* dst = (src == tmp) ? din : tmp;
        hipLaunchKernelGGL(conv_axis, dim3((unsigned)nb), dim3(256), 0, st, src, dst, n, ax[i].len, ax[i].stride, tp, 1,
                           (float*)nullptr);
        if ((rc = launch_status("conv_axis(adjoint)"))) return rc;
        src = dst;
    }
    return 0;
}

extern "C" int kccot_smooth_bwd_f32(const float* gout, const float* out, const float* max_in, int B, int H, int T,
                                    int W, int C, float sigma, int radius, unsigned flags, float* din, void* ws,
                                    size_t ws_bytes, kccot_stream_t stream) {
    if (flags & (KCCOT_SMOOTH_STATS_ONLY | KCCOT_SMOOTH_EXTERNAL_STATS))
        return fail(KCCOT_EINVAL, "smooth_bwd: the stats flags belong to kccot_smooth_bwd_sharded_f32");
    return smooth_bwd_impl(gout, out, max_in, nullptr, B, H, T, W, C, sigma, radius, flags, din, ws, ws_bytes, stream);
}

extern "C" int kccot_smooth_bwd_sharded_f32(const float* gout, const float* out, const float* max_in, float* stats_inout,
                                            int B, int H, int T, int W, int C, float sigma, int radius, unsigned flags,
                                            float* din, void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!(flags & (KCCOT_SMOOTH_STATS_ONLY | KCCOT_SMOOTH_EXTERNAL_STATS)))
        return fail(KCCOT_EINVAL, "smooth_bwd_sharded: give KCCOT_SMOOTH_STATS_ONLY or KCCOT_SMOOTH_EXTERNAL_STATS");
    return smooth_bwd_impl(gout, out, max_in, stats_inout, B, H, T, W, C, sigma, radius, flags, din, ws, ws_bytes, stream);
}
