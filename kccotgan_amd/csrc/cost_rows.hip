// Row block of the loss's three cost matrices for a batch-sharded rank, on the matrix pipe (round 3).
//
// Rank g of the sharded loss (kccotgan_amd/dist.py) owns samples I = [row_begin, row_begin + m) of
// the all-gathered batch and builds C_xy[I,:], C_xx[I,:], C_yy[I,:] (gan_utils.py:221-223).  Until round 2 that ran on the
// direct-difference VALU kernel (9.4 ms per rank at configs[4]).  Here the rank forms the Gram ROW BLOCK
//        G_I = [X_I ; E_I] [X ; E]^T            (2 m x 2 B,   E = fake - real)
// with the exact three-way bf16 split of cost_tile256.hip (same staging, same fragment layout: gram_q.h) and combines
// it in fp64 with the pair-difference identities of cost_mfma.hip.  Those also need the diagonal Gram entries of the
// COLUMN samples, g_jj = x_j.x_j, e_jj = e_j.e_j, x_jj = x_j.e_j for all j -- not part of a row block.  Every rank
// computes them for its own rows from its local shard (kccot_row_norms_f64, one streaming pass over 1/G of the batch,
// before the video all-gather) and the ranks all-gather 3 B doubles.
//
// Kernel: one workgroup (4 waves, one per SIMD) per (256 stack rows of the column side, K-chunk): A = the rank's 2 m
// rows (m = 32 or 64: AM = 64 or 128 LDS rows, E_I = F_I - X_I formed in registers), B = 256 stack rows as two 128-row
// halves, each either X rows or E rows (E = F - X formed in registers: a second row stream -- the rows of X it needs are
// the ones the X-type workgroups of the same K-range stream through the same L2).  No symmetry to exploit: a row block is
// 1/G of TWICE the symmetric Gram, so a rank does 2/G of the single-GPU contraction.
//
//   rows_gram<AM, E0, E1, RAGGED>   partial tiles [panel][chunk][AM][256] fp32
//   rows_gram_reduce                fp64 sum over the chunks (fixed order)
//   rows_gram_finalize              distances from the Gram row block + the gathered norms, scale, causal term
//   row_norms                       g_ii, e_ii, x_ii of a range of rows in fp64
#include "common.h"
#include "cost_internal.h"
#include "options.h"
#include "gram_q.h"

namespace kccot {

constexpr int RB = 256;                    // column-side stack rows per workgroup

struct RowsArgs {
    const float* x;        // gathered real [B,K]
    const float* f;        // gathered fake [B,K]
    int B, m, row_begin;
    int q0, nq;            // this launch's column panels [q0, q0 + nq)
    int nchunk;
    int64_t K, chunk;
    float* part;           // [npan][nchunk][AM * RB]
};

template <int AM, bool E0, bool E1, bool RAGGED>
__global__ __launch_bounds__(256) void rows_gram(RowsArgs a) {
    constexpr int NROWS = AM + RB, PLANE = NROWS * QROWB, SLOT = 3 * PLANE;
    constexpr int AP = AM / 32;                          // A passes: AP / 2 of X_I rows, then AP / 2 of F_I rows
    constexpr int NT = AM == 128 ? 8 : 4;                // accumulator tiles per wave
    __shared__ __attribute__((aligned(16))) unsigned char zs0[SLOT];
    __shared__ __attribute__((aligned(16))) unsigned char zs1[SLOT];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // XCD-aware map as gram_q256: XCD x = id % 8 takes the chunks x, x + 8, ..., within a chunk the panels in order
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int panel = __builtin_amdgcn_readfirstlane(a.q0 + slot % a.nq);
    const int chunk_id = __builtin_amdgcn_readfirstlane((slot / a.nq) * 8 + xcd);
    const int64_t K = a.K;
    const int64_t kbeg = (int64_t)chunk_id * a.chunk;
    if (kbeg >= K) return;                               // an empty trailing chunk
    const int64_t kend = (kbeg + a.chunk < K) ? kbeg + a.chunk : K;
    const int ng = (int)((kend - kbeg + QG - 1) / QG);

    // ---- staging role: thread = (row rr + 32 p of its panel part, columns 4 q .. 4 q + 3 of the granule)
    const int q = t & 7, rr = t >> 3;
    const int r0 = panel * RB - a.B, r1 = r0 + 128;      // E-type halves: first sample of the half (X-type: unused)
    const float* ax = a.x + (int64_t)a.row_begin * K;
    const float* af = a.f + (int64_t)a.row_begin * K;
    const float* b0m = E0 ? a.f + (int64_t)r0 * K : a.x + (int64_t)panel * RB * K;
    const float* b1m = E1 ? a.f + (int64_t)r1 * K : a.x + ((int64_t)panel * RB + 128) * K;
    const float* b0s = a.x + (int64_t)(E0 ? r0 : 0) * K;
    const float* b1s = a.x + (int64_t)(E1 ? r1 : 0) * K;
    const auto rax = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ax), 0, 0xFFFFFFFFu, 0x00020000);
    const auto raf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(af), 0, 0xFFFFFFFFu, 0x00020000);
    const auto rb0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b0m), 0, 0xFFFFFFFFu, 0x00020000);
    const auto rb1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b1m), 0, 0xFFFFFFFFu, 0x00020000);
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b0s), 0, 0xFFFFFFFFu, 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b1s), 0, 0xFFFFFFFFu, 0x00020000);
    const unsigned rstep = (unsigned)(32 * K * 4);       // bytes; 3 * rstep + a row < 4 GiB: K <= 2^22 (host)
    const unsigned lrow = (unsigned)((int64_t)rr * K * 4);
    const int64_t kmax = K - 4;
    const int woff = rr * QROWB + 4 * q;

    float4 GA[AP], GB[8], GS[8];                         // A rows; column-side rows; their subtrahends (E halves only)
    float2 carry[AP + 8];
    auto ld = [&](auto r, unsigned vo, int p) {
        return __builtin_bit_cast(float4, (qu32x4)__builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, (int)(p * rstep), 0));
    };
    auto issue = [&](int g) {
        int64_t k = kbeg + (int64_t)g * QG + 4 * q;
        k = k < kmax ? k : kmax;                          // unconditional loads, column clamped into the row
        const unsigned vo = lrow + (unsigned)(k * 4);
#pragma unroll
        for (int p = 0; p < AP / 2; ++p) { GA[p] = ld(rax, vo, p); GA[AP / 2 + p] = ld(raf, vo, p); }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            GB[p] = ld(rb0, vo, p);
            GB[4 + p] = ld(rb1, vo, p);
            if (E0) GS[p] = ld(rs0, vo, p);
            if (E1) GS[4 + p] = ld(rs1, vo, p);
        }
    };
    auto value = [&](int p) {                             // staged row p: A rows 0 .. AP - 1, then the eight column passes
        float4 v;
        if (p < AP) {
            v = GA[p];
            if (p >= AP / 2) { const float4 s = GA[p - AP / 2]; v.x -= s.x; v.y -= s.y; v.z -= s.z; v.w -= s.w; }   // E_I = F_I - X_I
        } else {
            const int pb = p - AP;
            v = GB[pb];
            if (pb < 4 ? E0 : E1) { const float4 s = GS[pb]; v.x -= s.x; v.y -= s.y; v.z -= s.z; v.w -= s.w; }
        }
        return v;
    };
    auto emit_even = [&](int g, unsigned char* zs) {
        const bool kok = kbeg + (int64_t)g * QG + 4 * q + 4 <= kend;
#pragma unroll
        for (int p = 0; p < AP + 8; ++p) {
            float4 v = value(p);
            if (RAGGED) { v.x = kok ? v.x : 0.f; v.y = kok ? v.y : 0.f; v.z = kok ? v.z : 0.f; v.w = kok ? v.w : 0.f; }
            gq_split_store<PLANE>(zs, woff + 32 * p * QROWB, v.x, v.y);
            carry[p] = make_float2(v.z, v.w);
        }
    };
    auto emit_odd = [&](unsigned char* zs) {
#pragma unroll
        for (int p = 0; p < AP + 8; ++p) gq_split_store<PLANE>(zs, woff + 32 * p * QROWB, carry[p].x, carry[p].y);
    };

    // ---- consuming role.  AM = 128: wave (wr, wc) = A rows 64 wr .. x columns 128 wc .. (2 x 4 tiles);
    //                       AM = 64:  wave w = all 64 A rows x columns 64 w .. (2 x 2 tiles)
    constexpr int NB = NT / 2;
    const int lo = (lane & 31) * QROWB + 16 * (lane >> 5);
    const int wr = AM == 128 ? wave >> 1 : 0, wc = AM == 128 ? wave & 1 : wave;
    const int aoff = (64 * wr) * QROWB + lo;
    const int boff = (AM + 32 * NB * wc) * QROWB + lo;
    qf32x16 acc[NT];
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t2][r] = 0.f;
    auto step = [&](const unsigned char* zs) {
        QFrag bf[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) bf[j] = gq_frag<PLANE>(zs, boff + j * 32 * QROWB);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const QFrag af2 = gq_frag<PLANE>(zs, aoff + i * 32 * QROWB);
#pragma unroll
            for (int j = 0; j < NB; ++j) gq_mfma6(acc[NB * i + j], af2, bf[j]);
        }
    };
    auto interleave = [&](bool with_loads) {              // see gram_q256: one wave per SIMD, the staging rides in the MFMAs' shadow
        constexpr int NL = AP + 8 + (E0 ? 4 : 0) + (E1 ? 4 : 0);
        constexpr int EVERY = (6 * NT) / NL > 0 ? (6 * NT) / NL : 1;
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int m2 = 0; m2 < 6 * NT; ++m2) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, AM == 128 ? 4 : 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (with_loads && (m2 % EVERY) == EVERY - 1) __builtin_amdgcn_sched_group_barrier(0x020, (NL + 6 * NT - 1) / (6 * NT), 0);
        }
    };
    auto granule = [&](int g) {
        step(zs0);
        emit_odd(zs1);
        interleave(false);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        __builtin_amdgcn_sched_barrier(0);
        step(zs1);
        emit_even(g + 1, zs0);
        issue(g + 2);
        interleave(true);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    issue(0);
    emit_even(0, zs0);
    issue(1);
    lds_barrier();
    for (int g = 0; g < ng; ++g) granule(g);

    float* o = a.part + ((int64_t)panel * a.nchunk + chunk_id) * (AM * RB) + (64 * wr + 4 * (lane >> 5)) * RB + 32 * NB * wc + (lane & 31);
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2) {
        float* ot = o + (32 * (t2 / NB)) * RB + 32 * (t2 % NB);
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[((r & 3) + 8 * (r >> 2)) * RB] = acc[t2][r];
    }
}

// fp64 sum over the chunks, fixed order; eight loads in flight.  grid (elems / 256, panels).  accumulate != 0: added to what
// gsum holds (a caller that feeds the columns in several calls, kccot_pairwise_cost3_rows_gram_sums_f64).
__global__ __launch_bounds__(256) void rows_gram_reduce(const float* __restrict__ part, int nstride, int nchunk, int elems,
                                                        double* __restrict__ gsum, int accumulate) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const float* p = part + (int64_t)blockIdx.y * nstride * elems + e;
    double s = 0.0;
    int c = 0;
    for (; c + 8 <= nchunk; c += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[(int64_t)(c + i) * elems];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (double)v[i];
    }
    for (; c < nchunk; ++c) s += (double)p[(int64_t)c * elems];
    double* g = gsum + (int64_t)blockIdx.y * elems + e;
    *g = accumulate ? *g + s : s;
}

struct RowsFin {
    const double* gsum;      // [npan][2 m][256]
    const double* norms;     // [B][3]: x.x, e.e, x.e of every sample (all ranks' rows, gathered)
    int B, m, row_begin;
    float* out[3];           // [m, B] each
    const float* h[3];       // rows: already offset to row_begin
    const float* M[3];
    float sc;
    int T, J;
};

// One 16 x 16 output tile per block; blockIdx.z = problem (xy, xx, yy).  The formulas of gram_finalize (cost_mfma.hip) with
// the diagonal Gram entries taken from `norms`.
__global__ __launch_bounds__(256) void rows_gram_finalize(RowsFin f) {
    __shared__ __attribute__((aligned(16))) float sh[CAUSAL_TILE * CAUSAL_PITCH];
    __shared__ __attribute__((aligned(16))) float sm[CAUSAL_TILE * CAUSAL_PITCH];
    const int p = blockIdx.z, B = f.B, m = f.m;
    const int i0 = blockIdx.y * CAUSAL_TILE, j0 = blockIdx.x * CAUSAL_TILE;
    const int i = i0 + (threadIdx.x >> 4), j = j0 + (threadIdx.x & 15);      // m % 16 == 0, B % 16 == 0: always in range
    const int am = 2 * m;
    auto G = [&](int arow, int srow) { return f.gsum[((int64_t)(srow >> 8) * am + arow) * RB + (srow & 255)]; };
    const int gi = f.row_begin + i;
    const bool diag = gi == j;
    const double g_ii = f.norms[3 * gi], e_ii = f.norms[3 * gi + 1], x_ii = f.norms[3 * gi + 2];
    const double g_jj = f.norms[3 * j], e_jj = f.norms[3 * j + 1], x_jj = f.norms[3 * j + 2];
    const double g_ij = G(i, j), e_ij = G(m + i, B + j);
    const double x_ij = diag ? x_jj : G(i, B + j);            // x_i . e_j (the diagonal one exactly the norm pass's)
    const double x_ji = diag ? x_jj : G(m + i, j);            // e_i . x_j
    const double dxx = diag ? 0.0 : g_ii + g_jj - 2.0 * g_ij;
    const double dxy = dxx + e_jj - 2.0 * (x_ij - x_jj);
    const double dee = e_ii + e_jj - 2.0 * e_ij;
    const double dyy = diag ? 0.0 : dxx + dee + 2.0 * (x_ii - x_ij - x_ji + x_jj);
    double D = (p == 1) ? dxx : (p == 0 ? dxy : dyy);
    if (D < 0.0) D = 0.0;
    float c = (float)D * f.sc;
    if (f.h[p]) c += causal_tile16(f.h[p], f.M[p], i0, j0, m, B, f.T, f.J, sh, sm) * f.sc;
    f.out[p][(int64_t)i * B + j] = c;
}

// x.x, e.e, x.e (e = fake - real in fp32, as the Gram kernels form it) of rows [0, rows): workgroup (row, s) sums the
// s-th of `nsplit` column ranges of the row (fp32 products, fp64 sums in a fixed order) into part[row][s][3]; nsplit == 1
// writes the result itself, otherwise row_norms_combine adds the ranges up in order.
__global__ __launch_bounds__(256) void row_norms(const float* __restrict__ x, const float* __restrict__ f, int64_t K,
                                                 int nsplit, double* __restrict__ out) {
    __shared__ double red[3][4];
    const float* xr = x + (int64_t)blockIdx.x * K;
    const float* fr = f + (int64_t)blockIdx.x * K;
    double sx = 0.0, se = 0.0, sxe = 0.0;
    const int64_t K4 = K >> 2;                                 // K % 4 == 0 (host)
    const int64_t per = ((K4 + nsplit - 1) / nsplit + 1023) / 1024 * 1024;
    const int64_t beg = (int64_t)blockIdx.y * per, end = beg + per < K4 ? beg + per : K4;
    for (int64_t i0 = beg + threadIdx.x; i0 < end; i0 += 256 * 4) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + 256 * u;
            const int64_t ic = i < K4 ? i : K4 - 1;
            a[u] = reinterpret_cast<const float4*>(xr)[ic];
            b[u] = reinterpret_cast<const float4*>(fr)[ic];
        }
        float px = 0.f, pe = 0.f, pxe = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + 256 * u >= end) continue;
            const float xs[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, fs[4] = {b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float e = fs[c] - xs[c];
                px = fmaf(xs[c], xs[c], px); pe = fmaf(e, e, pe); pxe = fmaf(xs[c], e, pxe);
            }
        }
        sx += (double)px; se += (double)pe; sxe += (double)pxe;
    }
    sx = wave_sum_d(sx); se = wave_sum_d(se); sxe = wave_sum_d(sxe);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = sx; red[1][w] = se; red[2][w] = sxe; }
    __syncthreads();
    if (threadIdx.x < 3)
        out[((int64_t)blockIdx.x * nsplit + blockIdx.y) * 3 + threadIdx.x] =
            (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}
__global__ __launch_bounds__(256) void row_norms_combine(const double* __restrict__ part, int n3, int nsplit, double* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;              // (row, component)
    if (e >= n3) return;
    const int row = e / 3, c = e - 3 * row;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += part[((int64_t)row * nsplit + k) * 3 + c];
    out[e] = s;
}

static int row_norms_split(int rows) { const int s = 2048 / (rows > 0 ? rows : 1); return s < 1 ? 1 : (s > 32 ? 32 : s); }

// norms of `rows` rows into out [rows][3]; scratch: rows * nsplit * 3 doubles (unused when nsplit == 1)
static int launch_row_norms(const float* real, const float* fake, int rows, int64_t K, double* out, double* scratch, hipStream_t st) {
    const int nsplit = scratch ? row_norms_split(rows) : 1;
    hipLaunchKernelGGL(row_norms, dim3(rows, nsplit), dim3(256), 0, st, real, fake, K, nsplit, nsplit > 1 ? scratch : out);
    int rc = launch_status("row_norms");
    if (rc || nsplit == 1) return rc;
    hipLaunchKernelGGL(row_norms_combine, dim3((3 * rows + 255) / 256), dim3(256), 0, st, (const double*)scratch, 3 * rows, nsplit, out);
    return launch_status("row_norms_combine");
}

// ---- host side ---------------------------------------------------------------------------------------
struct RowsPlan { int npan, nchunk, am; int64_t chunk; size_t part_bytes, gsum_bytes, norm_bytes, ws_bytes; };

static bool rows_shape_ok(int row_count, int B, int64_t K) {
    return (row_count == 32 || row_count == 64) && B % 128 == 0 && B >= 128 && B <= 4096 && K % 4 == 0 && K >= 256 && K <= (1 << 22);
}

static RowsPlan plan_rows(int row_count, int B, int64_t K) {
    RowsPlan pl{};
    pl.am = 2 * row_count;
    pl.npan = (2 * B + RB - 1) / RB;
    const int64_t ngran = (K + QG - 1) / QG;
    const int64_t nmin = ((ngran + Q_MAX_GRAN - 1) / Q_MAX_GRAN + 7) / 8 * 8;     // chunk length bounded by the accumulation (gram_q.h)
    int best = (int)nmin;
    double best_cost = 1e300;
    for (int64_t n = nmin; n <= 2 * nmin; n += 8) {
        const int64_t gpc = (ngran + n - 1) / n;
        const double cost = (double)(((int64_t)pl.npan * n + 255) / 256) * ((double)gpc + 3.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = (int)n; }
    }
    pl.nchunk = best;
    pl.chunk = ((ngran + best - 1) / best) * QG;
    pl.part_bytes = align_up((size_t)pl.npan * pl.nchunk * pl.am * RB * sizeof(float), 256);
    pl.gsum_bytes = align_up((size_t)pl.npan * pl.am * RB * sizeof(double), 256);
    pl.norm_bytes = align_up((size_t)B * 3 * sizeof(double), 256);
    pl.ws_bytes = pl.part_bytes + pl.gsum_bytes + pl.norm_bytes;
    return pl;
}

template <int AM, bool E0, bool E1>
static void launch_rows_gram(const RowsArgs& ra, bool ragged, hipStream_t st) {
    const dim3 grid(ra.nq * ra.nchunk), block(256);
    if (ragged) hipLaunchKernelGGL((rows_gram<AM, E0, E1, true>), grid, block, 0, st, ra);
    else hipLaunchKernelGGL((rows_gram<AM, E0, E1, false>), grid, block, 0, st, ra);
}

}  // namespace kccot

using namespace kccot;

extern "C" int kccot_pairwise_cost3_rows_gram_supported(int row_count, int B, int64_t K) {
    return (rows_shape_ok(row_count, B, K) && opt(OPT_COST_TILED) && !opt(OPT_GRAM_F32)) ? 1 : 0;
}

extern "C" size_t kccot_pairwise_cost3_rows_gram_workspace_bytes(int row_count, int B, int64_t K) {
    return rows_shape_ok(row_count, B, K) ? plan_rows(row_count, B, K).ws_bytes : 0;
}

extern "C" size_t kccot_row_norms_workspace_bytes(int rows) {
    return rows > 0 ? (size_t)rows * row_norms_split(rows) * 3 * sizeof(double) : 0;
}

extern "C" int kccot_row_norms_f64(const float* real, const float* fake, int rows, int64_t K, double* norms_out,
                                   void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!real || !fake || !norms_out) return fail(KCCOT_EINVAL, "row_norms: null pointer");
    if (rows <= 0 || K <= 0 || K % 4 != 0) return fail(KCCOT_EINVAL, "row_norms: bad shape rows=%d K=%lld (K %% 4 == 0)", rows, (long long)K);
    if (((uintptr_t)real | (uintptr_t)fake) % 16) return fail(KCCOT_EINVAL, "row_norms: rows must be 16-byte aligned");
    double* scratch = nullptr;
    if (ws && ws_bytes >= (size_t)rows * row_norms_split(rows) * 3 * sizeof(double)) scratch = static_cast<double*>(ws);
    return launch_row_norms(real, fake, rows, K, norms_out, scratch, (hipStream_t)stream);
}

// stage 1: the fp64 Gram sums of the row block over the K columns given (partial tiles + reduce); stage 2: the three cost
// row blocks from the sums and the norms.  kccot_pairwise_cost3_rows_gram_f32 runs both.
static int rows_gram_sums(const float* real, const float* fake, int B, int64_t K, int row_begin, int row_count, double* gsum,
                          int accumulate, float* part, const RowsPlan& pl, hipStream_t st) {
    int rc;
    const bool ragged = K % QG != 0;
    const int nxx = B / RB, nmix = (B % RB) ? 1 : 0, nee = pl.npan - nxx - nmix;     // column panels: X X | X E | E E
    RowsArgs ra{real, fake, B, row_count, row_begin, 0, 0, pl.nchunk, K, pl.chunk, part};
    const bool big = row_count == 64;
    if (nxx) {
        ra.q0 = 0; ra.nq = nxx;
        if (big) launch_rows_gram<128, false, false>(ra, ragged, st); else launch_rows_gram<64, false, false>(ra, ragged, st);
        if ((rc = launch_status("rows_gram<xx>"))) return rc;
    }
    if (nmix) {
        ra.q0 = nxx; ra.nq = 1;
        if (big) launch_rows_gram<128, false, true>(ra, ragged, st); else launch_rows_gram<64, false, true>(ra, ragged, st);
        if ((rc = launch_status("rows_gram<xe>"))) return rc;
    }
    if (nee) {
        ra.q0 = nxx + nmix; ra.nq = nee;
        if (big) launch_rows_gram<128, true, true>(ra, ragged, st); else launch_rows_gram<64, true, true>(ra, ragged, st);
        if ((rc = launch_status("rows_gram<ee>"))) return rc;
    }
    const int elems = pl.am * RB, nvalid = (int)((K + pl.chunk - 1) / pl.chunk);
    hipLaunchKernelGGL(rows_gram_reduce, dim3(elems / 256, pl.npan), dim3(256), 0, st, (const float*)part, pl.nchunk, nvalid, elems, gsum,
                       accumulate);
    return launch_status("rows_gram_reduce");
}

static int rows_gram_from_sums(const double* gsum, int B, float sc, const float* h_fake, const float* h_real, const float* m_real,
                               const float* m_fake, int T, int J, int row_begin, int row_count, const double* norms,
                               float* C3_rows, hipStream_t st) {
    const int64_t rb = (int64_t)row_count * B, tj = (int64_t)T * J;
    RowsFin f{};
    f.gsum = gsum; f.norms = norms; f.B = B; f.m = row_count; f.row_begin = row_begin; f.sc = sc; f.T = T; f.J = J;
    // gan_utils.py:221-223: xy = (h_fake rows, m_real cols), xx = (h_real, m_real), yy = (h_fake, m_fake)
    f.out[0] = C3_rows; f.out[1] = C3_rows + rb; f.out[2] = C3_rows + 2 * rb;
    f.h[0] = h_fake + row_begin * tj; f.h[1] = h_real + row_begin * tj; f.h[2] = h_fake + row_begin * tj;
    f.M[0] = m_real; f.M[1] = m_real; f.M[2] = m_fake;
    hipLaunchKernelGGL(rows_gram_finalize, dim3(B / CAUSAL_TILE, row_count / CAUSAL_TILE, 3), dim3(256), 0, st, f);
    return launch_status("rows_gram_finalize");
}

static int rows_gram_check(const char* who, const float* real, const float* fake, int B, int64_t K, int row_begin, int row_count) {
    if (!real || !fake) return fail(KCCOT_EINVAL, "%s: null pointer", who);
    if (B <= 0 || K <= 0 || row_begin < 0 || row_count <= 0 || row_begin + row_count > B)
        return fail(KCCOT_EINVAL, "%s: bad shape B=%d K=%lld rows [%d,%d)", who, B, (long long)K, row_begin, row_begin + row_count);
    if (!rows_shape_ok(row_count, B, K) || ((uintptr_t)real | (uintptr_t)fake) % 16)
        return fail(KCCOT_EUNSUPPORTED, "%s: needs 32 or 64 rows, B %% 128 == 0, K %% 4 == 0, 256 <= K <= 2^22 "
                    "(kccot_pairwise_cost3_rows_gram_supported); use kccot_pairwise_cost3_rows_f32", who);
    return 0;
}

extern "C" int kccot_pairwise_cost3_rows_gram_f32(const float* real, const float* fake, int B, int64_t K, float sc,
                                                  const float* h_fake, const float* h_real, const float* m_real,
                                                  const float* m_fake, int T, int J, int row_begin, int row_count,
                                                  const double* norms, float* C3_rows, void* ws, size_t ws_bytes,
                                                  kccot_stream_t stream) {
    int rc = rows_gram_check("pairwise_cost3_rows_gram", real, fake, B, K, row_begin, row_count);
    if (rc) return rc;
    if (!C3_rows) return fail(KCCOT_EINVAL, "pairwise_cost3_rows_gram: null pointer");
    if (!h_fake || !h_real || !m_real || !m_fake) return fail(KCCOT_EINVAL, "pairwise_cost3_rows_gram: all four feature tensors are needed");
    if (T < 1 || J < 1) return fail(KCCOT_EINVAL, "pairwise_cost3_rows_gram: bad feature shape T=%d J=%d", T, J);
    const RowsPlan pl = plan_rows(row_count, B, K);
    if (!ws || ws_bytes < pl.ws_bytes)
        return fail(KCCOT_EWORKSPACE, "pairwise_cost3_rows_gram: workspace %zu < required %zu", ws_bytes, pl.ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    float* part = static_cast<float*>(ws);
    double* gsum = reinterpret_cast<double*>(static_cast<char*>(ws) + pl.part_bytes);
    double* own_norms = reinterpret_cast<double*>(static_cast<char*>(ws) + pl.part_bytes + pl.gsum_bytes);
    if (!norms) {   // no gathered norms given: one pass over the whole batch (a single-process caller; the sharded one gathers them)
        if ((rc = launch_row_norms(real, fake, B, K, own_norms, reinterpret_cast<double*>(part), st))) return rc;   // scratch: the partial-tile area, not in use yet
        norms = own_norms;
    }
    if ((rc = rows_gram_sums(real, fake, B, K, row_begin, row_count, gsum, 0, part, pl, st))) return rc;
    return rows_gram_from_sums(gsum, B, sc, h_fake, h_real, m_real, m_fake, T, J, row_begin, row_count, norms, C3_rows, st);
}

// The same in two stages for a caller that receives the COLUMNS in pieces (a batch-sharded caller that all-gathers the
// videos in K-chunks and overlaps the transfers with this accumulation): `real` / `fake` are [B, K] arrays of ONE column
// range of all samples; the fp64 sums of the ranges add up (accumulate = 0 for the first range, 1 afterwards; stream order
// fixes the order of the additions).  The row norms must cover ALL columns (they are gathered separately).
extern "C" size_t kccot_pairwise_cost3_rows_gram_sums_count(int row_count, int B) {
    if (!(row_count == 32 || row_count == 64) || B < 128 || B % 128) return 0;
    return (size_t)((2 * B + RB - 1) / RB) * (2 * row_count) * RB;
}

extern "C" int kccot_pairwise_cost3_rows_gram_sums_f64(const float* real, const float* fake, int B, int64_t K, int row_begin,
                                                       int row_count, double* gsum, int accumulate, void* ws, size_t ws_bytes,
                                                       kccot_stream_t stream) {
    int rc = rows_gram_check("pairwise_cost3_rows_gram_sums", real, fake, B, K, row_begin, row_count);
    if (rc) return rc;
    if (!gsum) return fail(KCCOT_EINVAL, "pairwise_cost3_rows_gram_sums: null pointer");
    const RowsPlan pl = plan_rows(row_count, B, K);
    if (!ws || ws_bytes < pl.part_bytes)
        return fail(KCCOT_EWORKSPACE, "pairwise_cost3_rows_gram_sums: workspace %zu < required %zu", ws_bytes, pl.part_bytes);
    return rows_gram_sums(real, fake, B, K, row_begin, row_count, gsum, accumulate, static_cast<float*>(ws), pl, (hipStream_t)stream);
}

extern "C" int kccot_pairwise_cost3_rows_gram_from_sums_f32(const double* gsum, int B, float sc, const float* h_fake,
                                                            const float* h_real, const float* m_real, const float* m_fake,
                                                            int T, int J, int row_begin, int row_count, const double* norms,
                                                            float* C3_rows, kccot_stream_t stream) {
    if (!gsum || !norms || !C3_rows || !h_fake || !h_real || !m_real || !m_fake)
        return fail(KCCOT_EINVAL, "pairwise_cost3_rows_gram_from_sums: null pointer");
    if (!(row_count == 32 || row_count == 64) || B < 128 || B % 128 || row_begin < 0 || row_begin + row_count > B || T < 1 || J < 1)
        return fail(KCCOT_EINVAL, "pairwise_cost3_rows_gram_from_sums: bad shape B=%d rows [%d,%d) T=%d J=%d", B, row_begin,
                    row_begin + row_count, T, J);
    return rows_gram_from_sums(gsum, B, sc, h_fake, h_real, m_real, m_fake, T, J, row_begin, row_count, norms, C3_rows, (hipStream_t)stream);
}
