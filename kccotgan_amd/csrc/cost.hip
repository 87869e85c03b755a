// Pairwise cost matrices of the causal-OT loss (replaces gan_utils.py:6-72 of the reference).
//
//   C[i,j] = sc * sum_k (x[i,k]-y[j,k])^2 + sc * sum_{t<T-1,q} h[i,t,q]*(M[j,t+1,q]-M[j,t,q])
//
// The contraction length K = T*H*W*C is 1e5..2e6 while the output is only B x B (B = 8..512), so
// the work is split along K: every workgroup owns one 64x64 output tile over one K-chunk and
// writes a partial tile; `cost_finalize` sums the chunks (fp64 accumulate, fixed order, so the
// result is run-to-run deterministic), applies the scale, mirrors the lower triangle of x==y
// problems and adds the causal term (a [(T-1)J]-long contraction, negligible work).
//
// Two partial-tile kernels:
//   cost_direct_partial : VALU, exact direct-difference form (x-y)^2 as the reference computes
//                         it (gan_utils.py:16).  Any shape.
//   gram128_partial     : (cost_mfma.hip) stacked-Gram form on the f32 MFMA pipe, for operands
//                         of at most 64 rows (the BASELINE configs[1] shape).
#include "common.h"
#include "cost_internal.h"
#include "options.h"

namespace kccot {

// ------------------------------------------------------------------------------------------
// direct-difference partial tiles
// block = 256 threads as 16 (ty) x 16 (tx); thread owns rows ty+16a (a < RA), cols tx+16b (b < 4) of a
// (16 RA) x 64 output tile: RA = 8 for operands above 64 rows (12 ds_read_b128 feed 128 sub+fma pairs:
// the LDS port stays below its 128 B/clk while the VALU runs packed), RA = 4, 2, 1 for 64 / 32 / 16 rows
// (the row blocks of the batch-sharded caller), so that short operands do not pay for empty rows.
// The k loop is packed two wide (v_pk_add_f32 / v_pk_fma_f32 on the .xy and .zw halves of the staged
// float4s): the even and the odd k of a chunk accumulate separately and are added at the end.
// LDS tiles are [rows][KT+4] floats: the +4 pitch makes the 16 rows a 16-lane ds_read_b128 group
// touches land on 16 distinct 4-bank slots (36*r mod 64 is a permutation of multiples of 4).
// ------------------------------------------------------------------------------------------
constexpr int DT = 64;       // tile width (columns); rows per tile = 16 * RA
constexpr int DKT = 32;      // k-tile
constexpr int DPITCH = DKT + 4;
typedef float cost_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float4 load4_guard(const float* __restrict__ row, int64_t k, int64_t kend,
                                              bool rowok, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!rowok) return v;
    if (vec && k + 4 <= kend) return *reinterpret_cast<const float4*>(row + k);
    if (k + 0 < kend) v.x = row[k + 0];
    if (k + 1 < kend) v.y = row[k + 1];
    if (k + 2 < kend) v.z = row[k + 2];
    if (k + 3 < kend) v.w = row[k + 3];
    return v;
}

template <int RA>
__global__ __launch_bounds__(256) void cost_direct_partial(CostBatch cb, int64_t K, int64_t chunk,
                                                           int vec_ok) {
    constexpr int TROWS = 16 * RA;
    constexpr int NXL = (TROWS * 8 + 255) / 256;     // staged float4 per thread: x tile (4, 2, 1, 1) ...
    constexpr int NYL = 2;                           // ... and y tile (64 rows x 8)
    __shared__ __attribute__((aligned(16))) float xs[TROWS * DPITCH];
    __shared__ __attribute__((aligned(16))) float ys[DT * DPITCH];

    // decode (problem, tile_i, tile_j) from blockIdx.y
    int tile = blockIdx.y, p = 0;
    for (; p < cb.nprob; ++p) {
        int nt = cb.p[p].tiles_i * cb.p[p].tiles_j;
        if (tile < nt) break;
        tile -= nt;
    }
    if (p >= cb.nprob) return;
    const CostProb& pr = cb.p[p];
    const int ti = tile / pr.tiles_j, tj = tile % pr.tiles_j;
    const int i0 = ti * TROWS, j0 = tj * DT;
    // x == y problems: cost_finalize mirrors every 64-block below the block diagonal from above it, so a tile
    // whose columns all lie in 64-blocks left of its first row's block is never read
    if (pr.same && tj < i0 / DT) return;
    const int64_t kbeg = (int64_t)blockIdx.x * chunk;
    const int64_t kend = (kbeg + chunk < K) ? kbeg + chunk : K;
    if (kbeg >= kend) return;

    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    // staging assignment: float4 slot q = t + 256*m  ->  row q/8, column group q%8
    const int sc4 = (t & 7) * 4;
    const bool vec = vec_ok != 0;
    const float* xr[NXL];
    bool xok[NXL];
    int xrow[NXL];
#pragma unroll
    for (int m = 0; m < NXL; ++m) {
        xrow[m] = (t >> 3) + 32 * m;
        xok[m] = xrow[m] < TROWS && i0 + xrow[m] < pr.Bx;
        xr[m] = pr.x + (int64_t)(i0 + xrow[m]) * K;
    }
    const float* yr[NYL];
    bool yok[NYL];
#pragma unroll
    for (int m = 0; m < NYL; ++m) {
        yok[m] = j0 + (t >> 3) + 32 * m < pr.By;
        yr[m] = pr.y + (int64_t)(j0 + (t >> 3) + 32 * m) * K;
    }

    cost_f2 acc[RA][4];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = cost_f2{0.f, 0.f};

    float4 nx[NXL], ny[NYL];
#pragma unroll
    for (int m = 0; m < NXL; ++m) nx[m] = load4_guard(xr[m], kbeg + sc4, kend, xok[m], vec);
#pragma unroll
    for (int m = 0; m < NYL; ++m) ny[m] = load4_guard(yr[m], kbeg + sc4, kend, yok[m], vec);

    for (int64_t k0 = kbeg; k0 < kend; k0 += DKT) {
#pragma unroll
        for (int m = 0; m < NXL; ++m)
            if (xrow[m] < TROWS) *reinterpret_cast<float4*>(&xs[xrow[m] * DPITCH + sc4]) = nx[m];
#pragma unroll
        for (int m = 0; m < NYL; ++m) *reinterpret_cast<float4*>(&ys[((t >> 3) + 32 * m) * DPITCH + sc4]) = ny[m];
        __syncthreads();
        const int64_t kn = k0 + DKT;
        if (kn < kend) {  // prefetch the next k-tile while this one is consumed
#pragma unroll
            for (int m = 0; m < NXL; ++m) nx[m] = load4_guard(xr[m], kn + sc4, kend, xok[m], vec);
#pragma unroll
            for (int m = 0; m < NYL; ++m) ny[m] = load4_guard(yr[m], kn + sc4, kend, yok[m], vec);
        }
#pragma unroll
        for (int kk = 0; kk < DKT; kk += 4) {
            float4 xv[RA], yv[4];
#pragma unroll
            for (int a = 0; a < RA; ++a)
                xv[a] = *reinterpret_cast<const float4*>(&xs[(ty + 16 * a) * DPITCH + kk]);
#pragma unroll
            for (int b = 0; b < 4; ++b)
                yv[b] = *reinterpret_cast<const float4*>(&ys[(tx + 16 * b) * DPITCH + kk]);
#pragma unroll
            for (int a = 0; a < RA; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const cost_f2 xlo = {xv[a].x, xv[a].y}, xhi = {xv[a].z, xv[a].w};
                    const cost_f2 ylo = {yv[b].x, yv[b].y}, yhi = {yv[b].z, yv[b].w};
                    const cost_f2 d0 = xlo - ylo, d1 = xhi - yhi;
                    acc[a][b] = __builtin_elementwise_fma(d0, d0, acc[a][b]);
                    acc[a][b] = __builtin_elementwise_fma(d1, d1, acc[a][b]);
                }
        }
        __syncthreads();
    }

    float* part = pr.partial + (int64_t)blockIdx.x * pr.Bx * pr.By;
#pragma unroll
    for (int a = 0; a < RA; ++a) {
        const int i = i0 + ty + 16 * a;
        if (i >= pr.Bx) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + tx + 16 * b;
            if (j < pr.By) part[(int64_t)i * pr.By + j] = acc[a][b].x + acc[a][b].y;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Many K-chunks (few output tiles: the row blocks of the batch-sharded caller): sixteen lanes share one output
// element, lane q sums chunks q, q+16, ... in fp64 with eight loads in flight, the sixteen sums are added in lane
// order, and the total replaces the element's chunk-0 value -- cost_finalize then runs with one chunk.
// (One thread per element walking 341 chunks was 86 us of latency for a 8 x 64 row block; this is ~5 us.)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cost_reduce_chunks(CostBatch cb, int nchunk) {
    const CostProb& pr = cb.p[blockIdx.y];
    const int64_t n = (int64_t)pr.Bx * pr.By;
    const int lane16 = threadIdx.x & 15;
    const int64_t e = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t off = e < n ? e : n - 1;                       // clamped: every lane takes part in the shuffles
    const float* pp = pr.partial + off;
    double s = 0.0;
    int c = lane16;
    for (; c + 16 * 7 < nchunk; c += 16 * 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = pp[(int64_t)(c + 16 * q) * n];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += (double)v[q];
    }
    for (; c < nchunk; c += 16) s += (double)pp[(int64_t)c * n];
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += __shfl(s, (threadIdx.x & 48) + q, 64);   // lane order: fixed
    if (lane16 == 0 && e < n) pr.partial[off] = (float)tot;
}

// ------------------------------------------------------------------------------------------
// finalize: one thread per output element
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cost_finalize(CostBatch cb, int nchunk, float sc, int T, int J) {
    __shared__ __attribute__((aligned(16))) float sh[CAUSAL_TILE * CAUSAL_PITCH];
    __shared__ __attribute__((aligned(16))) float sm[CAUSAL_TILE * CAUSAL_PITCH];
    const int p = blockIdx.z;
    const CostProb& pr = cb.p[p];
    const int i0 = blockIdx.y * CAUSAL_TILE, j0 = blockIdx.x * CAUSAL_TILE;
    if (i0 >= pr.Bx || j0 >= pr.By) return;   // block-uniform
    const int i = i0 + (threadIdx.x >> 4), j = j0 + (threadIdx.x & 15);
    const bool ok = i < pr.Bx && j < pr.By;
    const int64_t n = (int64_t)pr.Bx * pr.By;
    float l2 = 0.f;
    if (ok && !(pr.same && i == j)) {   // (x-x)^2 summed is exactly 0 in the reference (gan_utils.py:16)
        // x==y problems only computed tiles with tile_j >= tile_i
        const bool mirror = pr.same && (j / pr.tile) < (i / pr.tile);
        const int64_t off = mirror ? (int64_t)j * pr.By + i : (int64_t)i * pr.By + j;
        // chunk order is fixed (run-to-run deterministic); sixteen loads are issued before the first add so that the
        // loop costs nchunk/16 memory round trips, not nchunk (341 chunks of a row block: 86 -> 9 us)
        double s = 0.0;
        const float* pp = pr.partial + off;
        int c = 0;
        for (; c + 16 <= nchunk; c += 16) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = pp[(int64_t)(c + q) * n];
#pragma unroll
            for (int q = 0; q < 16; ++q) s += (double)v[q];
        }
        for (; c < nchunk; ++c) s += (double)pp[(int64_t)c * n];
        l2 = (float)s * sc;
    }
    float c = l2;
    if (pr.h1) c += causal_tile16(pr.h1, pr.M1, i0, j0, pr.Bx, pr.By, T, J, sh, sm) * sc;
    if (pr.h2) c += causal_tile16(pr.h2, pr.M2, i0, j0, pr.Bx, pr.By, T, J, sh, sm) * sc;
    if (ok) pr.out[(int64_t)i * pr.By + j] = c;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// rows of a direct tile as a multiple of 16: 8 (128 rows) above 64-row operands, else the smallest of 4, 2, 1 that
// covers the operand
static int direct_ra(int nprob, const int* Bx) {
    int mx = 0;
    for (int p = 0; p < nprob; ++p) if (Bx[p] > mx) mx = Bx[p];
    return mx > 64 ? 8 : mx > 32 ? 4 : mx > 16 ? 2 : 1;
}

static CostPlan plan_direct(int nprob, const int* Bx, const int* By, const int* same, int64_t K) {
    CostPlan pl{};
    pl.use_mfma = false;
    pl.tile = DT;
    const int trows = 16 * direct_ra(nprob, Bx);
    int64_t ntiles = 0;
    for (int p = 0; p < nprob; ++p) {
        const int ti = (Bx[p] + trows - 1) / trows, tj = (By[p] + DT - 1) / DT;
        ntiles += same[p] ? ((int64_t)ti * tj + ti) / 2 + 1 : (int64_t)ti * tj;     // about half the tiles are skipped
    }
    const int64_t ksteps = (K + DKT - 1) / DKT;
    // ~4 workgroups per CU (18 KB LDS, <64 VGPRs each) so that staging latency is covered
    int64_t nchunk = (1024 + ntiles - 1) / ntiles;
    if (nchunk > ksteps) nchunk = ksteps;
    if (nchunk < 1) nchunk = 1;
    const int64_t steps_per_chunk = (ksteps + nchunk - 1) / nchunk;
    pl.chunk = steps_per_chunk * DKT;
    pl.nchunk = (int)((K + pl.chunk - 1) / pl.chunk);
    size_t bytes = 0;
    for (int p = 0; p < nprob; ++p) {
        pl.partial_off[p] = bytes;
        bytes += align_up((size_t)pl.nchunk * Bx[p] * By[p] * sizeof(float), 256);
    }
    pl.ws_bytes = bytes;
    return pl;
}

static int run_direct(CostBatch& cb, int64_t K, float sc, int T, int J, void* ws, size_t ws_bytes,
                      bool partial_only, hipStream_t st) {
    int Bx[3], By[3], same[3];
    for (int p = 0; p < cb.nprob; ++p) { Bx[p] = cb.p[p].Bx; By[p] = cb.p[p].By; same[p] = cb.p[p].same; }
    CostPlan pl = plan_direct(cb.nprob, Bx, By, same, K);
    if (ws_bytes < pl.ws_bytes || !ws)
        return fail(KCCOT_EWORKSPACE, "pairwise_cost: workspace %zu < required %zu", ws_bytes, pl.ws_bytes);
    const int ra = direct_ra(cb.nprob, Bx), trows = 16 * ra;
    int total_tiles = 0, max_bx = 0, max_by = 0;
    for (int p = 0; p < cb.nprob; ++p) {
        CostProb& pr = cb.p[p];
        pr.tile = DT;
        pr.tiles_i = (pr.Bx + trows - 1) / trows;
        pr.tiles_j = (pr.By + DT - 1) / DT;
        pr.partial = reinterpret_cast<float*>(static_cast<char*>(ws) + pl.partial_off[p]);
        total_tiles += pr.tiles_i * pr.tiles_j;
        if (pr.Bx > max_bx) max_bx = pr.Bx;
        if (pr.By > max_by) max_by = pr.By;
    }
    if (total_tiles > 65535) return fail(KCCOT_EUNSUPPORTED, "pairwise_cost: %d tiles", total_tiles);
    bool vec = (K % 4 == 0);
    for (int p = 0; p < cb.nprob; ++p)
        vec = vec && (((uintptr_t)cb.p[p].x | (uintptr_t)cb.p[p].y) % 16 == 0);
    dim3 grid(pl.nchunk, total_tiles);
    if (ra == 8) hipLaunchKernelGGL(cost_direct_partial<8>, grid, dim3(256), 0, st, cb, K, pl.chunk, vec ? 1 : 0);
    else if (ra == 4) hipLaunchKernelGGL(cost_direct_partial<4>, grid, dim3(256), 0, st, cb, K, pl.chunk, vec ? 1 : 0);
    else if (ra == 2) hipLaunchKernelGGL(cost_direct_partial<2>, grid, dim3(256), 0, st, cb, K, pl.chunk, vec ? 1 : 0);
    else hipLaunchKernelGGL(cost_direct_partial<1>, grid, dim3(256), 0, st, cb, K, pl.chunk, vec ? 1 : 0);
    int rc = launch_status("cost_direct_partial");
    if (rc || partial_only) return rc;
    dim3 fgrid((max_by + CAUSAL_TILE - 1) / CAUSAL_TILE, (max_bx + CAUSAL_TILE - 1) / CAUSAL_TILE, cb.nprob);
    int fchunks = pl.nchunk;
    if (pl.nchunk > 48) {
        hipLaunchKernelGGL(cost_reduce_chunks, dim3((unsigned)(((int64_t)max_bx * max_by + 15) / 16), cb.nprob), dim3(256), 0, st,
                           cb, pl.nchunk);
        if ((rc = launch_status("cost_reduce_chunks"))) return rc;
        fchunks = 1;
    }
    hipLaunchKernelGGL(cost_finalize, fgrid, dim3(256), 0, st, cb, fchunks, sc, T, J);
    return launch_status("cost_finalize");
}

static int run_cost(CostBatch& cb, bool loss3, int64_t K, float sc, int T, int J, unsigned flags,
                    void* ws, size_t ws_bytes, hipStream_t st) {
    if ((flags & KCCOT_COST_FORCE_DIRECT) && (flags & KCCOT_COST_FORCE_MFMA))
        return fail(KCCOT_EINVAL, "pairwise_cost: FORCE_DIRECT and FORCE_MFMA are exclusive");
    bool mfma;
    if (flags & KCCOT_COST_FORCE_MFMA) {
        if (!gram_eligible(cb, K, loss3))
            return fail(KCCOT_EUNSUPPORTED, "pairwise_cost: the MFMA path needs 16-byte aligned rows "
                        "(K %% 4 == 0) and at most 64 rows per operand (128 for x==y)");
        mfma = true;
    } else if (flags & KCCOT_COST_FORCE_DIRECT) {
        mfma = false;
    } else {
        mfma = gram_preferred(cb, K, loss3);
    }
    const bool partial_only = (flags & KCCOT_COST_PARTIAL_ONLY) != 0;
    // split at the fp64 Gram sums for the contraction-sharded caller (include/kccot.h)
    const int stage = (flags & KCCOT_COST_GRAM_SUMS_ONLY) ? 1 : ((flags & KCCOT_COST_FROM_GRAM_SUMS) ? 2 : 0);
    if (stage && (partial_only || (flags & KCCOT_COST_FORCE_DIRECT) || !loss3))
        return fail(KCCOT_EINVAL, "pairwise_cost: the Gram-sum split applies to kccot_pairwise_cost3_f32 on the Gram paths only");
    if (mfma) return run_gram(cb, loss3, K, sc, T, J, ws, ws_bytes, partial_only, st, stage);
    if (!(flags & KCCOT_COST_FORCE_DIRECT) && !partial_only && gram_q256_eligible(cb, K, loss3))
        return run_gram_q256(cb, K, sc, T, J, ws, ws_bytes, st, stage);
    if (!(flags & KCCOT_COST_FORCE_DIRECT) && !partial_only && gram_tiled_eligible(cb, K, loss3))
        return run_gram_tiled(cb, K, sc, T, J, ws, ws_bytes, st, stage);
    if (stage) return fail(KCCOT_EUNSUPPORTED, "pairwise_cost3: no Gram path for B=%d K=%lld (Gram-sum split)", cb.p[0].Bx, (long long)K);
    if (!(flags & KCCOT_COST_FORCE_DIRECT) && !partial_only && gram_blocked_eligible(cb, K, loss3))
        return run_gram_blocked(cb, K, sc, T, J, ws, ws_bytes, st);
    return run_direct(cb, K, sc, T, J, ws, ws_bytes, partial_only, st);
}

}  // namespace kccot

using namespace kccot;

extern "C" size_t kccot_pairwise_cost_workspace_bytes(int Bx, int By, int64_t K) {
    if (Bx <= 0 || By <= 0 || K <= 0) return 0;
    // sized for the larger of the two paths so that the path may be chosen at call time
    int same = 0;
    size_t a = plan_direct(1, &Bx, &By, &same, K).ws_bytes;
    size_t b = plan_gram(K).ws_bytes;
    return a > b ? a : b;
}

extern "C" int kccot_pairwise_cost_f32(const float* x, const float* y, int Bx, int By, int64_t K, float sc,
                                       const float* h1, const float* M1, const float* h2,
                                       const float* M2, int T, int J, unsigned flags, float* C_out,
                                       void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!x || !y || !C_out) return fail(KCCOT_EINVAL, "pairwise_cost: null pointer");
    if (Bx <= 0 || By <= 0 || K <= 0)
        return fail(KCCOT_EINVAL, "pairwise_cost: bad shape Bx=%d By=%d K=%lld", Bx, By, (long long)K);
    if ((h1 == nullptr) != (M1 == nullptr) || (h2 == nullptr) != (M2 == nullptr))
        return fail(KCCOT_EINVAL, "pairwise_cost: h and M must be given together");
    if ((h1 || h2) && (T < 1 || J < 1)) return fail(KCCOT_EINVAL, "pairwise_cost: bad T=%d J=%d", T, J);
    if ((flags & KCCOT_COST_SAME) && (x != y || Bx != By))
        return fail(KCCOT_EINVAL, "pairwise_cost: KCCOT_COST_SAME needs x == y and Bx == By");
    CostBatch cb{};
    cb.nprob = 1;
    cb.p[0] = CostProb{x, y, Bx, By, (flags & KCCOT_COST_SAME) ? 1 : 0, h1, M1, h2, M2, C_out, nullptr, 0, 0, 0};
    return run_cost(cb, false, K, sc, T, J, flags, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t kccot_pairwise_cost3_workspace_bytes(int B, int64_t K) {
    if (B <= 0 || K <= 0) return 0;
    int Bs[3] = {B, B, B}, same[3] = {0, 1, 1};
    size_t a = plan_direct(3, Bs, Bs, same, K).ws_bytes;
    size_t b = plan_gram(K).ws_bytes;
    size_t c = gram_tiled_workspace_bytes(B, K);
    size_t d = gram_q256_workspace_bytes(B, K);
    a = a > b ? a : b;
    c = c > d ? c : d;
    return a > c ? a : c;
}

extern "C" int kccot_pairwise_cost3_gram_sums_span(int B, int64_t K, size_t* byte_offset, size_t* n_doubles) {
    if (!byte_offset || !n_doubles) return fail(KCCOT_EINVAL, "gram_sums_span: null pointer");
    *byte_offset = 0; *n_doubles = 0;
    if (B <= 0 || K <= 0 || K % 4 != 0 || K < 256) return 0;
    const bool x3 = !opt(OPT_GRAM_F32);
    if (B <= 64) gram_sums_span(B, K, byte_offset, n_doubles);
    else if (gram_q256_applies(B, K)) gram_q256_sums_span(B, K, byte_offset, n_doubles);
    else if (x3 && B % 128 == 0 && B <= 4096 && opt(OPT_COST_TILED)) gram_tiled_sums_span(B, K, byte_offset, n_doubles);
    return 0;
}

extern "C" int kccot_pairwise_cost3_f32(const float* real, const float* fake, int B, int64_t K, float sc,
                                        const float* h_fake, const float* h_real, const float* m_real,
                                        const float* m_fake, int T, int J, unsigned flags, float* C3,
                                        void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!real || !fake || !C3) return fail(KCCOT_EINVAL, "pairwise_cost3: null pointer");
    const int nfeat = (h_fake != nullptr) + (h_real != nullptr) + (m_real != nullptr) + (m_fake != nullptr);
    if (nfeat != 0 && nfeat != 4)   // all four (the loss) or none (plain squared distances, e.g. for the RBF kernel)
        return fail(KCCOT_EINVAL, "pairwise_cost3: give all of h_fake, h_real, m_real, m_fake or none");
    if (nfeat == 0) { T = 1; J = 1; }
    if (B <= 0 || K <= 0 || T < 1 || J < 1)
        return fail(KCCOT_EINVAL, "pairwise_cost3: bad shape B=%d K=%lld T=%d J=%d", B, (long long)K, T, J);
    const int64_t bb = (int64_t)B * B;
    CostBatch cb{};
    cb.nprob = 3;
    // gan_utils.py:221-223: xy=(real,fake,h_fake,m_real) xx=(real,real,h_real,m_real) yy=(fake,fake,h_fake,m_fake)
    cb.p[0] = CostProb{real, fake, B, B, 0, h_fake, m_real, nullptr, nullptr, C3, nullptr, 0, 0, 0};
    cb.p[1] = CostProb{real, real, B, B, 1, h_real, m_real, nullptr, nullptr, C3 + bb, nullptr, 0, 0, 0};
    cb.p[2] = CostProb{fake, fake, B, B, 1, h_fake, m_fake, nullptr, nullptr, C3 + 2 * bb, nullptr, 0, 0, 0};
    return run_cost(cb, true, K, sc, T, J, flags, ws, ws_bytes, (hipStream_t)stream);
}

// Row blocks [row_count, B] of the three cost matrices of compute_sinkhorn_loss for the batch-sharded caller
// (rank g owns samples [row_begin, row_begin + row_count) of the gathered batch): ONE launch of the exact
// direct-difference kernel over the three problems instead of three calls.  real / fake are the gathered [B,K]
// tensors, the features the gathered [B,T,J] ones; C3_rows is [3, row_count, B].
extern "C" size_t kccot_pairwise_cost3_rows_workspace_bytes(int row_count, int B, int64_t K) {
    if (row_count <= 0 || B <= 0 || K <= 0) return 0;
    int Bx[3] = {row_count, row_count, row_count}, By[3] = {B, B, B}, same[3] = {0, 0, 0};
    return plan_direct(3, Bx, By, same, K).ws_bytes;
}

extern "C" int kccot_pairwise_cost3_rows_f32(const float* real, const float* fake, int B, int64_t K, float sc,
                                             const float* h_fake, const float* h_real, const float* m_real,
                                             const float* m_fake, int T, int J, int row_begin, int row_count,
                                             float* C3_rows, void* ws, size_t ws_bytes, kccot_stream_t stream) {
    if (!real || !fake || !C3_rows) return fail(KCCOT_EINVAL, "pairwise_cost3_rows: null pointer");
    if (!h_fake || !h_real || !m_real || !m_fake) return fail(KCCOT_EINVAL, "pairwise_cost3_rows: all four feature tensors are needed");
    if (B <= 0 || K <= 0 || T < 1 || J < 1 || row_begin < 0 || row_count <= 0 || row_begin + row_count > B)
        return fail(KCCOT_EINVAL, "pairwise_cost3_rows: bad shape B=%d K=%lld rows [%d,%d)", B, (long long)K, row_begin,
                    row_begin + row_count);
    const int64_t rb = (int64_t)row_count * B, tj = (int64_t)T * J;
    const float* xr = real + (int64_t)row_begin * K;
    const float* yr = fake + (int64_t)row_begin * K;
    const float* hf = h_fake + row_begin * tj;
    const float* hr = h_real + row_begin * tj;
    CostBatch cb{};
    cb.nprob = 3;
    // gan_utils.py:221-223, rows of this rank only (no x == y shortcut: a row block is not symmetric)
    cb.p[0] = CostProb{xr, fake, row_count, B, 0, hf, m_real, nullptr, nullptr, C3_rows, nullptr, 0, 0, 0};
    cb.p[1] = CostProb{xr, real, row_count, B, 0, hr, m_real, nullptr, nullptr, C3_rows + rb, nullptr, 0, 0, 0};
    cb.p[2] = CostProb{yr, fake, row_count, B, 0, hf, m_fake, nullptr, nullptr, C3_rows + 2 * rb, nullptr, 0, 0, 0};
    return run_cost(cb, false, K, sc, T, J, KCCOT_COST_FORCE_DIRECT, ws, ws_bytes, (hipStream_t)stream);
}
