// The loss's three cost matrices for batches of 256, 512, 768, ... (B % 256 == 0): ONE Gram matrix of the 2B-row stack
// S = [real ; E], E = fake - real (the pair-difference form of cost_mfma.hip, gan_utils.py:221-223), produced in
// 256 x 256 tiles on the bf16 matrix pipe with the exact three-way split (round 3; the 128 x 128 tiles of cost_tiled.hip
// pulled 3.3x / 9.5x the algorithmic bytes through L2 at B = 256 / 512 and kept the matrix pipe 27 % busy).
//
// A workgroup (4 waves, ONE per SIMD, up to 512 registers each) owns a pair (pa <= pb) of 256-row panels of S over one
// K-chunk and computes the full 256 x 256 cross block S_pa S_pb^T: twice the flops per staged byte of a 128 x 128 tile.
// Every wave stages AND consumes: there are no producer waves to share a SIMD with, the split's VALU work is issued in
// the shadow of the wave's own MFMAs.
//   * loads: 16 float4 per thread and 32-k granule (128-byte row pieces), issued two 16-k steps before they are split;
//   * split: element pair (k, k+1) of a float4 goes to the even step of the granule, (k+2, k+3) to the odd one, so one
//     granule of loads feeds two LDS stages of 16 k (3 planes x 512 rows x 32 bytes = 48 KB per stage, two stages
//     resident) and every ds_write is a full dword.  Which 16 columns make up a step is irrelevant to a contraction as
//     long as both panels use the same map;
//   * consumers: wave (wr, wc) owns rows 128 wr of panel A x columns 128 wc of panel B = 4 x 4 MFMA tiles of 32 x 32
//     (256 accumulator registers), two B tiles' fragments held at a time, A fragments streamed per row tile.
//   * accumulation: fp32 MFMA accumulation over one K-chunk of at most 1536 columns (576 accumulations, see plan_q256),
//     fp64 across the chunks.
// E is never formed by a separate pass: the pairs (X_i, E_i) run FIRST, as their own launch, with E_i = F_i - X_i formed
// in registers (the thread that holds row r of X_i holds row r of F_i) and written out as a by-product; all other
// pairs then read X and E rows only (no subtrahend loads, no second row stream per E panel).
//
//   gram_q256<EPAIR>  pairs (X_i, E_i), writes E             partial tiles [pair][chunk][256][256] fp32
//   gram_q256<DIAG>   pairs (p, p): one panel staged
//   gram_q256<OFF>    every other pair (pa < pb)
//   gram_q256_reduce  fp64 sum over the chunks (fixed order)
//   gram_q256_finalize  distances from the Gram entries in fp64 (the formulas of gram_finalize), scale, causal term
#include "common.h"
#include "cost_internal.h"
#include "options.h"
#include "gram_q.h"

namespace kccot {

constexpr int QP = 256;                    // rows of a panel
constexpr int QPLANE = 2 * QP * QROWB;     // 16384 bytes: A panel rows 0..255, B panel rows 256..511
constexpr int QSLOT = 3 * QPLANE;          // 49152 bytes per step
constexpr int QELEMS = QP * QP;

struct Q256Args {
    const float* x;       // real [B,K]
    const float* f;       // fake [B,K]
    float* e;             // E = fake - real [B,K] (workspace): written by the <true> launch, read by the <false> one
    int B, nx, nt, nchunk;
    int64_t K, chunk;
    float* part;          // [nt (nt + 1) / 2][nchunk][QELEMS]
};

__host__ __device__ inline int q256_pair_slot(int nt, int pa, int pb) { return pa * nt - (pa * (pa - 1)) / 2 + (pb - pa); }

// SAME (a diagonal pair, pa == pb: only panel A is staged) and RAGGED (K % 32 != 0: the last granule of the last chunk is
// partial, values past K are zeroed) are template parameters so that the staging code of each form is straight-line and the
// compiler interleaves ALL of it with the MFMAs of the step.
// HALF (B == 128, round 3): the whole stack [X ; E] is ONE 256-row panel -- a diagonal pair whose rows 128.. are formed in
// registers from the rows of `fake` the same thread loads (the staging of EPAIR without its E store: nothing else reads E).
template <bool EPAIR, bool SAME, bool RAGGED, bool HALF = false>
__device__ __forceinline__ void q256_body(const Q256Args& a, int pa, int pb, int chunk_id, unsigned char* zs0, unsigned char* zs1) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int NP = SAME ? 8 : 16;
    // granules in flight: a diagonal pair stages half as much and has the registers for two (its steps are short: 60
    // MFMAs against 96, so one step of load latency cover is not enough)
    constexpr int DEPTH = SAME ? 2 : 1;
    const int64_t K = a.K;
    const int64_t kbeg = (int64_t)chunk_id * a.chunk;
    const int64_t kend = (kbeg + a.chunk < K) ? kbeg + a.chunk : K;
    const int ng = (int)((kend - kbeg + QG - 1) / QG);

    // ---- staging role: thread = (row rr + 32 p, columns 4 q .. 4 q + 3 of the granule), p < 8 panel A, p >= 8 panel B
    const int q = t & 7, rr = t >> 3;
    // Loads (and the E stores) go through buffer descriptors: panel base in SGPRs, the lane's row / column part in ONE
    // 32-bit voffset, the pass's 32-row step as a scalar offset -- sixteen 64-bit flat addresses cost 32 registers.
    const float* apanel;
    const float* bpanel;
    static_assert(!HALF || (SAME && !EPAIR), "the single-panel form is a diagonal pair");
    if (HALF) {
        apanel = a.x;
        bpanel = a.f;
    } else if (EPAIR) {
        apanel = a.x + (int64_t)pa * QP * K;
        bpanel = a.f + (int64_t)pa * QP * K;
    } else {
        apanel = pa < a.nx ? a.x + (int64_t)pa * QP * K : a.e + (int64_t)(pa - a.nx) * QP * K;
        bpanel = pb < a.nx ? a.x + (int64_t)pb * QP * K : a.e + (int64_t)(pb - a.nx) * QP * K;
    }
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(apanel), 0, 0xFFFFFFFFu, 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bpanel), 0, 0xFFFFFFFFu, 0x00020000);
    const auto re = __builtin_amdgcn_make_buffer_rsrc(a.e + (int64_t)pa * QP * K, 0, 0xFFFFFFFFu, 0x00020000);   // EPAIR only
    const unsigned rstep = (unsigned)(32 * K * 4);                           // bytes; 7 * rstep + a row < 4 GiB: K <= 2^22 (host)
    const unsigned lrow = (unsigned)((int64_t)rr * K * 4);
    const int64_t kmax = K - 4;                                              // K % 4 == 0, K >= 256 (host checks)
    const int woff = rr * QROWB + 4 * q;                                     // + 32 p rows

    // The loads are UNCONDITIONAL (column clamped into the row): granules past the end of the chunk are fetched, split and
    // staged like any other -- into an LDS stage no step reads -- so the granule loop has one straight-line body.
    float4 G0[NP], G1[DEPTH == 2 ? NP : 1];
    float2 carry[NP];
    auto issue = [&](float4 (&G)[NP], int g) {
        int64_t k = kbeg + (int64_t)g * QG + 4 * q;
        k = k < kmax ? k : kmax;
        const unsigned vo = lrow + (unsigned)(k * 4);
#pragma unroll
        for (int p = 0; p < (HALF ? 4 : 8); ++p)
            G[p] = __builtin_bit_cast(float4, (qu32x4)__builtin_amdgcn_raw_buffer_load_b128(ra, (int)vo, (int)(p * rstep), 0));
        if constexpr (HALF) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                G[4 + p] = __builtin_bit_cast(float4, (qu32x4)__builtin_amdgcn_raw_buffer_load_b128(rb, (int)vo, (int)(p * rstep), 0));
        }
        if constexpr (!SAME) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                G[8 + p] = __builtin_bit_cast(float4, (qu32x4)__builtin_amdgcn_raw_buffer_load_b128(rb, (int)vo, (int)(p * rstep), 0));
        }
    };
    // granule g has landed: E rows formed (EPAIR), the even step's pairs split into `zs`, the odd step's kept in `carry`
    auto emit_even = [&](float4 (&G)[NP], int g, unsigned char* zs) {
        const int64_t k = kbeg + (int64_t)g * QG + 4 * q;
        const bool kok = k + 4 <= kend;                                      // false past the end of the chunk
        const unsigned vo = lrow + (unsigned)((k < kmax ? k : kmax) * 4);    // where issue() read these columns
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float4 v = G[p];
            if (HALF && p >= 4) { v.x -= G[p - 4].x; v.y -= G[p - 4].y; v.z -= G[p - 4].z; v.w -= G[p - 4].w; }
            if (EPAIR && p >= 8) {
                v.x -= G[p - 8].x; v.y -= G[p - 8].y; v.z -= G[p - 8].z; v.w -= G[p - 8].w;
                // unconditional: a granule past the end of the chunk re-writes the E values of columns another workgroup
                // owns -- the same bits from the same inputs -- rather than put a branch into the step's block
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(qu32x4, v), re, (int)vo, (int)((p - 8) * rstep), 0);
            }
            if (RAGGED) { v.x = kok ? v.x : 0.f; v.y = kok ? v.y : 0.f; v.z = kok ? v.z : 0.f; v.w = kok ? v.w : 0.f; }
            gq_split_store<QPLANE>(zs, woff + 32 * p * QROWB, v.x, v.y);
            carry[p] = make_float2(v.z, v.w);
        }
    };
    auto emit_odd = [&](unsigned char* zs) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            gq_split_store<QPLANE>(zs, woff + 32 * p * QROWB, carry[p].x, carry[p].y);
        }
    };

    // ---- consuming role.
    // Off-diagonal pair: wave (wr, wc) owns rows 128 wr .. of panel A x columns 128 wc .. of panel B = 4 x 4 tiles.
    // Diagonal pair (SAME): S_p S_p^T is symmetric, only the 36 tiles (r <= c) of its 8 x 8 tile grid are needed.  Each wave
    // takes a 2 x 4 block of them plus two more (ten accumulator tiles, 60 MFMAs per step instead of 96):
    //     wave 0: rows {0,1} x cols {0..3} (its (1,0) is redundant) + (2,2), (2,3)      wave 2: rows {0,1} x cols {4..7} + (3,3)
    //     wave 1: rows {4,5} x cols {4..7} (its (5,4) is redundant) + (6,6), (6,7)      wave 3: rows {2,3} x cols {4..7} + (7,7)
    // (waves 2 and 3 run their extra tile twice instead of branching; the copy is not stored).  The tiles below the
    // diagonal are never written: gram_q256_reduce zeroes them, q256_gram reads (min, max).
    constexpr int NT = SAME ? 10 : 16;
    const int wr = wave >> 1, wc = wave & 1;
    const int lo = (lane & 31) * QROWB + 16 * (lane >> 5);                   // row (lane & 31), k half (lane >> 5) of the step
    const int tr0 = SAME ? (wave == 1 ? 4 : (wave == 3 ? 2 : 0)) : 4 * wr;   // first row tile of the block
    const int tc0 = SAME ? (wave == 0 ? 0 : 4) : 4 * wc;                     // first column tile of the block
    const int ter = SAME ? (wave == 0 ? 2 : (wave == 1 ? 6 : (wave == 2 ? 3 : 7))) : 0;   // SAME: the extra tiles (ter, ter), (ter, ter + 1)
    const int aoff = tr0 * 32 * QROWB + lo;
    const int boff = (SAME ? 0 : QP * QROWB) + tc0 * 32 * QROWB + lo;        // a diagonal pair reads its B fragments from the A rows
    const int eoff = ter * 32 * QROWB + lo;
    QAcc acc[NT];
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) QACC(acc[t2], r) = 0.f;
    auto step = [&](const unsigned char* zs) {
        if constexpr (SAME) {
            QFrag bf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = gq_frag<QPLANE>(zs, boff + j * 32 * QROWB);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const QFrag af = gq_frag<QPLANE>(zs, aoff + i * 32 * QROWB);
#pragma unroll
                for (int j = 0; j < 4; ++j) gq_mfma6(acc[4 * i + j], af, bf[j]);
            }
            const QFrag ea = gq_frag<QPLANE>(zs, eoff);                            // row tile `ter` = column tile `ter`: one fragment
            gq_mfma6(acc[8], ea, ea);
            const QFrag eb = gq_frag<QPLANE>(zs, eoff + (wave < 2 ? 32 * QROWB : 0));
            gq_mfma6(acc[9], ea, eb);
        } else {
            // two B tiles at a time (24 fragment registers instead of 48; the A fragments are read twice per step)
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                const QFrag b0 = gq_frag<QPLANE>(zs, boff + (2 * jh) * 32 * QROWB), b1 = gq_frag<QPLANE>(zs, boff + (2 * jh + 1) * 32 * QROWB);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const QFrag af = gq_frag<QPLANE>(zs, aoff + i * 32 * QROWB);
                    gq_mfma6(acc[4 * i + 2 * jh], af, b0);
                    gq_mfma6(acc[4 * i + 2 * jh + 1], af, b1);
                }
            }
        }
    };

    // Interleave request for one step's block (LLVM sched_group_barrier): the fragments of the first tiles, then per MFMA
    // at most one LDS read, three VALU instructions of the split, one LDS write and one buffer load.  One wave per SIMD:
    // whatever is not issued inside an MFMA's 24 free issue cycles is exposed, and hipcc's own order left runs of ten
    // VALU instructions between two MFMAs next to runs of nine bare MFMAs.
    auto interleave = [&](bool with_loads) {
        __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
        for (int m = 0; m < 6 * NT; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            // the next granule's loads (and the E stores) spread over the whole step, one per 6 * NT / NP MFMAs: issued as a
            // burst they back up the CU's memory pipeline, and a wave stuck in a VMEM issue issues no MFMA either (one wave
            // per SIMD; measured: without the loads the same kernel keeps the matrix pipe 82 % busy instead of 59 %)
            if (with_loads && (m % (6 * NT / NP)) == (6 * NT / NP) - 1) __builtin_amdgcn_sched_group_barrier(0x030, EPAIR ? 2 : 1, 0);
        }
    };
    // One granule = two steps.  Its odd step splits the even half of the NEXT granule out of `G` (loads issued DEPTH granules
    // earlier) and re-issues `G` for the granule DEPTH further on.
    auto granule = [&](float4 (&G)[NP], int g) {
        step(zs0);
        emit_odd(zs1);
        interleave(false);
        __builtin_amdgcn_sched_barrier(0);                                   // (register-only MFMAs would drift across the asm barrier)
        lds_barrier();
        __builtin_amdgcn_sched_barrier(0);
        step(zs1);
        emit_even(G, g + 1, zs0);
        issue(G, g + 1 + DEPTH);
        interleave(true);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    issue(G0, 0);
    if constexpr (DEPTH == 2) issue(G1, 1);
    emit_even(G0, 0, zs0);
    issue(G0, DEPTH);
    lds_barrier();
    if constexpr (DEPTH == 2) {
        int g = 0;
        for (; g + 1 < ng; g += 2) {
            granule(G1, g);
            granule(G0, g + 1);
        }
        if (g < ng) granule(G1, g);
    } else {
        for (int g = 0; g < ng; ++g) granule(G0, g);
    }

    // accumulator register r of lane l is element ((r & 3) + 8 (r >> 2) + 4 (l >> 5), l & 31) of its 32 x 32 tile
    float* o = a.part + ((int64_t)q256_pair_slot(a.nt, pa, pb) * a.nchunk + chunk_id) * QELEMS + (4 * (lane >> 5)) * QP + (lane & 31);
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2) {
        int trow, tcol;
        if (SAME && t2 >= 8) { trow = ter; tcol = ter + (t2 - 8); }
        else { trow = tr0 + t2 / 4; tcol = tc0 + t2 % 4; }
        if (SAME && t2 == 9 && wave >= 2) continue;                          // the duplicate of waves 2 and 3
        float* ot = o + (32 * trow) * QP + 32 * tcol;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[((r & 3) + 8 * (r >> 2)) * QP] = QACC(acc[t2], r);
    }
}

// MODE 0: the pairs (X_i, E_i), which also write E (launched first); 1: the diagonal pairs (p, p); 2: every other pair.
// One kernel per mode: each is one straight-line instantiation of the body (two in one kernel spilled registers).
enum { Q256_EPAIR = 0, Q256_DIAG = 1, Q256_OFF = 2, Q256_HALF = 3 };
__host__ __device__ inline int q256_mode_pairs(int mode, int nx) {
    if (nx == 0) return mode == Q256_HALF ? 1 : 0;                           // B == 128: the stack is one panel
    if (mode == Q256_HALF) return 0;
    const int nt = 2 * nx;
    return mode == Q256_EPAIR ? nx : (mode == Q256_DIAG ? nt : nt * (nt - 1) / 2 - nx);
}

template <int MODE, bool RAGGED>
__global__ __launch_bounds__(256) void gram_q256(Q256Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char zs0[QSLOT];
    __shared__ __attribute__((aligned(16))) unsigned char zs1[QSLOT];
    // XCD-aware block -> (pair, chunk) map: workgroups are dealt to the 8 XCDs round-robin, so XCD x = id % 8 takes the
    // chunks x, x + 8, ... and, within a chunk, the pairs in order: the workgroups an XCD runs at a time are all pairs of
    // a few K-ranges, so a panel chunk is fetched from HBM once per XCD and served from its L2 to the other pairs.
    const int np = q256_mode_pairs(MODE, a.nx);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chunk_id = (slot / np) * 8 + xcd;
    int pa, pb;
    if (MODE == Q256_EPAIR) {
        pa = slot % np; pb = a.nx + pa;
    } else if (MODE == Q256_HALF) {
        pa = pb = 0;
    } else if (MODE == Q256_DIAG) {
        pa = pb = slot % np;
    } else {
        int r = slot % np;
        pa = 0;
        for (;;) {                                                            // row-major over pa < pb without the (i, nx + i)
            const int len = a.nt - pa - 1 - (pa < a.nx ? 1 : 0);
            if (r < len) break;
            r -= len; ++pa;
        }
        pb = pa + 1 + r;
        if (pa < a.nx && pb >= a.nx + pa) ++pb;
    }
    if ((int64_t)chunk_id * a.chunk >= a.K) return;                          // an empty trailing chunk
    // (the integer division above runs on the VALU: make the uniformity of what the buffer descriptors are built from explicit)
    q256_body<MODE == Q256_EPAIR, MODE == Q256_DIAG || MODE == Q256_HALF, RAGGED, MODE == Q256_HALF>(a, __builtin_amdgcn_readfirstlane(pa), __builtin_amdgcn_readfirstlane(pb),
                                                             __builtin_amdgcn_readfirstlane(chunk_id), zs0, zs1);
}

// fp64 sum over the chunks, fixed order; eight loads in flight
__global__ __launch_bounds__(256) void gram_q256_reduce(const float* __restrict__ part, int nstride, int nchunk, int nt,
                                                        double* __restrict__ gsum) {
    const int qs = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    {   // the tiles below the diagonal of a diagonal pair are never computed (gram_q256<DIAG>): their sums are defined as 0
        int pa = 0, r = qs;
        while (r >= nt - pa) { r -= nt - pa; ++pa; }
        if (r == 0 && ((e >> 8) >> 5) > ((e & 255) >> 5)) { gsum[(int64_t)qs * QELEMS + e] = 0.0; return; }
    }
    const float* p = part + (int64_t)qs * nstride * QELEMS + e;
    double s = 0.0;
    int c = 0;
    for (; c + 8 <= nchunk; c += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[(int64_t)(c + i) * QELEMS];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (double)v[i];
    }
    for (; c < nchunk; ++c) s += (double)p[(int64_t)c * QELEMS];
    gsum[(int64_t)qs * QELEMS + e] = s;
}

struct Q256Fin {
    const double* gsum;
    int B, nt;
    float* out[3];
    const float* h[3];
    const float* M[3];
    float sc;
    int T, J;
};

// Gram entry of stack rows (a, b)
__device__ __forceinline__ double q256_gram(const double* __restrict__ gs, int nt, int a, int b) {
    const bool sw = a > b;                                                   // panel order, and row <= column inside a diagonal pair
    const int a2 = sw ? b : a, b2 = sw ? a : b;
    return gs[(int64_t)q256_pair_slot(nt, a2 >> 8, b2 >> 8) * QELEMS + (a2 & 255) * QP + (b2 & 255)];
}

// One 16 x 16 output tile per block; blockIdx.z = problem (xy, xx, yy).  Formulas of gram_finalize (cost_mfma.hip).
__global__ __launch_bounds__(256) void gram_q256_finalize(Q256Fin f) {
    __shared__ __attribute__((aligned(16))) float sh[CAUSAL_TILE * CAUSAL_PITCH];
    __shared__ __attribute__((aligned(16))) float sm[CAUSAL_TILE * CAUSAL_PITCH];
    const int p = blockIdx.z, B = f.B, nt = f.nt;
    const int i0 = blockIdx.y * CAUSAL_TILE, j0 = blockIdx.x * CAUSAL_TILE;
    const int i = i0 + (threadIdx.x >> 4), j = j0 + (threadIdx.x & 15);      // B % 16 == 0: always in range
    const double* G = f.gsum;
    const double g_ii = q256_gram(G, nt, i, i), g_jj = q256_gram(G, nt, j, j), g_ij = q256_gram(G, nt, i, j);
    const double e_ii = q256_gram(G, nt, B + i, B + i), e_jj = q256_gram(G, nt, B + j, B + j), e_ij = q256_gram(G, nt, B + i, B + j);
    const double x_ii = q256_gram(G, nt, i, B + i), x_jj = q256_gram(G, nt, j, B + j);
    const double x_ij = q256_gram(G, nt, i, B + j), x_ji = q256_gram(G, nt, j, B + i);
    const bool diag = i == j;
    const double dxx = diag ? 0.0 : g_ii + g_jj - 2.0 * g_ij;
    const double dxy = dxx + e_jj - 2.0 * (x_ij - x_jj);
    const double dee = e_ii + e_jj - 2.0 * e_ij;
    const double dyy = diag ? 0.0 : dxx + dee + 2.0 * (x_ii - x_ij - x_ji + x_jj);
    double D = (p == 1) ? dxx : (p == 0 ? dxy : dyy);
    if (D < 0.0) D = 0.0;   // a squared distance; rounding of the Gram terms may leave -tiny
    float c = (float)D * f.sc;
    if (f.h[p]) c += causal_tile16(f.h[p], f.M[p], i0, j0, B, B, f.T, f.J, sh, sm) * f.sc;
    f.out[p][(int64_t)i * B + j] = c;
}

// ---- host side ---------------------------------------------------------------------------------------
struct Q256Plan { int nx, nt, npairs, nchunk; int64_t chunk; size_t part_bytes, gsum_bytes, e_bytes, ws_bytes; };

static Q256Plan plan_q256(int B, int64_t K) {
    Q256Plan pl{};
    pl.nx = B / QP;                                                          // 0 for B == 128: one panel [X ; E]
    pl.nt = pl.nx ? 2 * pl.nx : 1;
    pl.npairs = pl.nt * (pl.nt + 1) / 2;
    const int64_t ngran = (K + QG - 1) / QG;
    // Number of K-chunks (a multiple of 8: one residue class per XCD); the three launches run pairs(mode) * n workgroups
    // each, one per CU, i.e. ceil(. / 256) rounds of (granules per chunk + the fixed cost of a workgroup) each.
    // The chunk length is bounded by the ACCUMULATION, not by occupancy: an MFMA accumulation loses ~0.02 ulp of the running
    // sum (addends are truncated when aligned to it; measured: 2.3e-5 low after 14 000 accumulations of all-positive terms
    // in the 128-tile kernel, 1.5e-5 of a distance here with 2300), so one partial tile holds at most Q_MAX_GRAN granules
    // = 576 accumulations (a distance within 4e-6) and the fp64 reduction adds the tiles up.  No second accumulator level
    // as in the 128-tile kernel: there are no registers for one, and folding runs into the workgroup's own tile in memory
    // (accumulators read by VALU code, or a run loop around the pipeline) made hipcc spill the accumulator file.
    const int64_t nmin = ((ngran + Q_MAX_GRAN - 1) / Q_MAX_GRAN + 7) / 8 * 8;
    int best = (int)nmin;
    double best_cost = 1e300;
    for (int64_t n = nmin; n <= 2 * nmin; n += 8) {
        const int64_t gpc = (ngran + n - 1) / n;
        int64_t rounds = 0;
        for (int mode = 0; mode < 4; ++mode) rounds += ((int64_t)q256_mode_pairs(mode, pl.nx) * n + 255) / 256;
        const double cost = (double)rounds * ((double)gpc + 3.0);           // + the fixed cost of a workgroup (its 256 KB tile, ramp-up)
        if (cost < best_cost - 1e-9) { best_cost = cost; best = (int)n; }
    }
    pl.nchunk = best;
    pl.chunk = ((ngran + best - 1) / best) * QG;
    pl.part_bytes = align_up((size_t)pl.npairs * pl.nchunk * QELEMS * sizeof(float), 256);
    pl.gsum_bytes = align_up((size_t)pl.npairs * QELEMS * sizeof(double), 256);
    pl.e_bytes = pl.nx ? align_up((size_t)B * K * sizeof(float), 256) : 0;    // the single-panel form never writes E
    pl.ws_bytes = pl.part_bytes + pl.gsum_bytes + pl.e_bytes;
    return pl;
}

static bool q256_shape_ok(int B, int64_t K) {
    return (B == QP / 2 || (B >= QP && B % QP == 0)) && B <= 4096 && K % 4 == 0 && K >= 256 && K <= (1 << 22);   // K: 32-bit panel offsets
}

void gram_q256_sums_span(int B, int64_t K, size_t* off, size_t* n) {
    const Q256Plan pl = plan_q256(B, K);
    *off = pl.part_bytes;
    *n = (size_t)pl.npairs * QELEMS;
}

size_t gram_q256_workspace_bytes(int B, int64_t K) { return q256_shape_ok(B, K) ? plan_q256(B, K).ws_bytes : 0; }

bool gram_q256_applies(int B, int64_t K) {
    return q256_shape_ok(B, K) && opt(OPT_COST_TILED) && opt(OPT_COST_TILE256) && !opt(OPT_GRAM_F32);
}

bool gram_q256_eligible(const CostBatch& cb, int64_t K, bool loss3) {
    if (!loss3 || cb.nprob != 3) return false;
    const int B = cb.p[0].Bx;
    if (cb.p[0].By != B || !gram_q256_applies(B, K)) return false;
    return ((uintptr_t)cb.p[0].x % 16 == 0) && ((uintptr_t)cb.p[0].y % 16 == 0);
}

int run_gram_q256(const CostBatch& cb, int64_t K, float sc, int T, int J, void* ws, size_t ws_bytes, hipStream_t st,
                  int stage) {   // stage 0: everything; 1: stop after the fp64 sums; 2: finalize only
    const int B = cb.p[0].Bx;
    const Q256Plan pl = plan_q256(B, K);
    if (!ws || ws_bytes < pl.ws_bytes)
        return fail(KCCOT_EWORKSPACE, "pairwise_cost3(tile256): workspace %zu < required %zu", ws_bytes, pl.ws_bytes);
    float* part = static_cast<float*>(ws);
    double* gsum = reinterpret_cast<double*>(static_cast<char*>(ws) + pl.part_bytes);
    float* e = reinterpret_cast<float*>(static_cast<char*>(ws) + pl.part_bytes + pl.gsum_bytes);
    int rc;
    if (stage != 2) {
        Q256Args qa{cb.p[0].x, cb.p[0].y, e, B, pl.nx, pl.nt, pl.nchunk, K, pl.chunk, part};
        const bool ragged = K % QG != 0;
        for (int mode = 0; mode < 4; ++mode) {
            const int np = q256_mode_pairs(mode, pl.nx);
            if (np == 0) continue;
            const dim3 grid(np * pl.nchunk), block(256);
            if (mode == Q256_EPAIR) {
                if (ragged) hipLaunchKernelGGL((gram_q256<Q256_EPAIR, true>), grid, block, 0, st, qa);
                else hipLaunchKernelGGL((gram_q256<Q256_EPAIR, false>), grid, block, 0, st, qa);
            } else if (mode == Q256_DIAG) {
                if (ragged) hipLaunchKernelGGL((gram_q256<Q256_DIAG, true>), grid, block, 0, st, qa);
                else hipLaunchKernelGGL((gram_q256<Q256_DIAG, false>), grid, block, 0, st, qa);
            } else if (mode == Q256_HALF) {
                if (ragged) hipLaunchKernelGGL((gram_q256<Q256_HALF, true>), grid, block, 0, st, qa);
                else hipLaunchKernelGGL((gram_q256<Q256_HALF, false>), grid, block, 0, st, qa);
            } else {
                if (ragged) hipLaunchKernelGGL((gram_q256<Q256_OFF, true>), grid, block, 0, st, qa);
                else hipLaunchKernelGGL((gram_q256<Q256_OFF, false>), grid, block, 0, st, qa);
            }
            if ((rc = launch_status("gram_q256"))) return rc;
        }
        const int nvalid = (int)((K + pl.chunk - 1) / pl.chunk);
        hipLaunchKernelGGL(gram_q256_reduce, dim3(QELEMS / 256, pl.npairs), dim3(256), 0, st, (const float*)part, pl.nchunk, nvalid, pl.nt, gsum);
        if ((rc = launch_status("gram_q256_reduce"))) return rc;
        if (stage == 1) return 0;
    }
    Q256Fin f{};
    f.gsum = gsum; f.B = B; f.nt = pl.nt; f.sc = sc; f.T = T; f.J = J;
    for (int p = 0; p < 3; ++p) { f.out[p] = cb.p[p].out; f.h[p] = cb.p[p].h1; f.M[p] = cb.p[p].M1; }
    hipLaunchKernelGGL(gram_q256_finalize, dim3(B / CAUSAL_TILE, B / CAUSAL_TILE, 3), dim3(256), 0, st, f);
    return launch_status("gram_q256_finalize");
}

}  // namespace kccot
