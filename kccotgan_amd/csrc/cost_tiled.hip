// The loss's three cost matrices for batches of 128, 384, 640, ... (B % 128 == 0 that cost_tile256.hip's 256-row tiles do
// not take: B % 256 != 0, or the option "cost_tile256" = 0): ONE Gram matrix of the 2B-row
// stack S = [real ; E], E = fake - real (the pair-difference form of cost_mfma.hip, gan_utils.py:221-223), produced
// in 128 x 128 tiles on the bf16 matrix pipe with the exact three-way split.
//
// The blocked path (cost_mfma.hip: run_gram_blocked) reuses the 128-row stacked-Gram kernel per pair of 64-row
// blocks: every launch re-reads 128 rows and, off the diagonal, only 4 of its 10 sub-tiles are new information.
// Here a workgroup owns one pair (pa <= pb) of 128-row panels of S over one K-chunk and computes the full
// 128 x 128 cross block  S_pa S_pb^T  -- every 32 x 32 MFMA tile is needed exactly once (diagonal pairs compute
// their lower triangle too: 1/(nt+1) of the work).  Grid = (pairs, chunks) with the pair index fastest, so that the
// workgroups resident at one time work on the same few K-chunks of ALL panels: a panel chunk fetched from HBM for
// one pair is served from L2 / Infinity Cache to the other pairs that need it.
//
//   gram_tile_x3      partial tiles [pair][chunk][128][128] fp32
//   gram_tile_reduce  fp64 sum over the chunks (fixed order)
//   gram_tile_finalize  distances from the Gram entries in fp64 (the formulas of gram_finalize with global
//                     indices), scale, causal term (causal_tile16)
#include "common.h"
#include "cost_internal.h"
#include "options.h"

namespace kccot {

typedef __bf16 tbf16x8 __attribute__((ext_vector_type(8)));
typedef float tf32x16 __attribute__((ext_vector_type(16)));

constexpr int TK = 32;                       // floats of K per stage (two LDS buffers of 3 planes x 256 rows: 120 KB)
constexpr int TPITCH = TK * 2 + 16;          // 80 bytes per row of one bf16 plane: 16-lane b128 groups hit 16 distinct 4-bank slots
constexpr int TP = 128;                      // rows of a panel
constexpr int TROWS = 2 * TP;                // LDS rows: A panel 0..127, B panel 128..255
constexpr int TPLANE = TROWS * TPITCH;       // 20480 bytes
constexpr int TBUF = 3 * TPLANE;             // 61440 bytes
constexpr int TELEMS = TP * TP;

struct TileArgs {
    const float* real;
    const float* fake;
    int B, nt, nchunk;
    int64_t K, chunk;
    float* part;      // [npairs][nchunk][TELEMS]
    const float* ediff;             // E = fake - real [B][K] formed ONCE (ediff_rows), or null: E panels subtract while staging
};

// E = fake - real once, in fp32 (the same subtraction the producers do while staging: bit-identical sums).  Why it pays
// for large batches: the tile kernel is bound by the bytes its workgroups pull through L2 (PMC: 3.3x the algorithmic
// bytes at B = 256, 9.5x at B = 512 -- pairs that share a panel drift apart and re-fetch it), and a panel of E costs TWO
// row streams (fake and real) every time it is staged: 96 panel streams per K-range at B = 512 against 64 with E
// materialised (-33 % of the streams; measured -19 % of the kernel), for one extra streaming pass (read 2, write 1 tensors).
__global__ __launch_bounds__(256) void ediff_rows(const float* __restrict__ real, const float* __restrict__ fake, int64_t n4,
                                                  float* __restrict__ e) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(fake)[i], b = reinterpret_cast<const float4*>(real)[i];
        reinterpret_cast<float4*>(e)[i] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    }
}

__device__ __forceinline__ float4 tld4(const float* __restrict__ row, int64_t k, int64_t kend) {
    // K % 4 == 0 and 16-byte aligned rows (checked on the host); a stage may run past the chunk end
    return (k + 4 <= kend) ? *reinterpret_cast<const float4*>(row + k) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ void tsplit3_store(unsigned char* zs, int byte_off, float4 v) {
    const unsigned x[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    unsigned m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float h = __uint_as_float(x[i] & 0xFFFF0000u);
        const float r1 = __uint_as_float(x[i]) - h;                        // exact
        const float mm = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
        m[i] = __float_as_uint(r1);
        l[i] = __float_as_uint(r1 - mm);                                   // exact, <= 8 significant bits
    }
    uint2 ph, pm, pl;   // dword = bf16(e0) | bf16(e1) << 16
    ph.x = __builtin_amdgcn_perm(x[1], x[0], 0x07060302u); ph.y = __builtin_amdgcn_perm(x[3], x[2], 0x07060302u);
    pm.x = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u); pm.y = __builtin_amdgcn_perm(m[3], m[2], 0x07060302u);
    pl.x = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u); pl.y = __builtin_amdgcn_perm(l[3], l[2], 0x07060302u);
    *reinterpret_cast<uint2*>(zs + byte_off) = ph;
    *reinterpret_cast<uint2*>(zs + TPLANE + byte_off) = pm;
    *reinterpret_cast<uint2*>(zs + 2 * TPLANE + byte_off) = pl;
}

struct TFrag { tbf16x8 h, m, l; };
__device__ __forceinline__ TFrag tld_frag(const unsigned char* zs, int off) {
    TFrag f;
    f.h = *reinterpret_cast<const tbf16x8*>(zs + off);
    f.m = *reinterpret_cast<const tbf16x8*>(zs + TPLANE + off);
    f.l = *reinterpret_cast<const tbf16x8*>(zs + 2 * TPLANE + off);
    return f;
}
__device__ __forceinline__ void tmfma_x3(tf32x16& acc, const TFrag& a, const TFrag& b) {
    // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
}

// pair index q -> (pa, pb), pa <= pb < nt, row-major over the upper triangle
__device__ __forceinline__ void tile_pair(int q, int nt, int& pa, int& pb) {
    pa = 0;
    while (q >= nt - pa) { q -= nt - pa; ++pa; }
    pb = pa + q;
}

// panel p of the stack [real ; fake - real]: rows main[r] - sub[r] (sub null for the real part)
__device__ __forceinline__ void panel_rows(const TileArgs& a, int p, const float*& main, const float*& sub) {
    const int r0 = p * TP;
    if (r0 < a.B) { main = a.real + (int64_t)r0 * a.K; sub = nullptr; }
    else if (a.ediff) { main = a.ediff + (int64_t)(r0 - a.B) * a.K; sub = nullptr; }
    else { main = a.fake + (int64_t)(r0 - a.B) * a.K; sub = a.real + (int64_t)(r0 - a.B) * a.K; }
}

// Producer side of gram_tile_x3 (in-kernel split).  Thread: the float4 at columns c4..c4+3 of panel rows r0 + 32 j, j < 4,
// of both panels.  TWO stages of loads are in flight per producer thread (set 0: even stages, set 1: odd stages; a set
// is re-issued right after it has been split): with GBs of video behind the stream the HBM latency under load is several
// stage times.  For that to work the compiler must see STRAIGHT-LINE loads in the steady state -- with loads predicated on
// "subtrahend present", "off-diagonal pair" or "k inside the chunk" it waits with s_waitcnt vmcnt(0) right behind the
// issue and the second set buys nothing (round 2 finding).  So: AS / BS / SAME are template
// parameters, k offsets are clamped into the row and a ragged last stage is zeroed by selects after the wait, and the
// loop is peeled (prologue: two sets; steady state s + 3 < nstage: both re-issued; last two or three stages drain) so
// that the number of loads outstanding at every wait is static.  Barriers: nstage + 1, like the consumers.
struct TStageRegs { float4 va[4], vb[4], qa[4], qb[4]; };

template <bool AS, bool BS, bool SAME>
__device__ __forceinline__ void tile_issue(TStageRegs& g, const float* am, const float* as, const float* bm,
                                           const float* bs, int64_t rstep, int64_t koff) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t o = j * rstep + koff;
        g.va[j] = *reinterpret_cast<const float4*>(am + o);
        if (AS) g.qa[j] = *reinterpret_cast<const float4*>(as + o);
        if (!SAME) {
            g.vb[j] = *reinterpret_cast<const float4*>(bm + o);
            if (BS) g.qb[j] = *reinterpret_cast<const float4*>(bs + o);
        }
    }
}

template <bool AS, bool BS, bool SAME>
__device__ __forceinline__ void tile_emit(TStageRegs& g, bool kok, unsigned char* zb, int wbase) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 a = g.va[j];
        if (AS) { a.x -= g.qa[j].x; a.y -= g.qa[j].y; a.z -= g.qa[j].z; a.w -= g.qa[j].w; }
        a.x = kok ? a.x : 0.f; a.y = kok ? a.y : 0.f; a.z = kok ? a.z : 0.f; a.w = kok ? a.w : 0.f;
        tsplit3_store(zb, wbase + 32 * j * TPITCH, a);
        if (!SAME) {
            float4 b = g.vb[j];
            if (BS) { b.x -= g.qb[j].x; b.y -= g.qb[j].y; b.z -= g.qb[j].z; b.w -= g.qb[j].w; }
            b.x = kok ? b.x : 0.f; b.y = kok ? b.y : 0.f; b.z = kok ? b.z : 0.f; b.w = kok ? b.w : 0.f;
            tsplit3_store(zb, wbase + (TP + 32 * j) * TPITCH, b);
        }
    }
}

template <bool AS, bool BS, bool SAME>
__device__ __forceinline__ void tile_produce(const float* am, const float* as, const float* bm, const float* bs, int64_t K,
                                             int64_t kbeg, int64_t kend, int nstage, int t, unsigned char* zsA,
                                             unsigned char* zsB) {
    const int r0 = t >> 3, c4 = (t & 7) * 4;
    const int64_t roff = (int64_t)r0 * K, rstep = 32 * K;
    const int wbase = r0 * TPITCH + c4 * 2;
    am += roff; bm += roff;
    if (AS) as += roff;
    if (BS) bs += roff;
    const int64_t kmax = K - 4;                                            // K % 4 == 0, K >= TK (host checks)
    auto koff = [&](int s) { const int64_t k = kbeg + (int64_t)s * TK + c4; return k < kmax ? k : kmax; };
    auto kok = [&](int s) { return kbeg + (int64_t)s * TK + c4 + 4 <= kend; };   // false only in a ragged last stage
    TStageRegs s0, s1;
    if (nstage < 2) {
        tile_issue<AS, BS, SAME>(s0, am, as, bm, bs, rstep, koff(0));
        tile_emit<AS, BS, SAME>(s0, kok(0), zsA, wbase);
        __syncthreads();
        __syncthreads();
        return;
    }
    tile_issue<AS, BS, SAME>(s0, am, as, bm, bs, rstep, koff(0));
    tile_issue<AS, BS, SAME>(s1, am, as, bm, bs, rstep, koff(1));
    int s = 0;
    for (; s + 3 < nstage; s += 2) {
        // stage s goes into buffer A, stage s + 1 into buffer B (the consumers read a buffer one barrier later)
        tile_emit<AS, BS, SAME>(s0, true, zsA, wbase);
        tile_issue<AS, BS, SAME>(s0, am, as, bm, bs, rstep, koff(s + 2));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        tile_emit<AS, BS, SAME>(s1, true, zsB, wbase);
        tile_issue<AS, BS, SAME>(s1, am, as, bm, bs, rstep, koff(s + 3));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    // s + 1 < nstage <= s + 3: two or three stages left (only the very last one can be ragged)
    const bool three = s + 2 < nstage;
    tile_emit<AS, BS, SAME>(s0, kok(s), zsA, wbase);
    if (three) tile_issue<AS, BS, SAME>(s0, am, as, bm, bs, rstep, koff(s + 2));
    __syncthreads();
    tile_emit<AS, BS, SAME>(s1, kok(s + 1), zsB, wbase);
    __syncthreads();
    if (three) {
        tile_emit<AS, BS, SAME>(s0, kok(s + 2), zsA, wbase);
        __syncthreads();
    }
    __syncthreads();
}

// Wave-specialised like gram128_partial_x3ws: waves 0-3 PRODUCE (global loads, E = fake - real, three-way split,
// ds_write into the next LDS buffer), waves 4-7 CONSUME (ds_read + MFMA on the current buffer): a producer and a
// consumer wave share each SIMD, whose VALU and matrix pipe run concurrently.  One barrier per 32-k stage.
// (The first version did both roles in every wave with one wave per SIMD: the split sat between the MFMA phases
// and B = 512 ran slower than the blocked path.)
__global__ __launch_bounds__(512) void gram_tile_x3(TileArgs ta) {
    __shared__ __attribute__((aligned(16))) unsigned char zsA[TBUF];
    __shared__ __attribute__((aligned(16))) unsigned char zsB[TBUF];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // XCD-aware block -> (pair, chunk) map: workgroups are dealt to the 8 XCDs round-robin, so XCD x = id % 8 takes
    // the K-chunks [x cps, (x+1) cps) and, within them, the pairs in order: the ~32 workgroups an XCD runs at a time
    // are (almost) all pairs of ONE K-range -- a panel chunk is fetched from HBM once per XCD and served to the other
    // pairs that need it from that XCD's L2 (B = 512: every panel chunk has 9 readers).
    const int npairs = ta.nt * (ta.nt + 1) / 2, cps = ta.nchunk >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = slot % npairs, chunk_id = xcd * cps + slot / npairs;
    int pa, pb;
    tile_pair(pair, ta.nt, pa, pb);
    const bool same = pa == pb;
    const int64_t K = ta.K;
    const int64_t kbeg = (int64_t)chunk_id * ta.chunk;
    const int64_t kend = (kbeg + ta.chunk < K) ? kbeg + ta.chunk : K;
    if (kbeg >= kend) return;
    const int nstage = (int)((kend - kbeg + TK - 1) / TK);

    if (wave < 4) {
        // ------------------------------------------------------------------ producers (in-kernel split)
        const float *am, *as, *bm, *bs;
        panel_rows(ta, pa, am, as);
        panel_rows(ta, pb, bm, bs);
        // which tensors a stage reads is fixed per workgroup: one instantiation per combination, so that the loads in the
        // stage loop are unconditional (see tile_produce)
        if (same) {
            if (as) tile_produce<true, true, true>(am, as, bm, bs, K, kbeg, kend, nstage, t, zsA, zsB);
            else tile_produce<false, false, true>(am, as, bm, bs, K, kbeg, kend, nstage, t, zsA, zsB);
        } else if (as) {
            tile_produce<true, true, false>(am, as, bm, bs, K, kbeg, kend, nstage, t, zsA, zsB);      // pa <= pb: bs too
        } else if (bs) {
            tile_produce<false, true, false>(am, as, bm, bs, K, kbeg, kend, nstage, t, zsA, zsB);
        } else {
            tile_produce<false, false, false>(am, as, bm, bs, K, kbeg, kend, nstage, t, zsA, zsB);
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    // wave (wr, wc) owns rows 64 wr .. of the A panel and columns 64 wc .. of the B panel: 2 x 2 MFMA tiles
    const int w = wave - 4, wr = w >> 1, wc = w & 1;
    const int lo = (lane & 31) * TPITCH + 16 * (lane >> 5);   // row (lane & 31), k half (lane >> 5) of a 16-k block
    const int aoff0 = (64 * wr) * TPITCH + lo, aoff1 = aoff0 + 32 * TPITCH;
    const int bbase = same ? 0 : TP;                          // a diagonal pair reads the B fragments from the A rows
    const int boff0 = (bbase + 64 * wc) * TPITCH + lo, boff1 = boff0 + 32 * TPITCH;

    // Two-level fp32 accumulation.  One K-chunk is up to ~37 000 columns at configs[4] (B = 512: 64 chunks of
    // K = 2 359 296), i.e. ~14 000 MFMA accumulations into ONE fp32 register -- measured at full size against the fp64
    // oracle: the all-positive sums G_ee[j,j] (the near-regime diagonal of C_xy) came out 2.3e-5 low.  The MFMA
    // accumulators therefore only run over TFLUSH stages (96 accumulations) and are then folded into a second set of
    // fp32 sums (144 additions per chunk at that size): 64 v_add per 384 MFMAs, and the chunk sums stay fp64.
    constexpr int TFLUSH = 8;
    tf32x16 acc00, acc01, acc10, acc11, sum00, sum01, sum10, sum11;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f;
        sum00[r] = 0.f; sum01[r] = 0.f; sum10[r] = 0.f; sum11[r] = 0.f;
    }
    __syncthreads();                                          // stage 0 is in buffer A
    for (int s = 0; s < nstage; ++s) {
        const unsigned char* zs = (s & 1) ? zsB : zsA;
#pragma unroll
        for (int kb = 0; kb < TK / 16; ++kb) {
            const TFrag a0 = tld_frag(zs, aoff0 + kb * 32), a1 = tld_frag(zs, aoff1 + kb * 32);
            const TFrag b0 = tld_frag(zs, boff0 + kb * 32), b1 = tld_frag(zs, boff1 + kb * 32);
            // product-major order: the four accumulators take turns, so no MFMA waits on the one issued before it
#define KCCOT_T4(PA, PB)                                                                             \
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.PA, b0.PB, acc00, 0, 0, 0);               \
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.PA, b1.PB, acc01, 0, 0, 0);               \
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.PA, b0.PB, acc10, 0, 0, 0);               \
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.PA, b1.PB, acc11, 0, 0, 0);
            KCCOT_T4(m, m) KCCOT_T4(h, l) KCCOT_T4(l, h) KCCOT_T4(h, m) KCCOT_T4(m, h) KCCOT_T4(h, h)   // smallest terms first
#undef KCCOT_T4
        }
        if ((s % TFLUSH) == TFLUSH - 1) {
            sum00 += acc00; sum01 += acc01; sum10 += acc10; sum11 += acc11;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
        }
        __syncthreads();                                      // stage s consumed; stage s + 1 is complete
    }
    sum00 += acc00; sum01 += acc01; sum10 += acc10; sum11 += acc11;

    // accumulator register r of lane l is element ((r&3) + 8*(r>>2) + 4*(l>>5), l&31) of its 32 x 32 tile
    float* o = ta.part + ((int64_t)pair * ta.nchunk + chunk_id) * TELEMS;
    const int col = 64 * wc + (lane & 31), rowb = 64 * wr + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rowb + (r & 3) + 8 * (r >> 2);
        o[row * TP + col] = sum00[r];
        o[row * TP + col + 32] = sum01[r];
        o[(row + 32) * TP + col] = sum10[r];
        o[(row + 32) * TP + col + 32] = sum11[r];
    }
}

// fp64 sum over the chunks, fixed order; eight loads in flight
__global__ __launch_bounds__(256) void gram_tile_reduce(const float* __restrict__ part, int nstride, int nchunk,
                                                        double* __restrict__ gsum) {
    // nstride: chunk slots per pair; nchunk: the non-empty ones
    const int q = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const float* p = part + (int64_t)q * nstride * TELEMS + e;
    double s = 0.0;
    int c = 0;
    for (; c + 8 <= nchunk; c += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[(int64_t)(c + i) * TELEMS];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (double)v[i];
    }
    for (; c < nchunk; ++c) s += (double)p[(int64_t)c * TELEMS];
    gsum[(int64_t)q * TELEMS + e] = s;
}

struct TileFin {
    const double* gsum;
    int B, nt;
    float* out[3];
    const float* h[3];
    const float* M[3];
    float sc;
    int T, J;
};

// Gram entry of stack rows (a, b)
__device__ __forceinline__ double tgram(const double* __restrict__ gs, int nt, int a, int b) {
    const bool sw = (a >> 7) > (b >> 7);
    const int a2 = sw ? b : a, b2 = sw ? a : b;
    const int pa = a2 >> 7, pb = b2 >> 7;
    const int q = pa * nt - (pa * (pa - 1)) / 2 + (pb - pa);
    return gs[(int64_t)q * TELEMS + (a2 & 127) * TP + (b2 & 127)];
}

// One 16 x 16 output tile per block; blockIdx.z = problem (xy, xx, yy).  Formulas of gram_finalize (cost_mfma.hip).
__global__ __launch_bounds__(256) void gram_tile_finalize(TileFin f) {
    __shared__ __attribute__((aligned(16))) float sh[CAUSAL_TILE * CAUSAL_PITCH];
    __shared__ __attribute__((aligned(16))) float sm[CAUSAL_TILE * CAUSAL_PITCH];
    const int p = blockIdx.z, B = f.B, nt = f.nt;
    const int i0 = blockIdx.y * CAUSAL_TILE, j0 = blockIdx.x * CAUSAL_TILE;
    const int i = i0 + (threadIdx.x >> 4), j = j0 + (threadIdx.x & 15);      // B % 16 == 0: always in range
    const double* G = f.gsum;
    const double g_ii = tgram(G, nt, i, i), g_jj = tgram(G, nt, j, j), g_ij = tgram(G, nt, i, j);
    const double e_ii = tgram(G, nt, B + i, B + i), e_jj = tgram(G, nt, B + j, B + j), e_ij = tgram(G, nt, B + i, B + j);
    const double x_ii = tgram(G, nt, i, B + i), x_jj = tgram(G, nt, j, B + j);
    const double x_ij = tgram(G, nt, i, B + j), x_ji = tgram(G, nt, j, B + i);
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
struct TilePlan { int nt, npairs, nchunk; int64_t chunk; size_t part_bytes, gsum_bytes, ediff_bytes, ws_bytes; };

// E = fake - real materialised once for B >= 512.  Measured cost stage, the extra pass included (profiles/r03ac_ab_ediff.txt):
// B = 512 (K = 2.36 M) 22.6 -> 21.1-21.3 ms for 4.8 GB more workspace; B = 384 equal; B = 256 SLOWER (1.12 vs 1.04 ms).
static bool tiled_ediff(int B) { return B >= 512; }

static TilePlan plan_tiled(int B, int64_t K) {
    TilePlan pl{};
    pl.nt = 2 * B / TP;
    pl.npairs = pl.nt * (pl.nt + 1) / 2;
    const int64_t ksteps = (K + TK - 1) / TK;
    // chunks per XCD: each XCD runs npairs * cps workgroups on its 32 CUs (one workgroup per CU: 120 KB of LDS);
    // the smallest cps with at least 3 rounds whose last round is fullest
    int best = 1;
    double best_waste = 1e9;
    for (int cps = 1; cps <= 64; ++cps) {
        const int n = pl.npairs * cps;
        if (n < 96 && cps < 64) continue;
        const double waste = (double)((n + 31) / 32 * 32) / n;
        if (waste < best_waste - 1e-9) { best_waste = waste; best = cps; }
        if (n >= 512) break;
    }
    // Many pairs (B >= 512): SHORTER K-chunks than the occupancy rule asks for.  The pairs that share a panel run on one XCD
    // and re-read it from that XCD's L2 only while they stay within ~30 stages of each other; they drift apart over a
    // chunk (diagonal pairs stage half as much) and re-synchronise when the next chunk's workgroups are dispatched, so
    // shorter chunks = tighter lock-step.  Measured at B = 512, K = 2.36 M: 8 chunks per XCD 21.3 ms, 32: 20.5 ms, 64:
    // 20.4 ms (partials 0.15 / 0.6 / 1.2 GB); B = 256 does not care (1.05-1.15 ms either way).
    if (pl.npairs >= 36 && best < 32) best = 32;
    int64_t nchunk = 8 * (int64_t)best;
    while (nchunk > 8 && nchunk > ksteps) nchunk -= 8;
    const int64_t spc = (ksteps + nchunk - 1) / nchunk;
    pl.chunk = spc * TK;
    pl.nchunk = (int)nchunk;                                  // trailing chunks may be empty (kbeg >= K): they return at once
    pl.part_bytes = align_up((size_t)pl.npairs * pl.nchunk * TELEMS * sizeof(float), 256);
    pl.gsum_bytes = align_up((size_t)pl.npairs * TELEMS * sizeof(double), 256);
    pl.ediff_bytes = tiled_ediff(B) ? align_up((size_t)B * K * sizeof(float), 256) : 0;
    pl.ws_bytes = pl.part_bytes + pl.gsum_bytes + pl.ediff_bytes;
    return pl;
}

// where the fp64 sums sit in the workspace (the all-reducing caller's view)
void gram_tiled_sums_span(int B, int64_t K, size_t* off, size_t* n) {
    const TilePlan pl = plan_tiled(B, K);
    *off = pl.part_bytes;
    *n = (size_t)pl.npairs * TELEMS;
}

size_t gram_tiled_workspace_bytes(int B, int64_t K) {
    if (B < TP || B % TP != 0 || B > 4096) return 0;
    return plan_tiled(B, K).ws_bytes;
}

bool gram_tiled_eligible(const CostBatch& cb, int64_t K, bool loss3) {
    if (!loss3 || cb.nprob != 3 || !opt(OPT_COST_TILED)) return false;
    if (opt(OPT_GRAM_F32)) return false;                      // the f32-input MFMA request keeps the other paths
    const int B = cb.p[0].Bx;
    if (B < TP || B % TP != 0 || B > 4096 || cb.p[0].By != B || K % 4 != 0 || K < 256) return false;
    return ((uintptr_t)cb.p[0].x % 16 == 0) && ((uintptr_t)cb.p[0].y % 16 == 0);
}

int run_gram_tiled(const CostBatch& cb, int64_t K, float sc, int T, int J, void* ws, size_t ws_bytes, hipStream_t st,
                   int stage) {   // stage 0: everything; 1: stop after the fp64 sums; 2: finalize only
    const int B = cb.p[0].Bx;
    const TilePlan pl = plan_tiled(B, K);
    if (!ws || ws_bytes < pl.ws_bytes)
        return fail(KCCOT_EWORKSPACE, "pairwise_cost3(tiled): workspace %zu < required %zu", ws_bytes, pl.ws_bytes);
    if (pl.nchunk > 65535) return fail(KCCOT_EUNSUPPORTED, "pairwise_cost3(tiled): %d chunks", pl.nchunk);
    float* part = static_cast<float*>(ws);
    double* gsum = reinterpret_cast<double*>(static_cast<char*>(ws) + pl.part_bytes);
    int rc;
    if (stage != 2) {
        TileArgs ta{cb.p[0].x, cb.p[0].y, B, pl.nt, pl.nchunk, K, pl.chunk, part, nullptr};
        if (pl.ediff_bytes) {
            float* e = reinterpret_cast<float*>(static_cast<char*>(ws) + pl.part_bytes + pl.gsum_bytes);
            hipLaunchKernelGGL(ediff_rows, dim3(2048), dim3(256), 0, st, cb.p[0].x, cb.p[0].y, (int64_t)B * K / 4, e);
            if ((rc = launch_status("ediff_rows"))) return rc;
            ta.ediff = e;
        }
        hipLaunchKernelGGL(gram_tile_x3, dim3(pl.npairs * pl.nchunk), dim3(512), 0, st, ta);
        if ((rc = launch_status("gram_tile_x3"))) return rc;
        const int nvalid = (int)((K + pl.chunk - 1) / pl.chunk);
        hipLaunchKernelGGL(gram_tile_reduce, dim3(TELEMS / 256, pl.npairs), dim3(256), 0, st, (const float*)part, pl.nchunk, nvalid, gsum);
        if ((rc = launch_status("gram_tile_reduce"))) return rc;
        if (stage == 1) return 0;
    }
    TileFin f{};
    f.gsum = gsum; f.B = B; f.nt = pl.nt; f.sc = sc; f.T = T; f.J = J;
    for (int p = 0; p < 3; ++p) { f.out[p] = cb.p[p].out; f.h[p] = cb.p[p].h1; f.M[p] = cb.p[p].M1; }
    hipLaunchKernelGGL(gram_tile_finalize, dim3(B / CAUSAL_TILE, B / CAUSAL_TILE, 3), dim3(256), 0, st, f);
    return launch_status("gram_tile_finalize");
}

}  // namespace kccot
