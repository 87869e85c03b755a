// Stacked-Gram pairwise cost on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32, exact fp32 fma chain).
//
// For operands of at most 64 rows each (BASELINE configs[1]: B = 64) the two operands are
// stacked into one 128-row panel Z and ONE workgroup per K-chunk produces the partial Gram
// matrix Z Z^T of its chunk as 32x32 sub-tiles on/above the diagonal, so every input element is
// read from HBM exactly once for all three matrices of the mixed Sinkhorn divergence
// (gan_utils.py:221-223).  The squared distances are then formed from the chunk-summed Gram
// entries in fp64 (gram_finalize):  ||z_s - z_t||^2 = G_ss + G_tt - 2 G_st, whose diagonal is
// exactly 0 as in the reference's direct (x-y)^2 form (gan_utils.py:16).
//
// Cancellation.  In the GAN regime fake_i is close to real_i, and G_ss + G_tt - 2 G_st would lose
// the small distance under the fp32 rounding of three O(K) numbers.  For the loss (GRAM_LOSS3) the
// stack is therefore [X ; E] with E = fake - real formed in registers while staging (row i of
// both tensors is held by the same thread), and
//     D_xy[i,j] = D_xx[i,j] + G_ee[j,j] - 2 (G_xe[i,j] - G_xe[j,j])
//     D_yy[i,j] = D_xx[i,j] + D_ee[i,j] + 2 (G_xe[i,i] - G_xe[i,j] - G_xe[j,i] + G_xe[j,j])
// are identities in which every term that must cancel is itself small when E is small; when E is
// not small they cost nothing.  D_xy[j,j] = G_ee[j,j] = sum_k e_jk^2 is a sum of squares.
//
// MFMA fragment maps (cdna_hip_programming.md section 3): for mfma_f32_32x32x2f32 lane l supplies
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; accumulator register r of lane l is
// D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].  With A[i][k] = Z[32a+i][k] and B[k][j] = Z[32b+j][k]
// the result is the Gram sub-tile (a,b).  Each lane reads 4 consecutive k of its row with one
// ds_read_b128 (k = kk + 4*(l>>5) + {0..3}); A and B use the same k mapping, and a sum over k does
// not care in which order the eight k of a group are visited.
#include "common.h"
#include <stdlib.h>
#include "cost_internal.h"
#include "options.h"

namespace kccot {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GPITCH = GRAM_KT + 4;   // 36 floats: conflict-free ds_read_b128 over 32 rows
constexpr int GRAM_SLABS = 12;        // 10 sub-tiles + 2 second halves of the k-split ones
constexpr int GRAM_ELEMS = GRAM_NSUB * 1024;

// The causal sums  caus[slot][i][j] = sum_{t<T-1,q} h[i,t,q] (M[j,t+1,q] - M[j,t,q])  (gan_utils.py:37-46) do
// not depend on the Gram sums.  They are produced by spare workgroups of an earlier launch so that
// gram_finalize only gathers and adds.
struct CausalPre {
    const float* h[3];   // per slot: rows; null = slot unused
    const float* M[3];   // per slot: columns
    float* caus;         // [3][B1][B2]
    int B1, B2, T, J, nti, ntj;
};

struct GramArgs {
    const float* src1;
    const float* src2;
    int n1, n2;
    int pair_diff;
    unsigned mask;       // needed sub-tiles, bit sub_index(a,b)
    int64_t K, chunk;
    float* gpart;        // [nchunk][GRAM_SLABS][1024]
    int nchunk;          // workgroups [0, nchunk) do Gram chunks ...
    int ntiles;          // ... and workgroups nchunk + b (b < gridDim.x - nchunk) the causal tiles b, b + nb, ... < ntiles
    CausalPre cp;
};

// one 16x16 tile of causal sums by 256 threads (every one of them must call it); sh / sm: LDS scratch
template <int MG = CAUSAL_TILE>
__device__ __forceinline__ void causal_pre_tile(const CausalPre& cp, int tile, float* sh, float* sm, int t) {
    const int per = cp.nti * cp.ntj;
    const int slot = tile / per, rem = tile % per;
    if (slot >= 3 || cp.h[slot] == nullptr) return;      // uniform
    const int i0 = (rem / cp.ntj) * CAUSAL_TILE, j0 = (rem % cp.ntj) * CAUSAL_TILE;
    const float v = causal_tile16<MG>(cp.h[slot], cp.M[slot], i0, j0, cp.B1, cp.B2, cp.T, cp.J, sh, sm, t);
    const int i = i0 + (t >> 4), j = j0 + (t & 15);
    if (i < cp.B1 && j < cp.B2) cp.caus[((int64_t)slot * cp.B1 + i) * cp.B2 + j] = v;
}

struct WaveWork {        // up to three (sub-tile, k-group range) entries per wave
    int a[3], b[3], slab[3], lo[3], hi[3];
    int n;
};

__device__ __forceinline__ void sub_ab(int idx, int& a, int& b) {
    if (idx < 4) { a = 0; b = idx; }
    else if (idx < 7) { a = 1; b = idx - 3; }
    else if (idx < 9) { a = 2; b = idx - 5; }
    else { a = 3; b = 3; }
}

__device__ __forceinline__ int nth_set_bit(unsigned mask, int n) {
    for (int idx = 0; idx < GRAM_NSUB; ++idx)
        if ((mask >> idx) & 1u) {
            if (n == 0) return idx;
            --n;
        }
    return -1;
}

// Every field is written with a compile-time index (a runtime-indexed register array would be
// demoted to scratch memory).
__device__ __forceinline__ WaveWork wave_work(unsigned mask, int wave) {
    WaveWork w;
#pragma unroll
    for (int s = 0; s < 3; ++s) { w.a[s] = w.b[s] = w.slab[s] = 0; w.lo[s] = 0; w.hi[s] = 0; }
    if (mask == 0x3FFu) {
        // all ten sub-tiles: two whole ones per wave plus one half (in k) of sub-tile 8 or 9,
        // i.e. 2.5 sub-tiles of MFMA work on every SIMD
        sub_ab(2 * wave, w.a[0], w.b[0]);     w.slab[0] = 2 * wave;     w.lo[0] = 0; w.hi[0] = 4;
        sub_ab(2 * wave + 1, w.a[1], w.b[1]); w.slab[1] = 2 * wave + 1; w.lo[1] = 0; w.hi[1] = 4;
        const int split = 8 + (wave >> 1), second = wave & 1;
        sub_ab(split, w.a[2], w.b[2]);
        w.slab[2] = second ? 10 + (wave >> 1) : split;
        w.lo[2] = second ? 2 : 0;
        w.hi[2] = second ? 4 : 2;
        w.n = 3;
        return w;
    }
    // fewer sub-tiles (at most 9): deal them round-robin, whole
    const int i0 = nth_set_bit(mask, wave), i1 = nth_set_bit(mask, wave + 4), i2 = nth_set_bit(mask, wave + 8);
    w.n = (i0 >= 0) + (i1 >= 0) + (i2 >= 0);
    if (i0 >= 0) { sub_ab(i0, w.a[0], w.b[0]); w.slab[0] = i0; w.hi[0] = 4; }
    if (i1 >= 0) { sub_ab(i1, w.a[1], w.b[1]); w.slab[1] = i1; w.hi[1] = 4; }
    if (i2 >= 0) { sub_ab(i2, w.a[2], w.b[2]); w.slab[2] = i2; w.hi[2] = 4; }
    return w;
}

__device__ __forceinline__ float4 ld4(const float* __restrict__ row, int64_t k, int64_t kend, bool ok) {
    if (ok && k + 4 <= kend) return *reinterpret_cast<const float4*>(row + k);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

#define KCCOT_MFMA4(ACC, A4, B4)                                              \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A4.x, B4.x, ACC, 0, 0, 0);     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A4.y, B4.y, ACC, 0, 0, 0);     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A4.z, B4.z, ACC, 0, 0, 0);     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A4.w, B4.w, ACC, 0, 0, 0);

template <bool FULL10>
__global__ __launch_bounds__(256) void gram128_partial(GramArgs ga) {
    __shared__ __attribute__((aligned(16))) float zs[GRAM_ROWS * GPITCH];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t kbeg = (int64_t)blockIdx.x * ga.chunk;
    const int64_t kend = (kbeg + ga.chunk < ga.K) ? kbeg + ga.chunk : ga.K;
    if (kbeg >= kend) return;

    // staging: thread holds the float4 at columns c4..c4+3 of stack rows r0, r0+32, 64+r0, 96+r0
    const int r0 = t >> 3, c4 = (t & 7) * 4;
    const bool ok0 = r0 < ga.n1, ok1 = r0 + 32 < ga.n1, ok2 = r0 < ga.n2, ok3 = r0 + 32 < ga.n2;
    const float* p0 = ga.src1 + (int64_t)r0 * ga.K;
    const float* p1 = ga.src1 + (int64_t)(r0 + 32) * ga.K;
    const float* p2 = ga.src2 + (int64_t)r0 * ga.K;
    const float* p3 = ga.src2 + (int64_t)(r0 + 32) * ga.K;

    const WaveWork ww = wave_work(ga.mask, wave);
    const int arow0 = (ww.a[0] * 32 + (lane & 31)) * GPITCH + 4 * (lane >> 5);
    const int brow0 = (ww.b[0] * 32 + (lane & 31)) * GPITCH + 4 * (lane >> 5);
    const int arow1 = (ww.a[1] * 32 + (lane & 31)) * GPITCH + 4 * (lane >> 5);
    const int brow1 = (ww.b[1] * 32 + (lane & 31)) * GPITCH + 4 * (lane >> 5);
    const int arow2 = (ww.a[2] * 32 + (lane & 31)) * GPITCH + 4 * (lane >> 5);
    const int brow2 = (ww.b[2] * 32 + (lane & 31)) * GPITCH + 4 * (lane >> 5);

    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }

    float4 v0 = ld4(p0, kbeg + c4, kend, ok0);
    float4 v1 = ld4(p1, kbeg + c4, kend, ok1);
    float4 v2 = ld4(p2, kbeg + c4, kend, ok2);
    float4 v3 = ld4(p3, kbeg + c4, kend, ok3);

    for (int64_t k0 = kbeg; k0 < kend; k0 += GRAM_KT) {
        if (ga.pair_diff) {  // E = src2 - src1 (n1 == n2, so a row is valid in both or in neither)
            v2.x -= v0.x; v2.y -= v0.y; v2.z -= v0.z; v2.w -= v0.w;
            v3.x -= v1.x; v3.y -= v1.y; v3.z -= v1.z; v3.w -= v1.w;
        }
        *reinterpret_cast<float4*>(&zs[r0 * GPITCH + c4]) = v0;
        *reinterpret_cast<float4*>(&zs[(r0 + 32) * GPITCH + c4]) = v1;
        *reinterpret_cast<float4*>(&zs[(r0 + 64) * GPITCH + c4]) = v2;
        *reinterpret_cast<float4*>(&zs[(r0 + 96) * GPITCH + c4]) = v3;
        __syncthreads();
        const int64_t kn = k0 + GRAM_KT;
        if (kn < kend) {  // next stage's HBM reads fly under this stage's MFMAs
            v0 = ld4(p0, kn + c4, kend, ok0);
            v1 = ld4(p1, kn + c4, kend, ok1);
            v2 = ld4(p2, kn + c4, kend, ok2);
            v3 = ld4(p3, kn + c4, kend, ok3);
        }
        if constexpr (FULL10) {
            // all ten sub-tiles: entries 0 and 1 over the whole k-tile, entry 2 over its half --
            // straight-line, so the scheduler can hoist every ds_read_b128 above the MFMA chain
            const int half = (wave & 1) * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 A0 = *reinterpret_cast<const float4*>(&zs[arow0 + 8 * g]);
                const float4 B0 = *reinterpret_cast<const float4*>(&zs[brow0 + 8 * g]);
                const float4 A1 = *reinterpret_cast<const float4*>(&zs[arow1 + 8 * g]);
                const float4 B1 = *reinterpret_cast<const float4*>(&zs[brow1 + 8 * g]);
                KCCOT_MFMA4(acc0, A0, B0)
                KCCOT_MFMA4(acc1, A1, B1)
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4 A2 = *reinterpret_cast<const float4*>(&zs[arow2 + half + 8 * g]);
                const float4 B2 = *reinterpret_cast<const float4*>(&zs[brow2 + half + 8 * g]);
                KCCOT_MFMA4(acc2, A2, B2)
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // four groups of 8 k
                if (ww.n > 0 && g >= ww.lo[0] && g < ww.hi[0]) {
                    const float4 A = *reinterpret_cast<const float4*>(&zs[arow0 + 8 * g]);
                    const float4 B = *reinterpret_cast<const float4*>(&zs[brow0 + 8 * g]);
                    KCCOT_MFMA4(acc0, A, B)
                }
                if (ww.n > 1 && g >= ww.lo[1] && g < ww.hi[1]) {
                    const float4 A = *reinterpret_cast<const float4*>(&zs[arow1 + 8 * g]);
                    const float4 B = *reinterpret_cast<const float4*>(&zs[brow1 + 8 * g]);
                    KCCOT_MFMA4(acc1, A, B)
                }
                if (ww.n > 2 && g >= ww.lo[2] && g < ww.hi[2]) {
                    const float4 A = *reinterpret_cast<const float4*>(&zs[arow2 + 8 * g]);
                    const float4 B = *reinterpret_cast<const float4*>(&zs[brow2 + 8 * g]);
                    KCCOT_MFMA4(acc2, A, B)
                }
            }
        }
        __syncthreads();
    }

    // accumulator register r of lane l is element ((r&3) + 8*(r>>2) + 4*(l>>5), l&31)
    float* base = ga.gpart + (int64_t)blockIdx.x * GRAM_SLABS * 1024;
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    if (ww.n > 0) {
        float* o = base + ww.slab[0] * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + rbase) * 32 + col] = acc0[r];
    }
    if (ww.n > 1) {
        float* o = base + ww.slab[1] * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + rbase) * 32 + col] = acc1[r];
    }
    if (ww.n > 2) {
        float* o = base + ww.slab[2] * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + rbase) * 32 + col] = acc2[r];
    }
}

// ------------------------------------------------------------------------------------------
// Same stacked Gram, all ten sub-tiles, on the bf16 MFMA pipe with an EXACT three-way split.
//
// Every fp32 value is cut into three bf16 pieces by truncation, x = h + m + l exactly (8 + 8 + 8
// significand bits: h = top half of the word, m = top half of x - h, l = x - h - m), while the
// tile is staged; the Gram entry is accumulated in fp32 as
//     sum_k  xh*yh + (xh*ym + xm*yh) + (xh*yl + xl*yh + xm*ym)
// -- six v_mfma_f32_32x32x16_bf16 per 16 k instead of eight v_mfma_f32_32x32x2_f32, each of them
// four times shorter in issue cycles per k: the f32-MFMA-bound kernel becomes HBM-bound.  bf16 x
// bf16 products are exact in fp32; the dropped terms xm*yl + xl*ym + xl*yl are below 2^-24 of
// |x*y|, i.e. under the rounding of the fp32 accumulation itself, so this is fp32 arithmetic to
// working precision (parity tests run both kernels against the same golden vectors).
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int XKT = 64;                         // floats of K per stage
constexpr int XPITCH = XKT * 2 + 16;            // bytes per row of one bf16 plane (+16: conflict-free b128 reads)
constexpr int XPLANE = GRAM_ROWS * XPITCH;      // 18432 bytes

__device__ __forceinline__ void split3_store(unsigned char* zs, int byte_off, float4 v) {
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
    // pack the upper halves of consecutive elements: dword = bf16(e0) | bf16(e1) << 16
    uint2 ph, pm, pl;
    ph.x = __builtin_amdgcn_perm(x[1], x[0], 0x07060302u); ph.y = __builtin_amdgcn_perm(x[3], x[2], 0x07060302u);
    pm.x = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u); pm.y = __builtin_amdgcn_perm(m[3], m[2], 0x07060302u);
    pl.x = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u); pl.y = __builtin_amdgcn_perm(l[3], l[2], 0x07060302u);
    *reinterpret_cast<uint2*>(zs + byte_off) = ph;
    *reinterpret_cast<uint2*>(zs + XPLANE + byte_off) = pm;
    *reinterpret_cast<uint2*>(zs + 2 * XPLANE + byte_off) = pl;
}

// ---- consumer side of the wave-specialised kernel -------------------------------------------------
// One ds_read_b128 moves 1 KB = 8 cycles of the CU's LDS port, one bf16 MFMA keeps a SIMD's matrix
// pipe busy for 32 cycles and four consumer waves run concurrently: at one fragment read per MFMA
// (six reads for the six products of a sub-tile, as the single-role kernels do) the LDS port is
// saturated before the matrix pipes are.  So the 10 sub-tiles (+ 2 half sub-tiles, for balance)
// are dealt to the waves such that the three sub-tiles of a wave SHARE row blocks and every fragment
// triple (h, m, l) is read once per 16-k step and used by every product that needs it:
//     wave W owns row block X = W and its diagonal sub-tile (X,X) (A and B fragments coincide),
//     an off-diagonal sub-tile with block Y, and one k-half of a sub-tile that pairs X or Y with Z:
//        W   X  Y  Z   sub-tiles
//        0   0  1  2   (0,0) (0,1) (0,2) k-half 0
//        1   1  2  0   (1,1) (1,2) (0,2) k-half 1
//        2   2  3  1   (2,2) (2,3) (1,3) k-half 0
//        3   3  0  1   (3,3) (0,3) (1,3) k-half 1
//
// Round 4:
//  * A DIAGONAL sub-tile is symmetric, and of its six split products hh and mm are symmetric themselves while mh = (hm)^T and
//    lh = (hl)^T.  The wave accumulates S = mm + hh and A = hl + hm in two accumulators (FOUR products per 16 k instead of
//    six: 52 instead of 60 MFMAs per stage and SIMD) and forms  G = S + A + A^T  once per workgroup, through LDS.
//  * The two k-halves of the split sub-tiles are added in LDS before anything is stored, the diagonal sub-tiles are stored as
//    packed upper triangles, and the whole partial leaves the CU as ONE contiguous 33 KB record written with 16-byte
//    stores by all consumer threads:  12 slabs x 4 KB = 48 KB per chunk  ->  8256 floats (GRAM_CPT) -- 11.8 MB -> 7.9 MB of
//    partial tiles at configs[1], for the store AND for gram_reduce's read.
struct Frag3 { bf16x8 h, m, l; };

__device__ __forceinline__ Frag3 ld_frag3(const unsigned char* zs, int off) {
    Frag3 f;
    f.h = *reinterpret_cast<const bf16x8*>(zs + off);
    f.m = *reinterpret_cast<const bf16x8*>(zs + XPLANE + off);
    f.l = *reinterpret_cast<const bf16x8*>(zs + 2 * XPLANE + off);
    return f;
}

// acc += A * B^T with the exact three-way split: hh + (hm + mh) + (hl + lh + mm), smallest terms first
__device__ __forceinline__ void mfma_x3(f32x16& acc, const Frag3& A, const Frag3& B) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l, B.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.h, acc, 0, 0, 0);
}

// diagonal sub-tile: S += mm + hh (symmetric products), A += hl + hm (their transposes are the two products left out)
__device__ __forceinline__ void mfma_x3_diag(f32x16& accS, f32x16& accA, const Frag3& F) {
    accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.m, F.m, accS, 0, 0, 0);
    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.h, F.l, accA, 0, 0, 0);
    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.h, F.m, accA, 0, 0, 0);
    accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.h, F.h, accS, 0, 0, 0);
}

// Compact partial record of one K-chunk (floats): the six off-diagonal sub-tiles (a < b) row-major [32][32] at
// gram_od(a, b) * 1024, then the four diagonal sub-tiles as packed upper triangles (528 each) at 6144 + 528 a.
constexpr int GRAM_TRI = 528;
constexpr int GRAM_CPT = 6 * 1024 + 4 * GRAM_TRI;        // 8256 = 129 * 64
static_assert(GRAM_CPT % 64 == 0 && GRAM_CPT * 4 <= 3 * XPLANE, "one 256-byte line per reducing wave; fits a stage buffer");
__host__ __device__ __forceinline__ int gram_od(int a, int b) { return a == 0 ? b - 1 : (a == 1 ? b + 1 : 5); }
__host__ __device__ __forceinline__ int gram_tri(int r, int c) { return r * 32 - ((r * (r - 1)) >> 1) + (c - r); }   // r <= c

// accumulator register r of lane l is element ((r&3) + 8*(r>>2) + 4*(l>>5), l&31)
__device__ __forceinline__ void img_tile(float* img, int od, int lane, const f32x16& acc) {
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    float* o = img + od * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + rbase) * 32 + col] = acc[r];
}

template <int W>
__device__ __forceinline__ void x3ws_consume(unsigned char* zsA, unsigned char* zsB, int nstage, int lane, float* out) {
    constexpr int X = W, Y = (W + 1) & 3, Z = (W == 0) ? 2 : (W == 1 ? 0 : 1);
    constexpr int KB0 = (W & 1) * 2;                           // the k-half of the stage this wave takes of its split sub-tile
    const int lo = (lane & 31) * XPITCH + 16 * (lane >> 5);    // row (lane&31), k half (lane>>5) of a 16-k block
    const int ox = X * 32 * XPITCH + lo, oy = Y * 32 * XPITCH + lo, oz = Z * 32 * XPITCH + lo;
    f32x16 accS, accA, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accS[r] = 0.f; accA[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }
    __syncthreads();                                  // stage 0 is in buffer A
    for (int s = 0; s < nstage; ++s) {
        const unsigned char* zs = (s & 1) ? zsB : zsA;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const Frag3 fx = ld_frag3(zs, ox + kb * 32), fy = ld_frag3(zs, oy + kb * 32);
            mfma_x3_diag(accS, accA, fx);
            if (W != 3) mfma_x3(acc1, fx, fy); else mfma_x3(acc1, fy, fx);      // (X,Y), or (0,3) = (Y,X) for W = 3
            if (kb >= KB0 && kb < KB0 + 2) {
                const Frag3 fz = ld_frag3(zs, oz + kb * 32);
                if (W == 0) mfma_x3(acc2, fx, fz);          // (0,2)
                else if (W == 3) mfma_x3(acc2, fz, fx);     // (1,3)
                else mfma_x3(acc2, fz, fy);                 // (0,2) resp. (1,3)
            }
        }
        __syncthreads();                              // stage s consumed; stage s+1 is complete
    }
    // ---- epilogue: the producers have left (their barrier count is complete); both stage buffers are free.
    // zsA: the compact record (33 KB).  zsB: per-wave transpose scratch [32][33] floats, then the two k-half hand-overs.
    float* img = reinterpret_cast<float*>(zsA);
    float* tr = reinterpret_cast<float*>(zsB) + W * (32 * 33);
    float* half = reinterpret_cast<float*>(zsB) + 4 * (32 * 33) + (W >> 1) * 1024;
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    if (W & 1) {                                       // second k-half: same lane <-> element map as the first half's wave
#pragma unroll
        for (int r = 0; r < 16; ++r) half[r * 64 + lane] = acc2[r];
    }
    // G = S + A + A^T of the diagonal sub-tile, upper triangle only
#pragma unroll
    for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + rbase) * 33 + col] = accA[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        float* o = img + 6 * 1024 + X * GRAM_TRI;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + rbase;
            const float g = (accS[r] + accA[r]) + tr[col * 33 + row];
            if (row <= col) o[gram_tri(row, col)] = g;
        }
    }
    img_tile(img, (W == 0) ? gram_od(0, 1) : (W == 1 ? gram_od(1, 2) : (W == 2 ? gram_od(2, 3) : gram_od(0, 3))), lane, acc1);
    __syncthreads();                                  // the second halves are in LDS
    if (!(W & 1)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] += half[r * 64 + lane];
        img_tile(img, (W == 0) ? gram_od(0, 2) : gram_od(1, 3), lane, acc2);
    }
    __syncthreads();                                  // the record is complete
    // 8256 floats = 2064 float4 by the 256 consumer threads, contiguous 16-byte stores
    const int ct = W * 64 + lane;
    const float4* src = reinterpret_cast<const float4*>(img);
    float4* dst = reinterpret_cast<float4*>(out);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[ct + 256 * i] = src[ct + 256 * i];
    if (ct < GRAM_CPT / 4 - 2048) dst[ct + 2048] = src[ct + 2048];
}

// ------------------------------------------------------------------------------------------
// Wave-specialised form of the x3 kernel: 8 waves per workgroup, two per SIMD.  Waves 0-3 are
// PRODUCERS (global loads, E = src2 - src1, three-way split, ds_write into the next LDS buffer),
// waves 4-7 are CONSUMERS (ds_read + the 60 bf16 MFMAs of the stage on the current buffer).  The
// matrix pipe and the VALU are separate pipes of a SIMD, so a producer wave and a consumer wave
// that share a SIMD run concurrently: the staging no longer sits between the MFMA phases of the
// same wave.  One s_barrier per stage (every wave executes the same number of them).
// ------------------------------------------------------------------------------------------
// Producer side with TWO stages of loads in flight (register sets v: even stages, w: odd stages).  Everything the
// steady-state loop does is unconditional: rows and k offsets are CLAMPED into the tensor and the out-of-range values are
// replaced by zeros with selects after the wait, so that the compiler sees straight-line load / use code and waits with
// partial counters (vmcnt(8..15): only the OLDER set has to have landed).  With predicated loads (ld4) it falls back to
// s_waitcnt vmcnt(0) directly behind the issue and the second set buys nothing (measured: round 2).
// (FOUR sets in flight -- a ring re-issued four stages ahead, vmcnt(24) in steady state -- measured SLOWER in round 4: kernel
// 20.9 -> 22.0 us, same box, profiles/r4_ab_gram_deep4.txt: 64 KB per CU in flight already cover the latency, 128 KB queue up.)
// The loop is peeled so that the number of outstanding loads at every wait is static: prologue 2 sets, steady state
// (s + 3 < nstage) re-issues both, the last 2 or 3 stages drain.  Barrier count = nstage + 1, as the consumers'.
struct DeepSet { float4 x[8]; };

__device__ __forceinline__ void x3ws_issue(DeepSet& d, const float* const (&rp)[8], int64_t koff) {
#pragma unroll
    for (int j = 0; j < 8; ++j) d.x[j] = *reinterpret_cast<const float4*>(rp[j] + koff);
}

template <bool MASK>
__device__ __forceinline__ void x3ws_emit(DeepSet& d, const bool (&ok)[8], bool kok, bool pair_diff, unsigned char* zb,
                                          int wbase) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 a = d.x[j], b = d.x[j + 4];
        if (MASK) {
            const bool va = ok[j] && kok, vb = ok[j + 4] && kok;
            a.x = va ? a.x : 0.f; a.y = va ? a.y : 0.f; a.z = va ? a.z : 0.f; a.w = va ? a.w : 0.f;
            b.x = vb ? b.x : 0.f; b.y = vb ? b.y : 0.f; b.z = vb ? b.z : 0.f; b.w = vb ? b.w : 0.f;
        }
        if (pair_diff) { b.x -= a.x; b.y -= a.y; b.z -= a.z; b.w -= a.w; }
        // (the subtractions as packed v_pk_add_f32, half as many instructions, measured SLOWER: 20.7-21.2 us vs 19.1-20.4)
        split3_store(zb, wbase + 16 * j * XPITCH, a);
        split3_store(zb, wbase + (64 + 16 * j) * XPITCH, b);
    }
}

template <bool MASK>
__device__ __forceinline__ void x3ws_produce_deep_impl(const GramArgs& ga, unsigned char* zsA, unsigned char* zsB, int t,
                                                  int64_t kbeg, int64_t kend, int nstage) {
    const int r0 = t >> 4, c4 = (t & 15) * 4;
    const float* rp[8];
    bool ok[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = r0 + 16 * (j & 3);
        ok[j] = r < (j < 4 ? ga.n1 : ga.n2);
        rp[j] = (j < 4 ? ga.src1 : ga.src2) + (int64_t)(ok[j] ? r : 0) * ga.K;
    }
    const int wbase = r0 * XPITCH + c4 * 2;
    const int64_t kmax = ga.K - 4;                                        // K >= 4 (host check)
    auto koff = [&](int s) { const int64_t k = kbeg + (int64_t)s * XKT + c4; return k < kmax ? k : kmax; };
    auto kok = [&](int s) { return kbeg + (int64_t)s * XKT + c4 + 4 <= kend; };
    const bool pd = ga.pair_diff;
    DeepSet v, w;
    x3ws_issue(v, rp, koff(0));
    if (nstage < 2) {                                                     // a chunk of a single stage
        x3ws_emit<true>(v, ok, kok(0), pd, zsA, wbase);
        __syncthreads();
        __syncthreads();
        return;
    }
    x3ws_issue(w, rp, koff(1));
    int s = 0;
    for (; s + 3 < nstage; s += 2) {
        x3ws_emit<MASK>(v, ok, kok(s), pd, zsA, wbase);
        x3ws_issue(v, rp, koff(s + 2));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        x3ws_emit<MASK>(w, ok, kok(s + 1), pd, zsB, wbase);
        x3ws_issue(w, rp, koff(s + 3));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    // s + 1 < nstage <= s + 3: two or three stages left, v = stage s, w = stage s + 1
    const bool three = s + 2 < nstage;
    x3ws_emit<MASK>(v, ok, kok(s), pd, zsA, wbase);
    if (three) x3ws_issue(v, rp, koff(s + 2));
    __syncthreads();
    x3ws_emit<MASK>(w, ok, kok(s + 1), pd, zsB, wbase);
    __syncthreads();
    if (three) {
        x3ws_emit<MASK>(v, ok, kok(s + 2), pd, zsA, wbase);
        __syncthreads();
    }
    __syncthreads();
}

__device__ __forceinline__ void x3ws_produce_deep(const GramArgs& ga, unsigned char* zsA, unsigned char* zsB, int t,
                                                  int64_t kbeg, int64_t kend, int nstage) {
    // uniform: full row blocks and whole stages need no masking at all
    const bool whole = ga.n1 == 64 && ga.n2 == 64 && kend - kbeg == (int64_t)nstage * XKT;
    if (whole) x3ws_produce_deep_impl<false>(ga, zsA, zsB, t, kbeg, kend, nstage);
    else x3ws_produce_deep_impl<true>(ga, zsA, zsB, t, kbeg, kend, nstage);
}

__global__ __launch_bounds__(512) void gram128_partial_x3ws(GramArgs ga) {
    __shared__ __attribute__((aligned(16))) unsigned char zsA[3 * XPLANE];
    __shared__ __attribute__((aligned(16))) unsigned char zsB[3 * XPLANE];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if ((int)blockIdx.x >= ga.nchunk) {
        // spare workgroups (the K-split leaves a few CUs without a chunk): the causal sums, a few tiles
        // each, finished long before the Gram chunks are
        if (t >= 256) return;                                // whole waves; the barriers below count the live ones
        float* sh = reinterpret_cast<float*>(zsA);
        float* sm = sh + CAUSAL_TILE * CAUSAL_PITCH;
        for (int tile = blockIdx.x - ga.nchunk; tile < ga.ntiles; tile += gridDim.x - ga.nchunk)
            causal_pre_tile<4>(ga.cp, tile, sh, sm, t);
        return;
    }
    const int64_t kbeg = (int64_t)blockIdx.x * ga.chunk;
    const int64_t kend = (kbeg + ga.chunk < ga.K) ? kbeg + ga.chunk : ga.K;
    if (kbeg >= kend) return;
    const int nstage = (int)((kend - kbeg + XKT - 1) / XKT);

    if (wave < 4) {
        // ------------------------------------------------------------------ producers
        x3ws_produce_deep(ga, zsA, zsB, t, kbeg, kend, nstage);
        return;
    }

    // ---------------------------------------------------------------------- consumers
    float* out = ga.gpart + (int64_t)blockIdx.x * GRAM_CPT;      // this chunk's compact record
    switch (wave - 4) {
        case 0: x3ws_consume<0>(zsA, zsB, nstage, lane, out); break;
        case 1: x3ws_consume<1>(zsA, zsB, nstage, lane, out); break;
        case 2: x3ws_consume<2>(zsA, zsB, nstage, lane, out); break;
        default: x3ws_consume<3>(zsA, zsB, nstage, lane, out); break;
    }
}

// Sum the per-chunk partial Gram sub-tiles in fp64.  One workgroup of 1024 threads owns 64
// consecutive Gram entries (one 256-byte line per chunk): thread (e = t & 63, grp = t >> 6) adds
// chunks grp, grp+16, ... in increasing order, then the 16 group sums are combined in fixed order
// through LDS -> run-to-run deterministic.
//
// The causal terms of the cost (gan_utils.py:37-46) do not depend on the Gram sums, so they are
// computed here as well, by extra workgroups of the same launch (blockIdx >= GRAM_ELEMS/64): the
// first four waves of each compute one 16x16 output tile (the dot products are LDS-bandwidth bound,
// so one tile per CU; the other waves retire at once), results to `caus` [slot][B1][B2].
// gram_finalize then only gathers and adds -- its dependent chain (launch -> Gram gathers -> feature
// loads -> LDS dot products) was as long as the whole reduction.

enum { GRAM_SPLIT_NONE = 0, GRAM_SPLIT_89 = 1 };

__global__ __launch_bounds__(1024) void gram_reduce(const float* __restrict__ gpart, int nchunk, unsigned mask, int split_mode,
                                                    double* __restrict__ gsum, CausalPre cp) {
    __shared__ double part[16][64];
    __shared__ __attribute__((aligned(16))) float csh[CAUSAL_TILE * CAUSAL_PITCH];
    __shared__ __attribute__((aligned(16))) float csm[CAUSAL_TILE * CAUSAL_PITCH];
    if (blockIdx.x >= GRAM_ELEMS / 64) {
        if (threadIdx.x >= 256) return;                      // whole waves; barriers below count the live ones
        causal_pre_tile(cp, blockIdx.x - GRAM_ELEMS / 64, csh, csm, threadIdx.x);
        return;
    }
    const int t = threadIdx.x, el = t & 63, grp = t >> 6;
    const int e = blockIdx.x * 64 + el;
    const int sub = e >> 10;
    const bool need = (mask >> sub) & 1u;
    // sub-tiles whose k-groups are shared by two waves keep the second half in slab 10 / 11
    const int second = (split_mode == GRAM_SPLIT_89 && sub >= 8) ? sub + 2 : -1;
    const bool split = second >= 0;
    const int soff = split ? (second - sub) * 1024 : 0;
    double s = 0.0;
    if (need) {
        const float* p = gpart + e;
        const int64_t cs = (int64_t)GRAM_SLABS * 1024;
        int c = grp;
        // four chunks per trip: eight independent loads in flight, added in chunk order
        for (; c + 48 < nchunk; c += 64) {
            const float a0 = p[c * cs], a1 = p[(c + 16) * cs], a2 = p[(c + 32) * cs], a3 = p[(c + 48) * cs];
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
            if (split) { b0 = p[c * cs + soff]; b1 = p[(c + 16) * cs + soff]; b2 = p[(c + 32) * cs + soff]; b3 = p[(c + 48) * cs + soff]; }
            s += (double)a0; s += (double)b0; s += (double)a1; s += (double)b1;
            s += (double)a2; s += (double)b2; s += (double)a3; s += (double)b3;
        }
        for (; c < nchunk; c += 16) {
            s += (double)p[c * cs];
            if (split) s += (double)p[c * cs + soff];
        }
    }
    part[grp][el] = s;
    __syncthreads();
    if (grp == 0) {
        double tot = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) tot += part[g][el];
        gsum[e] = tot;
    }
}

// The same for the compact records of gram128_partial_x3ws ([nchunk][GRAM_CPT] floats, no split slabs): entry e of every
// chunk, one 256-byte line per wave and chunk.  Thread (e = t & 63, grp = t >> 6) owns chunks grp, grp + 16, ... (at most 16
// of them: nchunk <= 256) and has ALL of its loads in flight before the first add -- one memory round trip instead of four
// (round 3's loop kept four chunks in flight: 5.3 us for 11.8 MB; the work is one pass over 7.9 MB now).  Unconditional loads at
// clamped chunk indices, out-of-range values replaced by zeros: adds in chunk order, run-to-run deterministic as before.
__global__ __launch_bounds__(1024) void gram_reduce_compact(const float* __restrict__ gpart, int nchunk, double* __restrict__ gsum) {
    __shared__ double part[16][64];
    const int t = threadIdx.x, el = t & 63, grp = t >> 6;
    const float* p = gpart + blockIdx.x * 64 + el;
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = grp + 16 * q;
        v[q] = p[(int64_t)(c < nchunk ? c : nchunk - 1) * GRAM_CPT];
    }
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += (grp + 16 * q < nchunk) ? (double)v[q] : 0.0;
    part[grp][el] = s;
    __syncthreads();
    if (grp == 0) {
        double tot = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) tot += part[g][el];
        gsum[blockIdx.x * 64 + el] = tot;
    }
}

struct GramFin {
    const double* gsum;
    int compact;         // gsum holds GRAM_CPT entries in the compact record's order (gram128_partial_x3ws), else slabs
    int mode;
    int B1, B2;          // output rows / cols
    float* out[3];
    const float* h[3];   // causal term per output (rows), or null
    const float* M[3];
    const float* h2;     // bi-causal second term (GRAM_XY / GRAM_SAME only)
    const float* M2;
    const float* caus;   // [3][B1][B2] causal sums written by gram_reduce's extra workgroups: slot p for h[p]; slot 1 for h2
    float sc;
    int T, J;
    int pitch;           // row pitch of the outputs (blocks of a larger matrix), >= B2
    float* out_t;        // GRAM_XY only: blockIdx.z == 1 writes the transposed distances + the (h2, M2) causal term here
};

template <bool COMPACT>
__device__ __forceinline__ double gram_at(const double* __restrict__ gsum, int s, int t) {
    // branch-free (selects): only sub-tiles on/above the diagonal are stored
    const bool sw = (s >> 5) > (t >> 5);
    const int s2 = sw ? t : s, t2 = sw ? s : t;
    if (!COMPACT) return gsum[sub_index(s2 >> 5, t2 >> 5) * 1024 + (s2 & 31) * 32 + (t2 & 31)];
    // compact record: off-diagonal sub-tiles whole, diagonal ones as packed upper triangles
    const int a = s2 >> 5, b = t2 >> 5, r = s2 & 31, c = t2 & 31;
    const int rr = r < c ? r : c, cc = r < c ? c : r;
    const int off = gram_od(a, b) * 1024 + r * 32 + c, dia = 6 * 1024 + a * GRAM_TRI + gram_tri(rr, cc);
    return gsum[a == b ? dia : off];
}

// One 16x16 output tile per block: distances from the summed Gram entries (fp64), scale, causal term.
template <bool COMPACT>
__device__ __forceinline__ void gram_finalize_body(const GramFin& f) {
    const int p = blockIdx.z;
    const int i0 = blockIdx.y * CAUSAL_TILE, j0 = blockIdx.x * CAUSAL_TILE;
    const int i = i0 + (threadIdx.x >> 4), j = j0 + (threadIdx.x & 15);
    const bool ok = i < f.B1 && j < f.B2;
    const double* G = f.gsum;
    // all Gram entries are fetched up front (no control flow between the loads: one memory round trip)
    const int ii = ok ? i : 0, jj = ok ? j : 0;
    const int64_t plane = (int64_t)f.B1 * f.B2, at = (int64_t)ii * f.B2 + jj;
    const float ca = (f.h[f.out_t ? 0 : p]) ? f.caus[(f.out_t ? 0 : p) * plane + at] : 0.f;
    const float cb = (p == 0 && f.h2 && !f.out_t) ? f.caus[plane + at] : 0.f;
    double D;
    if (f.mode == GRAM_LOSS3) {
        const double g_ii = gram_at<COMPACT>(G, ii, ii), g_jj = gram_at<COMPACT>(G, jj, jj), g_ij = gram_at<COMPACT>(G, ii, jj);
        const double e_ii = gram_at<COMPACT>(G, 64 + ii, 64 + ii), e_jj = gram_at<COMPACT>(G, 64 + jj, 64 + jj), e_ij = gram_at<COMPACT>(G, 64 + ii, 64 + jj);
        const double x_ii = gram_at<COMPACT>(G, ii, 64 + ii), x_jj = gram_at<COMPACT>(G, jj, 64 + jj);
        const double x_ij = gram_at<COMPACT>(G, ii, 64 + jj), x_ji = gram_at<COMPACT>(G, jj, 64 + ii);
        const bool diag = ii == jj;
        const double dxx = diag ? 0.0 : g_ii + g_jj - 2.0 * g_ij;
        const double dxy = dxx + e_jj - 2.0 * (x_ij - x_jj);
        const double dee = e_ii + e_jj - 2.0 * e_ij;
        const double dyy = diag ? 0.0 : dxx + dee + 2.0 * (x_ii - x_ij - x_ji + x_jj);
        D = (p == 1) ? dxx : (p == 0 ? dxy : dyy);
    } else if (f.mode == GRAM_XY) {
        D = gram_at<COMPACT>(G, ii, ii) + gram_at<COMPACT>(G, 64 + jj, 64 + jj) - 2.0 * gram_at<COMPACT>(G, ii, 64 + jj);
    } else {  // GRAM_SAME: row i of x is stack row i (rows 64.. come from src2 = x + 64 rows)
        const double g_ii = gram_at<COMPACT>(G, ii, ii), g_jj = gram_at<COMPACT>(G, jj, jj), g_ij = gram_at<COMPACT>(G, ii, jj);
        D = (ii == jj) ? 0.0 : g_ii + g_jj - 2.0 * g_ij;
    }
    if (D < 0.0) D = 0.0;   // a squared distance; rounding of the Gram terms may leave -tiny
    float c = (float)D * f.sc;
    if (f.out_t) {
        // mirror block of an x == y problem (blocked path): blockIdx.z == 0 is the block itself with the (h, M) term,
        // blockIdx.z == 1 its transpose with the (h2, M2) term (rows = this block's columns)
        if (p == 0) { if (f.h[0]) c += ca * f.sc; if (ok) f.out[0][(int64_t)i * f.pitch + j] = c; }
        else { if (f.h2) c += f.caus[plane + (int64_t)jj * f.B2 + ii] * f.sc; if (ok) f.out_t[(int64_t)j * f.pitch + i] = c; }
        return;
    }
    if (f.h[p]) c += ca * f.sc;
    if (p == 0 && f.h2) c += cb * f.sc;
    if (ok) f.out[p][(int64_t)i * f.pitch + j] = c;
}

__global__ __launch_bounds__(256) void gram_finalize(GramFin f) {
    if (f.compact) gram_finalize_body<true>(f); else gram_finalize_body<false>(f);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return ((uintptr_t)p % 16) == 0; }

bool gram_eligible(const CostBatch& cb, int64_t K, bool loss3) {
    if (K % 4 != 0 || K < GRAM_KT) return false;
    for (int p = 0; p < cb.nprob; ++p)
        if (!aligned16(cb.p[p].x) || !aligned16(cb.p[p].y)) return false;
    if (loss3) return cb.nprob == 3 && cb.p[0].Bx <= 64;
    if (cb.nprob != 1) return false;
    if (cb.p[0].same) return cb.p[0].Bx <= 128;
    return cb.p[0].Bx <= 64 && cb.p[0].By <= 64;
}

bool gram_preferred(const CostBatch& cb, int64_t K, bool loss3) {
    // the direct kernel always computes whole 64x64 tiles on the VALU; the MFMA path skips
    // 32-row blocks that hold no valid row, so it is preferred whenever it is eligible
    return gram_eligible(cb, K, loss3) && K >= 256;
}

static int gram_target_wgs() { return 256; }

static bool gram_use_x3() { return !opt(OPT_GRAM_F32); }   // option "gram_f32" = 1: the f32-input MFMA kernel (parity runs)

GramPlan plan_gram(int64_t K) {
    GramPlan pl{};
    // chunks are multiples of the larger stage (64) so that either kernel can run the plan
    const int64_t ksteps = (K + XKT - 1) / XKT;
    int64_t nchunk = gram_target_wgs();
    if (nchunk > ksteps) nchunk = ksteps;
    if (nchunk < 1) nchunk = 1;
    const int64_t spc = (ksteps + nchunk - 1) / nchunk;
    pl.chunk = spc * XKT;
    pl.nchunk = (int)((K + pl.chunk - 1) / pl.chunk);
    // sized for the largest plan the knob can produce so that the query stays an upper bound
    int64_t max_chunks = ksteps < 1024 ? ksteps : 1024;
    if (max_chunks < pl.nchunk) max_chunks = pl.nchunk;
    pl.gpart_bytes = align_up((size_t)max_chunks * GRAM_SLABS * 1024 * sizeof(float), 256);
    pl.gsum_bytes = align_up((size_t)GRAM_ELEMS * sizeof(double), 256);
    pl.caus_bytes = align_up((size_t)3 * GRAM_ROWS * GRAM_ROWS * sizeof(float), 256);
    pl.ws_bytes = pl.gpart_bytes + pl.gsum_bytes + pl.caus_bytes;
    return pl;
}

static unsigned gram_mask(int mode, int n1, int n2) {
    bool valid[4] = {n1 > 0, n1 > 32, n2 > 0, n2 > 32};
    unsigned m = 0;
    for (int a = 0; a < 4; ++a)
        for (int b = a; b < 4; ++b) {
            if (!valid[a] || !valid[b]) continue;
            bool need = true;
            if (mode == GRAM_XY) need = (a == b) || (a < 2 && b >= 2);   // norms + the xy block
            if (need) m |= 1u << sub_index(a, b);
        }
    return m;
}

void gram_sums_span(int B, int64_t K, size_t* off, size_t* n) {
    const GramPlan pl = plan_gram(K);
    *off = pl.gpart_bytes;
    // more than 32 rows per operand: all ten sub-tiles, and on the bf16 pipe the compact record (run_gram)
    *n = (B > 32 && gram_use_x3()) ? GRAM_CPT : GRAM_ELEMS;
}

// stage 0: everything; 1: stop after the fp64 Gram sums and the causal sums are in the workspace; 2: finalize only
int run_gram(const CostBatch& cb, bool loss3, int64_t K, float sc, int T, int J, void* ws,
             size_t ws_bytes, bool partial_only, hipStream_t st, int stage) {
    GramPlan pl = plan_gram(K);
    if (!ws || ws_bytes < pl.ws_bytes)
        return fail(KCCOT_EWORKSPACE, "pairwise_cost(mfma): workspace %zu < required %zu", ws_bytes, pl.ws_bytes);
    if (pl.nchunk > 1024) return fail(KCCOT_EUNSUPPORTED, "pairwise_cost(mfma): %d chunks", pl.nchunk);
    if (pl.nchunk > 256) return fail(KCCOT_EUNSUPPORTED, "pairwise_cost(mfma): %d chunks (the reduction holds 16 x 16)", pl.nchunk);
    GramArgs ga{};
    GramFin gf{};
    const CostProb& p0 = cb.p[0];
    int mode, nout;
    if (loss3) {
        mode = GRAM_LOSS3; nout = 3;
        ga.src1 = p0.x; ga.src2 = p0.y; ga.n1 = p0.Bx; ga.n2 = p0.By; ga.pair_diff = 1;
        gf.B1 = p0.Bx; gf.B2 = p0.By;
        for (int p = 0; p < 3; ++p) { gf.out[p] = cb.p[p].out; gf.h[p] = cb.p[p].h1; gf.M[p] = cb.p[p].M1; }
    } else if (p0.same) {
        mode = GRAM_SAME; nout = 1;
        ga.src1 = p0.x; ga.n1 = p0.Bx < 64 ? p0.Bx : 64;
        ga.n2 = p0.Bx > 64 ? p0.Bx - 64 : 0;
        ga.src2 = ga.n2 ? p0.x + (int64_t)64 * K : p0.x;
        ga.pair_diff = 0;
        gf.B1 = gf.B2 = p0.Bx;
        gf.out[0] = p0.out; gf.h[0] = p0.h1; gf.M[0] = p0.M1; gf.h2 = p0.h2; gf.M2 = p0.M2;
    } else {
        mode = GRAM_XY; nout = p0.out_t ? 2 : 1;
        ga.src1 = p0.x; ga.src2 = p0.y; ga.n1 = p0.Bx; ga.n2 = p0.By; ga.pair_diff = 0;
        gf.B1 = p0.Bx; gf.B2 = p0.By;
        gf.out[0] = p0.out; gf.h[0] = p0.h1; gf.M[0] = p0.M1; gf.h2 = p0.h2; gf.M2 = p0.M2;
        gf.out_t = p0.out_t;
    }
    gf.pitch = p0.out_pitch > 0 ? p0.out_pitch : gf.B2;
    ga.mask = gram_mask(mode, ga.n1, ga.n2);
    // full 64 + 64 stacks: the bf16 kernel computes all ten sub-tiles (two more than a plain x-y problem reads)
    if (mode == GRAM_XY && ga.n1 == 64 && ga.n2 == 64 && gram_use_x3()) ga.mask = 0x3FFu;
    ga.K = K; ga.chunk = pl.chunk;
    ga.gpart = static_cast<float*>(ws);
    double* gsum = reinterpret_cast<double*>(static_cast<char*>(ws) + pl.gpart_bytes);
    // causal sums: slots and tiles
    CausalPre cp{};
    cp.caus = reinterpret_cast<float*>(static_cast<char*>(ws) + pl.gpart_bytes + pl.gsum_bytes);
    cp.B1 = gf.B1; cp.B2 = gf.B2; cp.T = T; cp.J = J;
    cp.nti = (gf.B1 + CAUSAL_TILE - 1) / CAUSAL_TILE; cp.ntj = (gf.B2 + CAUSAL_TILE - 1) / CAUSAL_TILE;
    int nslot = 0;
    if (loss3) {
        for (int p = 0; p < 3; ++p) { cp.h[p] = gf.h[p]; cp.M[p] = gf.M[p]; if (gf.h[p]) nslot = p + 1; }
    } else {
        cp.h[0] = gf.h[0]; cp.M[0] = gf.M[0]; cp.h[1] = gf.h2; cp.M[1] = gf.M2;
        nslot = gf.h2 ? 2 : (gf.h[0] ? 1 : 0);
    }
    int ncausal = nslot * cp.nti * cp.ntj;                  // tiles still to be done by gram_reduce's extra workgroups
    ga.nchunk = pl.nchunk; ga.ntiles = 0; ga.cp = cp;
    int split_mode = (ga.mask == 0x3FFu) ? GRAM_SPLIT_89 : GRAM_SPLIT_NONE;
    const bool compact = ga.mask == 0x3FFu && gram_use_x3();   // gram128_partial_x3ws writes compact records
    gf.compact = compact ? 1 : 0;
    if (stage == 2) {
        gf.gsum = gsum; gf.caus = cp.caus; gf.mode = mode; gf.sc = sc; gf.T = T; gf.J = J;
        hipLaunchKernelGGL(gram_finalize, dim3((gf.B2 + CAUSAL_TILE - 1) / CAUSAL_TILE, (gf.B1 + CAUSAL_TILE - 1) / CAUSAL_TILE, nout),
                           dim3(256), 0, st, gf);
        return launch_status("gram_finalize");
    }
    if (compact) {
        // 16 spare workgroups (one per CU the 240-way K-split leaves idle) take the causal tiles
        int spare = 0;
        if (ncausal > 0 && !partial_only) { spare = ncausal < 16 ? ncausal : 16; ga.ntiles = ncausal; ncausal = 0; }
        // (producers with two stages of unconditional clamped loads in flight: 19.1-20.4 us against 21.0-21.8 us for one
        // stage of predicated loads, profiles/r02x_ab_gram_producers.txt)
        hipLaunchKernelGGL(gram128_partial_x3ws, dim3(pl.nchunk + spare), dim3(512), 0, st, ga);
    }
    else if (ga.mask == 0x3FFu) hipLaunchKernelGGL(gram128_partial<true>, dim3(pl.nchunk), dim3(256), 0, st, ga);
    else hipLaunchKernelGGL(gram128_partial<false>, dim3(pl.nchunk), dim3(256), 0, st, ga);
    int rc = launch_status("gram128_partial");
    if (rc || partial_only) return rc;
    if (compact)
        // (its causal tiles ran in the spare workgroups of the Gram launch: ncausal == 0 here)
        hipLaunchKernelGGL(gram_reduce_compact, dim3(GRAM_CPT / 64), dim3(1024), 0, st, (const float*)ga.gpart, pl.nchunk, gsum);
    else
        hipLaunchKernelGGL(gram_reduce, dim3(GRAM_ELEMS / 64 + ncausal), dim3(1024), 0, st,
                           (const float*)ga.gpart, pl.nchunk, ga.mask, split_mode, gsum, cp);
    rc = launch_status("gram_reduce");
    if (rc || stage == 1) return rc;
    gf.gsum = gsum; gf.caus = cp.caus; gf.mode = mode; gf.sc = sc; gf.T = T; gf.J = J;
    hipLaunchKernelGGL(gram_finalize, dim3((gf.B2 + CAUSAL_TILE - 1) / CAUSAL_TILE, (gf.B1 + CAUSAL_TILE - 1) / CAUSAL_TILE, nout),
                       dim3(256), 0, st, gf);
    return launch_status("gram_finalize");
}

}  // namespace kccot

namespace kccot {

// ---- batches above 64: the loss's three matrices in 64 x 64 blocks -----------------------------------------
// Diagonal blocks (I,I) carry the entries the pair-difference form exists for (sample i against its own fake):
// they run the GRAM_LOSS3 kernel on (real_I, fake_I).  Off-diagonal blocks compare different samples -- distances
// of the size of the operands' norms, where the plain Gram form x.x + y.y - 2 x.y (exact bf16 split, fp64
// combination) is accurate -- and run the GRAM_XY kernel on a [rows_I ; cols_J] stack: xy needs both (I,J) and
// (J,I); xx and yy are symmetric in their distances, so one launch per unordered pair writes the block and, with
// the mirror block's own causal term, its transpose.  nb + 2 nb (nb - 1) launches of three kernels each.
bool gram_blocked_eligible(const CostBatch& cb, int64_t K, bool loss3) {
    if (!loss3 || cb.nprob != 3 || !gram_use_x3() || !opt(OPT_COST_BLOCKED)) return false;
    const int B = cb.p[0].Bx;
    if (B <= 64 || B % 64 != 0 || cb.p[0].By != B || K % 4 != 0 || K < 256) return false;
    for (int p = 0; p < 3; ++p)
        if ((uintptr_t)cb.p[p].x % 16 || (uintptr_t)cb.p[p].y % 16) return false;
    return true;
}

int run_gram_blocked(const CostBatch& cb, int64_t K, float sc, int T, int J, void* ws, size_t ws_bytes, hipStream_t st) {
    const int B = cb.p[0].Bx, nb = B / 64;
    const float* real = cb.p[0].x;
    const float* fake = cb.p[0].y;
    const int64_t tj = (int64_t)T * J;
    auto rows = [&](const float* v, int blk) { return v + (int64_t)blk * 64 * K; };
    auto feat = [&](const float* f, int blk) { return f ? f + (int64_t)blk * 64 * tj : nullptr; };
    auto block = [&](float* out, int I, int Jb) { return out + (int64_t)I * 64 * B + (int64_t)Jb * 64; };
    int rc;
    for (int I = 0; I < nb; ++I) {
        // diagonal block: the pair-difference kernel, three outputs
        CostBatch d{};
        d.nprob = 3;
        for (int p = 0; p < 3; ++p) {
            d.p[p] = cb.p[p];
            d.p[p].x = rows(cb.p[p].x, I); d.p[p].y = rows(cb.p[p].y, I);
            d.p[p].Bx = d.p[p].By = 64;
            d.p[p].h1 = feat(cb.p[p].h1, I); d.p[p].M1 = feat(cb.p[p].M1, I);
            d.p[p].out = block(cb.p[p].out, I, I); d.p[p].out_pitch = B;
        }
        if ((rc = run_gram(d, true, K, sc, T, J, ws, ws_bytes, false, st))) return rc;
        for (int Jb = 0; Jb < nb; ++Jb) {
            if (Jb == I) continue;
            // xy block (I,J): real rows I against fake rows J, features of problem 0
            CostBatch o{};
            o.nprob = 1;
            o.p[0] = CostProb{rows(real, I), rows(fake, Jb), 64, 64, 0, feat(cb.p[0].h1, I), feat(cb.p[0].M1, Jb), nullptr, nullptr,
                              block(cb.p[0].out, I, Jb), nullptr, 0, 0, 0, B, nullptr};
            if ((rc = run_gram(o, false, K, sc, T, J, ws, ws_bytes, false, st))) return rc;
            if (Jb < I) continue;
            // xx and yy blocks (I,J) and their mirrors (J,I): one launch per unordered pair
            for (int p = 1; p <= 2; ++p) {
                const float* v = p == 1 ? real : fake;
                CostBatch m{};
                m.nprob = 1;
                m.p[0] = CostProb{rows(v, I), rows(v, Jb), 64, 64, 0, feat(cb.p[p].h1, I), feat(cb.p[p].M1, Jb),
                                  feat(cb.p[p].h1, Jb), feat(cb.p[p].M1, I), block(cb.p[p].out, I, Jb), nullptr, 0, 0, 0, B,
                                  block(cb.p[p].out, Jb, I)};
                if ((rc = run_gram(m, false, K, sc, T, J, ws, ws_bytes, false, st))) return rc;
            }
        }
    }
    return 0;
}

}  // namespace kccot
