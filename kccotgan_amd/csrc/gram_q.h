// Building blocks shared by the "one wave per SIMD, every wave stages and consumes" Gram kernels (cost_tile256.hip: 256 x 256
// pair tiles of the single-GPU loss; cost_rows.hip: the row block of a batch-sharded rank): the exact three-way bf16 split of
// an element pair into an LDS stage of 16 columns, fragment reads and the six-product MFMA chain.
//
// LDS stage = 3 planes (h, m, l) x NROWS rows x 32 bytes (16 bf16: the pairs (k, k+1) of eight float4 column pieces of a
// 32-column load granule; the odd step of the granule holds the pairs (k+2, k+3)).  A fragment is rows 32 r .. 32 r + 31,
// lane l reading 16 bytes at row (l & 31), column half (l >> 5): contiguous 1 KB per plane, conflict-free.
#pragma once
#include "common.h"

namespace kccot {

typedef __bf16 qbf16x8 __attribute__((ext_vector_type(8)));
typedef float qf32x16 __attribute__((ext_vector_type(16)));

constexpr int QROWB = 32;                  // bytes of one row of one plane of a 16-k step
constexpr int QG = 32;                     // columns per load granule = two steps
// Granules per K-chunk.  The chunk length is bounded by the ACCUMULATION, not by occupancy: an MFMA accumulation loses
// ~0.02 ulp of the running sum (addends are truncated when aligned to it; measured: 2.3e-5 low after 14 000 accumulations of
// all-positive terms in the 128-tile kernel, 1.5e-5 of a distance with 2300), so one partial tile holds at most
// Q_MAX_GRAN granules = 1536 columns = 576 accumulations (a distance within 4e-6); the fp64 reduction adds the tiles up.
constexpr int Q_MAX_GRAN = 48;

// split two floats exactly into three bf16 pieces each; dword = bf16(a) | bf16(b) << 16 per plane
template <int PLANE>
__device__ __forceinline__ void gq_split_store(unsigned char* zs, int off, float a, float b) {
    const unsigned xa = __float_as_uint(a), xb = __float_as_uint(b);
    const float ra = a - __uint_as_float(xa & 0xFFFF0000u), rb = b - __uint_as_float(xb & 0xFFFF0000u);      // exact
    const unsigned ma = __float_as_uint(ra), mb = __float_as_uint(rb);
    const float la = ra - __uint_as_float(ma & 0xFFFF0000u), lb = rb - __uint_as_float(mb & 0xFFFF0000u);      // exact, <= 8 bits
    *reinterpret_cast<unsigned*>(zs + off) = __builtin_amdgcn_perm(xb, xa, 0x07060302u);
    *reinterpret_cast<unsigned*>(zs + PLANE + off) = __builtin_amdgcn_perm(mb, ma, 0x07060302u);
    *reinterpret_cast<unsigned*>(zs + 2 * PLANE + off) = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

struct QFrag { qbf16x8 h, m, l; };
template <int PLANE>
__device__ __forceinline__ QFrag gq_frag(const unsigned char* zs, int off) {
    QFrag f;
    f.h = *reinterpret_cast<const qbf16x8*>(zs + off);
    f.m = *reinterpret_cast<const qbf16x8*>(zs + PLANE + off);
    f.l = *reinterpret_cast<const qbf16x8*>(zs + 2 * PLANE + off);
    return f;
}
// One 32 x 32 accumulator tile.  (KCCOT_ABLATE_MFMA16: the TIMING-ONLY build of tools/micro/q256_mfma16_ablate.sh -- results are
// garbage -- keeps it as four 16 x 16 quarter tiles and issues twelve v_mfma_f32_16x16x32_bf16 per fragment pair: the same FLOPs,
// operand reads and registers; does the shape's higher sustained clock (tools/micro/mfma_shape.hip) survive next to the
// kernels' LDS and VALU traffic?)
#ifdef KCCOT_ABLATE_MFMA16
typedef float qf32x4 __attribute__((ext_vector_type(4)));
struct QAcc { qf32x4 q[4]; };
#define QACC(a, r) (a).q[(r) >> 2][(r) & 3]
__device__ __forceinline__ void gq_mfma6(QAcc& acc, const QFrag& a, const QFrag& b) {
    acc.q[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, acc.q[0], 0, 0, 0);
    acc.q[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, acc.q[1], 0, 0, 0);
    acc.q[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, acc.q[2], 0, 0, 0);
    acc.q[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, acc.q[3], 0, 0, 0);
    acc.q[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, acc.q[0], 0, 0, 0);
    acc.q[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, acc.q[1], 0, 0, 0);
    acc.q[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, acc.q[2], 0, 0, 0);
    acc.q[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, acc.q[3], 0, 0, 0);
    acc.q[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, acc.q[0], 0, 0, 0);
    acc.q[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, acc.q[1], 0, 0, 0);
    acc.q[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, acc.q[2], 0, 0, 0);
    acc.q[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, acc.q[3], 0, 0, 0);
}
#else
typedef qf32x16 QAcc;
#define QACC(a, r) (a)[r]
__device__ __forceinline__ void gq_mfma6(QAcc& acc, const QFrag& a, const QFrag& b) {   // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
}
#endif

typedef unsigned int qu32x4 __attribute__((ext_vector_type(4)));

}  // namespace kccot
