// Video gradient of large batches on 256 x 256 output tiles (round 4).
//
// dfake[m, c] = sum_r W[m, r] Z[r, c]   (m: sample = output row, r: stack row of [X ; Y], c: video column; gan_utils.py:14-17
// differentiated, kernel_train.py:289 is the caller).  apply_coeffs_x3_m256n128 (cost_bwd.hip) gives a consumer wave 64 rows x
// 128 columns and streams its W fragments from L2 into registers: 24 KB of W per 16-row k-step and CU for 192 MFMAs -- the
// stream its ablation priced at ~3 of 12.2 ms at BASELINE configs[4] (profiles/r3_apply_ablation.txt).  Here the tile is the
// one of the 256-row Gram kernels (cost_tile256.hip, gram_q.h): a workgroup of four waves, ONE per SIMD, owns 256 output rows
// x 256 columns; every wave stages AND consumes; wave (wr, wc) accumulates 128 x 128 = 4 x 4 MFMA tiles (256 accumulator
// registers).  Per 16-k step the workgroup brings
//   * the W panel: 8 row tiles x 3 planes = 24 fragment-major KiB of the pre-split coefficients (retile_coeffs) from L2 into
//     the LDS stage -- a plain 16-byte copy per lane, no arithmetic; 24 KB per step and CU for 384 MFMAs: HALF the W bytes per
//     MFMA, and they reach the matrix pipe through LDS fragment reads like every other operand;
//   * the Z panel: 16 stack rows x 256 columns of real / fake from HBM; a thread holds 4 rows x 4 columns, i.e. for each of
//     its columns 4 consecutive k: split exactly into the three bf16 planes in registers and written as 8-byte pieces at
//     [column][k] -- the transposition costs nothing because the k values a lane owns are adjacent.
// LDS bandwidth is the co-limit of this tile (36 KB of fragment reads + 12 KB of stage writes per wave and step against 96
// MFMAs), so every access is conflict-free BY CONSTRUCTION: the lane's k-group is its low two bits (the four lanes of a
// column group fill one 32-byte row), column 4 g + c of the tile is stored at LDS row 64 c + g (sixteen lanes = 128
// contiguous bytes), and the MFMA lane <-> column map follows from that: column tile j = wc + 2 u (u = 0..3) of wave (wr, wc)
// holds columns 128 wc + 4 lane + u -- a lane's four column tiles are four ADJACENT columns, stored as one float4.  The W
// fragments are copied with the lane order of the LDS rows.  (First build: column group = lane, k-group = wave: every
// 8-byte stage write hit the same bank pair -- 64 cycles per write instruction, 18.1 ms against 13.2 for the old kernel.)
// The stage layout, fragment reads and the six-product chain are gram_q.h's (3 planes x 512 rows x 32 bytes: W rows 0..255,
// Z columns as rows 256..511); two stages are resident (96 KB).  Same products in the same order per output as the other
// apply kernels (k ascending, mm, hl, lh, hm, mh, hh), fp32 accumulation over the 2B stack rows.
#include "common.h"
#include "cost_internal.h"
#include "gram_q.h"

namespace kccot {

constexpr int AQ_P = 256;                         // rows of a panel (output rows / columns of the tile)
constexpr int AQ_PLANE = 2 * AQ_P * QROWB;        // 16384 bytes: W rows 0..255, Z columns 256..511
constexpr int AQ_SLOT = 3 * AQ_PLANE;             // 49152 bytes per 16-k stage

struct ApplyQ {
    const unsigned short* Wt3;   // fragment-major planes (retile_coeffs), positioned at output row tile 0 of this launch
    int Bt, Rt;                  // rows / stack rows of the whole coefficient matrix (plane geometry)
    const float* src1; int n1;   // stack rows [0, n1): rows of src1; [n1, n1 + n2): rows of src2   (n1 % 16 == 0)
    const float* src2; int n2;
    int64_t K;
    int64_t ncol;                // column tiles of 256 (the last may be partial)
    int nrow;                    // row tiles of 256
    float* out;                  // [nrow * 256][K]
};

template <bool RAGGED>
__global__ __launch_bounds__(256) void apply_q256(ApplyQ a) {
    __shared__ __attribute__((aligned(16))) unsigned char zs[2 * AQ_SLOT];
    unsigned char* const zs0 = zs;
    unsigned char* const zs1 = zs + AQ_SLOT;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t K = a.K;
    const int nsteps = (a.n1 + a.n2) >> 4;
    const int64_t wplane = (int64_t)(a.Bt / 32) * (a.Rt / 16) * 512;          // shorts per plane

    // ---- staging role
    // W: fragment (plane pl, row tile mt of the workgroup's 8, this k-step) = 1 KiB, lane-major (lane L: row L & 31, k-half
    // L >> 5); wave w copies fragments 6 w .. 6 w + 5 of the step's 24 (f = 3 mt + pl).  Thread l takes the 16 bytes that
    // belong at LDS offset 16 l (row l >> 1, k-half l & 1), i.e. source lane (l >> 1) + 32 (l & 1): contiguous LDS writes
    const int wsrc = ((lane >> 1) + 32 * (lane & 1)) * 8;                    // shorts
    const int wlds = 16 * lane;
    // Z: thread = (stack rows 4 rg .. 4 rg + 3 of the step, columns 4 cg .. 4 cg + 3), rg = its low two lane bits
    const int rg = lane & 3, cg = 16 * wave + (lane >> 2);
    const int zlds = AQ_P * QROWB + cg * QROWB + 8 * rg;                    // + 64 c rows per column c of the group, + plane

    // ---- consuming role: wave (wr, wc) owns rows 128 wr .. x columns 128 wc .. of the tile
    const int wr = wave >> 1, wc = wave & 1;
    const int lo = (lane & 31) * QROWB + 16 * (lane >> 5);
    const int aoff = 4 * wr * 32 * QROWB + lo;
    const int boff = AQ_P * QROWB + wc * 32 * QROWB + lo;                    // column tile u of the wave: + 2 u tiles

    // XCD-aware tile order: the row tiles of one column tile run side by side on ONE XCD (blocks b and b + 8 share an XCD), so
    // that the second reader of a Z panel finds it in that XCD's L2
    const int64_t ntiles = a.ncol * a.nrow;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;       // gridDim.x % 8 == 0 (host)
    for (int64_t it = 0;; ++it) {
        // tiles of this XCD in round `it`: per_xcd consecutive (column-major: row tile fastest)
        const int64_t lin = (it * 8 + xcd) * per_xcd + slot;
        if (lin >= ntiles) break;
        const int64_t ctile = lin / a.nrow;
        const int rtile = (int)(lin % a.nrow);
        const int64_t col0 = ctile * AQ_P;

        // TWO register sets: the loads of a k-step are issued two steps (~3 us of MFMAs) before they are split -- one step of
        // lead did not cover the HBM latency under load (15.6 ms against 13.2 for the kernel this one replaces)
        struct StageRegs { uint4 Wg[6]; float4 Zg[4]; };
        StageRegs ra, rb;
        // Loads go through buffer descriptors (raw_buffer_load), not plain pointer loads: hipcc treated the plain loads of the
        // read-only coefficient planes as rematerialisable and RE-ISSUED them right in front of their LDS write -- load,
        // s_waitcnt vmcnt(0), ds_write, four times per step, each wait also draining the step's HBM loads (15.7 ms).
        const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.Wt3), 0, 0xFFFFFFFFu, 0x00020000);
        const unsigned wvo = (unsigned)((((int64_t)(rtile * 8) * (a.Rt / 16)) * 512 + wsrc) * 2);         // bytes; planes < 4 GiB (host)
        auto issue = [&](StageRegs& R, int s) {   // loads of k-step s (clamped to the last one: past the end they are staged but never read)
            const int sc = s < nsteps ? s : nsteps - 1;
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                const int ff = 6 * wave + f, mt = ff / 3, pl = ff % 3;
                const unsigned so = (unsigned)((pl * wplane + ((int64_t)mt * (a.Rt / 16) + sc) * 512) * 2);
                R.Wg[f] = __builtin_bit_cast(uint4, (qu32x4)__builtin_amdgcn_raw_buffer_load_b128(rw, (int)wvo, (int)so, 0));
            }
            const int r0 = 16 * sc;                       // the step's 16 stack rows lie in one source (n1 % 16 == 0)
            const float* base = r0 < a.n1 ? a.src1 + (int64_t)r0 * K : a.src2 + (int64_t)(r0 - a.n1) * K;
            const auto rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0xFFFFFFFFu, 0x00020000);
            int64_t c = col0 + 4 * cg;
            if (RAGGED) c = c + 4 <= K ? c : K - 4;
            const unsigned zvo = (unsigned)(((int64_t)(4 * rg) * K + c) * 4);                              // 15 rows < 4 GiB: K <= 2^26 (host)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                R.Zg[j] = __builtin_bit_cast(float4, (qu32x4)__builtin_amdgcn_raw_buffer_load_b128(rz, (int)zvo, (int)((unsigned)(j * K * 4)), 0));
        };
        auto emit = [&](const StageRegs& R, unsigned char* zst) {
            const uint4 (&Wg)[6] = R.Wg;
            const float4 (&Zg)[4] = R.Zg;
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                const int ff = 6 * wave + f, mt = ff / 3, pl = ff % 3;
                *reinterpret_cast<uint4*>(zst + pl * AQ_PLANE + mt * 32 * QROWB + wlds) = Wg[f];
            }
            const bool cok = !RAGGED || col0 + 4 * cg + 4 <= K;
            const float zc[4][4] = {{Zg[0].x, Zg[1].x, Zg[2].x, Zg[3].x}, {Zg[0].y, Zg[1].y, Zg[2].y, Zg[3].y},
                                    {Zg[0].z, Zg[1].z, Zg[2].z, Zg[3].z}, {Zg[0].w, Zg[1].w, Zg[2].w, Zg[3].w}};
#pragma unroll
            for (int c = 0; c < 4; ++c) {    // column 4 cg + c: k = 4 rg .. 4 rg + 3 of the step, three planes of 8 bytes
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = cok ? zc[c][j] : 0.f;
                    const unsigned xb = __float_as_uint(x);
                    const float r1 = x - __uint_as_float(xb & 0xFFFF0000u);            // exact
                    const unsigned mb = __float_as_uint(r1);
                    const float r2 = r1 - __uint_as_float(mb & 0xFFFF0000u);           // exact, <= 8 bits
                    h[j] = xb; m[j] = mb; l[j] = __float_as_uint(r2);
                }
                const uint2 ph = {__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
                const uint2 pm = {__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
                const uint2 pl = {__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
                *reinterpret_cast<uint2*>(zst + zlds + 64 * c * QROWB) = ph;
                *reinterpret_cast<uint2*>(zst + AQ_PLANE + zlds + 64 * c * QROWB) = pm;
                *reinterpret_cast<uint2*>(zst + 2 * AQ_PLANE + zlds + 64 * c * QROWB) = pl;
            }
        };

        QAcc acc[16];
#pragma unroll
        for (int t2 = 0; t2 < 16; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) QACC(acc[t2], r) = 0.f;
        auto step = [&](const unsigned char* zst) {
            // all four column tiles held (48 fragment registers), every W fragment read ONCE: 24 fragment reads per step
            QFrag bf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bf[u] = gq_frag<AQ_PLANE>(zst, boff + 2 * u * 32 * QROWB);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const QFrag af = gq_frag<AQ_PLANE>(zst, aoff + i * 32 * QROWB);
#pragma unroll
                for (int u = 0; u < 4; ++u) gq_mfma6(acc[4 * i + u], af, bf[u]);
            }
        };
        // per MFMA at most one LDS read, two VALU instructions of the split, one LDS write; the next step's ten loads spread
        // over the step (cost_tile256.hip has the measurements behind this mix)
        auto interleave = [&]() {
            __builtin_amdgcn_sched_group_barrier(0x100, 15, 0);
#pragma unroll
            for (int m = 0; m < 96; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (m % 9 == 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        };

        issue(ra, 0);
        issue(rb, 1);
        emit(ra, zs0);
        issue(ra, 2);
        lds_barrier();
        for (int s = 0; s < nsteps; s += 2) {            // nsteps is even (host: (n1 + n2) % 32 == 0)
            step(zs0);
            emit(rb, zs1);                               // k-step s + 1
            issue(rb, s + 3);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();
            __builtin_amdgcn_sched_barrier(0);
            step(zs1);
            emit(ra, zs0);                               // k-step s + 2 (past the end: staged, never read)
            issue(ra, s + 4);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }

        // accumulator register r of lane l is element ((r & 3) + 8 (r >> 2) + 4 (l >> 5), column lane l & 31) of its tile; the
        // lane's four column tiles are the columns 128 wc + 4 (l & 31) + u, u = 0..3: one 16-byte store per (row tile, r)
        const int64_t cbase = col0 + 128 * wc + 4 * (lane & 31);
        float* o = a.out + ((int64_t)rtile * AQ_P + 128 * wr + 4 * (lane >> 5)) * K + cbase;
        if (!RAGGED || cbase + 4 <= K) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float4 v = {QACC(acc[4 * i], r), QACC(acc[4 * i + 1], r), QACC(acc[4 * i + 2], r), QACC(acc[4 * i + 3], r)};
                    *reinterpret_cast<float4*>(o + (int64_t)(32 * i + (r & 3) + 8 * (r >> 2)) * K) = v;
                }
        }
    }
}

// Wt3use: the fragment-major planes positioned at the first wanted output row tile (a multiple of 256 rows)
bool apply_q256_applies(int Bout, int n1, int n2, int64_t K) {
    return Bout % AQ_P == 0 && n1 % 16 == 0 && (n1 + n2) % 32 == 0 && K % 4 == 0 && K >= AQ_P && K <= ((int64_t)1 << 26);
}

int launch_apply_q256(const unsigned short* Wt3use, int Bt, int Rt, const float* s1, int n1, const float* s2, int n2, int Bout,
                      int64_t K, float* out, hipStream_t st) {
    ApplyQ a{};
    a.Wt3 = Wt3use; a.Bt = Bt; a.Rt = Rt; a.src1 = s1; a.n1 = n1; a.src2 = s2; a.n2 = n2; a.K = K;
    a.ncol = (K + AQ_P - 1) / AQ_P; a.nrow = Bout / AQ_P; a.out = out;
    const int64_t ntiles = a.ncol * a.nrow;
    int grid = ntiles < 256 ? (int)((ntiles + 7) / 8 * 8) : 256;          // one workgroup per CU (96 KB of LDS), a multiple of 8
    if (K % AQ_P == 0) hipLaunchKernelGGL(apply_q256<false>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(apply_q256<true>, dim3(grid), dim3(256), 0, st, a);
    return launch_status("apply_q256");
}

}  // namespace kccot
