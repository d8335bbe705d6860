// k_gapless.hpp -- exhaustive gapless (ungapped-diagonal) scan of every target, gfx950.
//
// Semantics: SmithWaterman::ungapped_alignment (reference M/src/alignment/StripedSmithWaterman.cpp:1817-1876):
//   S(q,i) = sat_u8(S(q-1,i-1) + prof) -sat bias ;  score = max S.   Because the profile carries +bias and
//   the only clamp that can bind before the maximum is taken is the upper one, this equals
//   min(255 - bias, best local run sum along any diagonal)   (proof sketch in DESIGN.md; pinned by tests).
//
// Mapping (not the CPU's striping, not a CUDA warp tiling):
//   * 8 lanes x 2 strips x R registers hold one target's DP column; 8 targets per wave64.
//   * scores live in packed FP16 pairs scaled by 2^-11 (score s is stored as s/2048): every integer 0..2048 and every
//     profile entry is exact in FP16, "v_pk_add_f16 clamp" performs add + floor at zero (+ cap at 2048, far above the
//     255 - bias the result is cut to) in ONE instruction, and gfx950's three-operand v_pk_maximum3_f16 folds TWO score
//     registers into the running maximum per instruction -> 3 VALU ops per 4 cells (the int16 formulation needs 4:
//     there is no packed integer max3).  Exactness: a diagonal that ever exceeds 2048 already pins its target at the
//     cap, below that all sums are exact, so the result equals the integer recurrence bit for bit (tests).
//   * the 22 x (16R) FP16 profile sits in LDS in a 2-copy, bank-row aligned image: all ds_read_b128 are
//     conflict free (fs_kernels.h).  The image is chunk-major with a 256-byte row stride, so the row address of a
//     target residue is (code << 8) | laneOffset: ONE v_perm_b32 straight from the packed residue word (the
//     4-register chunks sit at immediate offsets k * 22 * 256).  LDS bytes/cell = 2.
//   * per target column a lane issues 1.5 R DP ops + 3 others (row address, dpp, hand-off perm): measured
//     tools/ubench/gapless_ablate.hip, profiles/r01_q_gapless_ablation_ubench.txt.
//   * the target DB is pre-tiled in HBM as 8-target stripes (targets grouped by length) interleaved at 16-byte granularity: one wave-level
//     global load = one 128-byte line, every byte of the DB is read exactly once per query.
//   * diagonal hand-off between lanes: v_mov_b32_dpp row_shr:1 + v_perm_b32 (no LDS round trip).
//   * waves pull work items from an atomic queue ordered by descending length (LPT).  An item is a stripe or, for
//     stripes longer than the per-wave share of the launch, a column segment of it that starts ceil(Lq/16) chunks
//     early: after that many warm-up columns every diagonal that reaches the segment's own columns has been followed from
//     its row 0, warm-up cells only ever hold values <= the true ones, so max over segments == the stripe's maximum
//     (combined with an atomic max on the score bytes).  Without it the longest stripe alone outlasts the average wave.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "fs_kernels.h"

namespace fs {

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_adds_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}

// FP16-domain helpers (scores scaled by 2^-11)
__device__ __forceinline__ uint32_t pk_addc_f16(uint32_t a, uint32_t b) {      // clamp(a + b, 0, 1) per half
    uint32_t d;
    asm("v_pk_add_f16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_max3_f16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// bits of the FP16 value i / 2048 for a small integer i (|i| <= 2048): exact
__device__ __forceinline__ uint32_t f16ScaledBits(int i) {
    return (uint32_t) __half_as_ushort(__float2half_rn((float) i * (1.0f / 2048.0f)));
}
constexpr uint32_t kDead2 = 0xBC00BC00u;     // packed (-1.0, -1.0): a dead profile row forces the cell to zero

// One query of a launch.  A launch covers nQueries queries of the SAME register class: workgroups
// [q * blocksPerQuery, (q + 1) * blocksPerQuery) serve query q (own profile image, own work queue, own score array).  The
// hardware hands out workgroups in index order, so query q + 1's workgroups move in as query q's drain: the tail of one
// scan is filled by the head of the next inside ONE launch -- what several host threads with a stream each only
// approximate -- and a launch's duration is an honest per-launch figure.
struct GaplessQuery {
    const int8_t *pssm;         // [21][L] query profile (device copy)
    uint8_t *scores;            // [nTargets]
    uint32_t *queue;            // work counter, zeroed before launch
    int L;
    int cap;                    // min(cap, score)
    // PAIRED instantiations: a second query of the same 16-row class in the upper half of the rows (lanes 4..7 of every target group)
    const int8_t *pssmB;
    uint8_t *scoresB;
    int LB;
    int capB;
};

struct GaplessArgs {
    const GaplessQuery *queries; // [gridDim.x / blocksPerQuery]; nullptr: the single query described by pssm / L / cap / scores / queue below
    uint32_t blocksPerQuery;
    const uint4 *scan;          // stripe-interleaved target residues (codes 0..20, 21 = past end)
    const uint64_t *stripeOff;  // [nStripes] offset in uint4 units
    const uint32_t *stripeLen;  // [nStripes] length in 16-column chunks
    const uint32_t *stripeTargets; // [nStripes][8] target id per stripe slot, 0xffffffff = empty slot
    const uint4 *items;         // [nItems] work items, longest first: {stripe, split << 31 | firstChunk << 16 | endChunk, stripe offset lo, hi}
    uint32_t nItems;
    uint32_t nTargets;
    const int8_t *pssm;         // [21][L] query profile (device copy)
    int L;
    int cap;                    // min(cap, score)
    uint8_t *scores;            // [nTargets]
    uint32_t *queue;            // work counter, zeroed before launch
    // query row tiles (queries longer than 16*R rows): tile t covers rows [tileBase, tileBase + 16R)
    int tileBase;
    int firstTile, lastTile;
    const uint16_t *borderIn;   // S(last row of the previous tile, column) for every target column, scan-layout order
    uint16_t *borderOut;
    int16_t *scoreAcc;          // running maximum over tiles (biased domain), [nTargets]
};

// PAIRED: two SHORT queries of one 16-row class share the kernel's rows -- query A in lanes 0..3 of every target group (rows
// 0 .. 8R-1), query B in lanes 4..7 -- so the 3 per-column instructions that do not depend on R are paid once for both: a query of
// 128 residues runs with R = 16 registers per lane instead of 8, 11 % instead of 20 % of the column's instructions are overhead.
// Lane 4 starts B's diagonals at zero exactly like lane 0 starts A's; the final maximum is taken over 4 lanes per query.
template <int R, bool TILED, bool PAIRED = false>
__global__ __launch_bounds__(gaplessBlockThreads(R)) void k_gapless(GaplessArgs a) {
    static_assert(R >= 1 && R <= (TILED ? kGaplessMaxR : kGaplessMaxRUntiled), "register count out of range");
    static_assert(!PAIRED || (!TILED && R % 2 == 0), "paired queries: untiled, even register count");
    constexpr int BLOCK = gaplessBlockThreads(R);
    constexpr int CHB = gaplessChunkBytes();
    constexpr int NCH = gaplessChunks(R);         // ds_read_b128 per column; the last one may carry unused registers
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the kernel's only LDS: starts at LDS address 0
    uint32_t blockInQuery = blockIdx.x, blocksOfQuery = gridDim.x;
    const int8_t *pssmB = nullptr; uint8_t *scoresB = nullptr; int LB = 0, capB = 0;
    if (a.queries) {                                  // multi-query launch: this workgroup's query (uniform scalar loads)
        const uint32_t qi = blockIdx.x / a.blocksPerQuery;
        blockInQuery = blockIdx.x - qi * a.blocksPerQuery;
        blocksOfQuery = a.blocksPerQuery;
        const GaplessQuery gq = a.queries[qi];
        a.pssm = gq.pssm; a.scores = gq.scores; a.queue = gq.queue; a.L = gq.L; a.cap = gq.cap;
        if constexpr (PAIRED) { pssmB = gq.pssmB; scoresB = gq.scoresB; LB = gq.LB; capB = gq.capB; }
    }

    // ---- build the LDS image from the int8 pssm (once per workgroup) ----
    // One item = (profile row, lane g, 4-register chunk k): 4 + 4 profile bytes -> one 16-byte slot, stored to both
    // copies.  All byte loads of an item are independent and the item loop is unrolled, so the build costs a couple of
    // global round trips instead of one per dword.
    {
        const int L = a.L;
        constexpr int nItems = (kAlphabet + 1) * 8 * NCH;
        constexpr int nIter = (nItems + BLOCK - 1) / BLOCK;
#pragma unroll
        for (int it = 0; it < nIter; it++) {
            const int idx = it * BLOCK + threadIdx.x;
            if (idx < nItems) {
                const int g = idx & 7, k = (idx >> 3) % NCH, row = (idx >> 3) / NCH;
                uint32_t v[4];
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const int r = 4 * k + w;
                    const int qlo = a.tileBase + g * 2 * R + r, qhi = qlo + R;
                    if (row == kDeadCode || r >= R) {
                        v[w] = kDead2;             // r >= R: padding of the last chunk, never read into the recurrence
                    } else if (PAIRED && g >= 4) {
                        const int blo = qlo - 8 * R, bhi = qhi - 8 * R;        // rows of query B
                        const int lo = blo < LB ? (int) pssmB[row * LB + blo] : 0;
                        const int hi = bhi < LB ? (int) pssmB[row * LB + bhi] : 0;
                        v[w] = f16ScaledBits(lo) | (f16ScaledBits(hi) << 16);
                    } else {
                        const int lo = qlo < L ? (int) a.pssm[row * L + qlo] : 0;
                        const int hi = qhi < L ? (int) a.pssm[row * L + qhi] : 0;
                        v[w] = f16ScaledBits(lo) | (f16ScaledBits(hi) << 16);
                    }
                }
                unsigned char *dst = smem + k * CHB + row * 256 + g * 16;
                *(u32x4 *) dst = (u32x4){v[0], v[1], v[2], v[3]};
                *(u32x4 *) (dst + 128) = (u32x4){v[0], v[1], v[2], v[3]};
            }
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int j = lane >> 3;          // target slot inside the stripe
    const int g = lane & 7;           // lane inside the target group
    const uint32_t laneOff = (uint32_t) (((j >> 1) & 1) * 128 + g * 16);       // < 256: fits the low address byte
    // v_perm selector for the hand-off: lo half <- hi half of the previous lane's last register,
    // hi half <- lo half of my own last register.  {S0 = prev (bytes 4..7), S1 = own (bytes 0..3)}.
    // First lane of a target group (untiled): lo half <- constant zero (selector byte 0x0c), so the value the dpp
    // move delivered from the neighbouring target is never looked at.
    const uint32_t sel = (!TILED && (g == 0 || (PAIRED && g == 4))) ? 0x01000c0cu : 0x01000706u;

    // The first item of a wave is static (the nWaves longest items), the following ones come from the atomic queue: no
    // ticket ramp at kernel start, and one 16-byte record per item keeps the dependent loads per stripe at two
    // (record, first column chunk) -- what matters for short queries (tools/ubench/gapless_ablate.hip, "v6").
    const uint32_t nWaves = blocksOfQuery * (BLOCK / 64);
    uint32_t w = __builtin_amdgcn_readfirstlane(blockInQuery * (BLOCK / 64) + (threadIdx.x >> 6));
    for (; w < a.nItems;) {
        const uint4 item = a.items[w];
        const uint32_t stripe = item.x;
        const bool split = (item.y >> 31) & 1;
        const uint32_t cBegin = (item.y >> 16) & 0x7fffu, cEnd = item.y & 0xffffu;
        const uint64_t soff = ((uint64_t) item.w << 32) | item.z;
        const uint4 *src = a.scan + soff + j;
        // border arrays use the scan layout at 2 bytes per residue: 32 bytes per (chunk, target)
        const uint4 *bin = TILED ? (const uint4 *) a.borderIn + (soff + j) * 2 : nullptr;
        uint4 *bout = TILED ? (uint4 *) a.borderOut + (soff + j) * 2 : nullptr;
        uint32_t carryPrevChunk = 0;                      // border value of the column before this chunk

        uint32_t S[R];
        uint32_t M = 0, M2 = 0;                           // two running maxima: no dependent max3 chain
#pragma unroll
        for (int r = 0; r < R; r++) S[r] = 0;

        uint4 nxt = src[(size_t) cBegin * 8];
        for (uint32_t c = cBegin; c < cEnd; c++) {
            const uint4 cur = nxt;
            if (c + 1 < cEnd) nxt = src[(size_t) (c + 1) * 8];
            const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
            uint32_t bi[8], bo[8];
            if constexpr (TILED) {
                if (!a.firstTile && g == 0) {
                    const uint4 x0 = bin[(size_t) c * 16], x1 = bin[(size_t) c * 16 + 1];
                    bi[0] = x0.x; bi[1] = x0.y; bi[2] = x0.z; bi[3] = x0.w; bi[4] = x1.x; bi[5] = x1.y; bi[6] = x1.z; bi[7] = x1.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++) bi[k] = 0;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) bo[k] = 0;
            }
#pragma unroll
            for (int b = 0; b < 16; b++) {
                // LDS row address, bytes {0, 0, code, laneOff}: S0 = residue word (selector bytes 4..7), S1 = laneOff
                const uint32_t addr = __builtin_amdgcn_perm(words[b >> 2], laneOff, 0x0c0c0000u | ((4u + (b & 3)) << 8));
                const unsigned char __attribute__((address_space(3))) *rowp =
                    (const unsigned char __attribute__((address_space(3))) *) (uintptr_t) addr;
                uint32_t P[4 * NCH];
#pragma unroll
                for (int k = 0; k < NCH; k++) {
                    const u32x4 v = *(const u32x4 __attribute__((address_space(3))) *) (rowp + k * CHB);
                    P[4 * k + 0] = v.x; P[4 * k + 1] = v.y; P[4 * k + 2] = v.z; P[4 * k + 3] = v.w;
                    // R % 4 == 1: only v.x of the last chunk feeds the recurrence and the compiler narrows the access to a
                    // ds_read_b32 -- whose 64 lanes are serviced together, so the four targets of a copy collide on one bank
                    // (SQ_LDS_BANK_CONFLICT = 15-20 % of the LDS cycles for R = 17, 21, 25, 29, zero for every other R:
                    // profiles/r02_a_pmc_*).  Keeping the unused words alive keeps the conflict-free 128-bit form.
                    if constexpr (R % 4 == 1) { if (k == NCH - 1) asm volatile("" :: "v"(v.y), "v"(v.z), "v"(v.w)); }
                }
                // diagonal hand-off
                uint32_t prev = __builtin_amdgcn_mov_dpp(S[R - 1], 0x111 /*row_shr:1*/, 0xf, 0xf, true);
                if constexpr (TILED) {
                    // first lane: the diagonal enters from the previous row tile, column b - 1
                    const uint32_t fromTile = (b == 0) ? carryPrevChunk : ((b & 1) ? (bi[(b - 1) >> 1] & 0xffffu) : (bi[(b - 1) >> 1] >> 16));
                    prev = (g == 0) ? (fromTile << 16) : prev;
                }
                const uint32_t in = __builtin_amdgcn_perm(prev, S[R - 1], sel);
#pragma unroll
                for (int r = R - 1; r >= 1; r--) S[r] = pk_addc_f16(S[r - 1], P[r]);
                S[0] = pk_addc_f16(in, P[0]);
#pragma unroll
                for (int r = 0; r + 3 < R; r += 4) {
                    M = pk_max3_f16(M, S[r], S[r + 1]);
                    M2 = pk_max3_f16(M2, S[r + 2], S[r + 3]);
                }
                if constexpr (R % 4 == 1) M = pk_max3_f16(M, S[R - 1], S[R - 1]);
                if constexpr (R % 4 == 2) M = pk_max3_f16(M, S[R - 2], S[R - 1]);
                if constexpr (R % 4 == 3) { M = pk_max3_f16(M, S[R - 3], S[R - 2]); M2 = pk_max3_f16(M2, S[R - 1], S[R - 1]); }
                if constexpr (TILED) {
                    // last lane: its bottom row (high half of the last register) is the next tile's input
                    const uint32_t v = S[R - 1] >> 16;
                    bo[b >> 1] |= (b & 1) ? (v << 16) : v;
                    // Without an ordering point per column the compiler hoists the LDS profile reads of all 16 unrolled
                    // columns above the border bookkeeping: 256 VGPRs + 346 spilled (kernel 12x slower).  An empty asm that
                    // ties the border word to the running maximum pins each column's work in place: 92 VGPRs, no scratch.
                    asm volatile("" : "+v"(bo[b >> 1]), "+v"(M), "+v"(M2));
                }
            }
            if constexpr (TILED) {
                carryPrevChunk = bi[7] >> 16;
                if (!a.lastTile && g == 7) {
                    bout[(size_t) c * 16] = make_uint4(bo[0], bo[1], bo[2], bo[3]);
                    bout[(size_t) c * 16 + 1] = make_uint4(bo[4], bo[5], bo[6], bo[7]);
                }
            }
        }
        // max over both strips and the 8 lanes of the group; non-negative FP16 values order like their bit patterns
        M = pk_max3_f16(M, M2, M2);
        int m = max((int) (M & 0xffff), (int) (M >> 16));
        m = max(m, __shfl_xor(m, 1));
        m = max(m, __shfl_xor(m, 2));
        if constexpr (!PAIRED) m = max(m, __shfl_xor(m, 4));
        const uint32_t tid = a.stripeTargets[stripe * kStripeTargets + j];
        if constexpr (PAIRED) { if (g == 4) { a.scores = scoresB; a.cap = capB; } }       // lane 4 reports query B
        if ((g == 0 || (PAIRED && g == 4)) && tid < a.nTargets) {
            if constexpr (TILED) {
                if (!a.firstTile) m = max(m, (int) a.scoreAcc[tid]);
                if (!a.lastTile) a.scoreAcc[tid] = (int16_t) m;
            }
            if (!TILED || a.lastTile) {
                int sc = (int) (__half2float(__ushort_as_half((unsigned short) m)) * 2048.0f + 0.5f);
                sc = sc < a.cap ? sc : a.cap;
                if (!split) {
                    a.scores[tid] = (uint8_t) sc;
                } else {
                    // column segment: byte-wise maximum into the (zeroed) score array
                    uint32_t *word = (uint32_t *) (a.scores + (tid & ~3u));
                    const int sh = (int) (tid & 3u) * 8;
                    uint32_t old = *word;
                    while ((int) ((old >> sh) & 0xffu) < sc) {
                        const uint32_t want = (old & ~(0xffu << sh)) | ((uint32_t) sc << sh);
                        const uint32_t seen = atomicCAS(word, old, want);
                        if (seen == old) break;
                        old = seen;
                    }
                }
            }
        }
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(a.queue, 1u);
        w = nWaves + __builtin_amdgcn_readfirstlane(t);
    }
}

} // namespace fs
