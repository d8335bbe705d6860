// k_sw.hpp -- dual-profile (3Di + AA) affine-gap local alignment, score + end position, gfx950.
//
// Semantics: StructureSmithWaterman::sw_sse2_word / sw_sse2_int (reference F/src/commons/StructureSmithWaterman.cpp:
// 1093-1322, 1325-1554) INCLUDING the striped kernel's quirk that the horizontal gap state E only sees vertical
// gaps opened inside the same stripe segment (lazy-F never updates E, :1230).  In row order that is
//     Hm = max(sat(Hdiag + s), E, Fseg)     t  = max(Hm - go, 0)
//     E' = max(E - ge, 0, t)                Fseg' = max(Fseg - ge, 0, t), reset to 0 at rows q % segLen == 0
//     Ffull' = max(Ffull - ge, 0, t)        H  = max(Hm, Ffull)
// (oracle/fs_oracle.c: fso_sw_rowmajor is the CPU statement of the same recurrence, tested against the literal
// lane-by-lane emulation and against the reference's compiled code).  Tie-breaks: first target column that reaches
// the maximum, then the smallest query row in that column (:1271-1292).
//
// Mapping: one wave64 per (query, target) pair; lane l owns R consecutive query rows; lanes run the anti-diagonal
// wavefront (lane l works on target column s - l at step s).  Everything that crosses a lane boundary -- H of the
// row above, both F chains, and the target residue itself -- rides a one-lane-per-step conveyor built from
// v_mov_b32_dpp wave_shr:1.  The forward-query and reversed-query passes of structurealign (structurealign.cpp:46,65)
// share the target and the control flow, so they are packed into the two int16 halves of every register:
// v_pk_add_i16 clamp reproduces _mm256_adds_epi16, v_pk_sub_u16 clamp reproduces _mm256_subs_epu16.
// Pairs whose int16 score saturates (32767) are re-run by the int32 instantiation with segLen = ceil(L/8), as
// alignScoreEndPos does (:313-336).  Queries longer than 64*R rows are processed in row tiles; the three
// boundary values per target column travel between tile launches through HBM (border arrays).
#pragma once
#include <hip/hip_runtime.h>
#include "fs_kernels.h"
#include "k_gapless.hpp"   // packed helpers

namespace fs {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct Pk16 {                        // two independent int16 problems per register (lo: forward, hi: reversed query)
    static constexpr bool packed = true;
    static __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) {
        return __builtin_bit_cast(uint32_t, (s16x2) (__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
    }
    static __device__ __forceinline__ uint32_t adds(uint32_t a, uint32_t b) { return pk_adds_i16(a, b); }
    static __device__ __forceinline__ uint32_t max(uint32_t a, uint32_t b) { return pk_max_i16(a, b); }
    static __device__ __forceinline__ uint32_t subus(uint32_t a, uint32_t b) {
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
    }
    // per-half all-ones where a > b (values are non-negative)
    static __device__ __forceinline__ uint32_t gtMask(uint32_t a, uint32_t b) {
        uint32_t d = subus(a, b);
        u16x2 one = {1, 1};
        u16x2 x = __builtin_elementwise_min(__builtin_bit_cast(u16x2, d), one);
        u16x2 z = {0, 0};
        return __builtin_bit_cast(uint32_t, (u16x2) (z - x));
    }
    static __device__ __forceinline__ uint32_t splat(uint32_t v) { return (v & 0xffffu) | (v << 16); }
};

struct I32 {                         // one int32 problem per register (re-run of saturated pairs)
    static constexpr bool packed = false;
    static __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return a + b; }
    static __device__ __forceinline__ uint32_t adds(uint32_t a, uint32_t b) { return a + b; }
    static __device__ __forceinline__ uint32_t max(uint32_t a, uint32_t b) { return (uint32_t) ::max((int) a, (int) b); }
    static __device__ __forceinline__ uint32_t subus(uint32_t a, uint32_t b) { return (uint32_t) ::max((int) a - (int) b, 0); }
    static __device__ __forceinline__ uint32_t gtMask(uint32_t a, uint32_t b) { return ((int) a > (int) b) ? 0xffffffffu : 0u; }
    static __device__ __forceinline__ uint32_t splat(uint32_t v) { return v; }
};

struct SwArgs {
    const uint8_t *aa;            // plain target AA codes (unmasked), may be NULL when !HAS_AA
    const uint8_t *ss;            // plain target 3Di codes (unmasked)
    const uint64_t *offsets;
    const int32_t *lengths;
    const uint32_t *targetIds;    // [nPairs]
    int nPairs;
    const uint32_t *profSS;       // LDS image of this tile, [21][64R] dwords
    const uint32_t *profAA;
    int tileBase;                 // first query row of this tile
    int rowsInTile;               // valid rows (<= 64R)
    int segLen;                   // ceil(L/16) for the int16 pass, ceil(L/8) for the int32 pass
    uint32_t go, ge;              // gap open / extend, splat for the arithmetic in use
    int tileIn, tileOut;          // this launch continues / is continued by another row tile
    const uint32_t *borderIn;     // [nPairs][borderStride][3]
    uint32_t *borderOut;
    uint32_t borderStride;
    uint64_t *keys;               // [nPairs][2] running best across tiles (packed: fwd, rev; int32: slot 0)
    int32_t *res0;                // fsgpu_swres[nPairs] as int32 x4 (packed: forward; int32: the direction re-run)
    int32_t *res1;                // packed only: reversed query
    // k_sw2 (multi-query launches, single-tile queries of one R class): workgroup b serves blocks[b]
    const struct SwBlockDesc *blocks;
    int dir;                      // k_sw2 only: 0 = forward-query halves of the image, 1 = reversed-query halves
    // k_sw, multi-query row-tiled launches (queries longer than 64*R rows): workgroup b serves tblocks[b] -- one row tile of up
    // to blockDim.x/64 pairs of one query; tile geometry and image come from the descriptor instead of the fields above, the
    // border columns of pair p start at slot borderBase[p] instead of p * borderStride
    const struct SwTileBlock *tblocks = nullptr;
    const uint32_t *borderBase = nullptr;
};

struct SwTileBlock {
    uint32_t imgOff;              // dword offset of this (query, tile) image inside SwArgs::profSS ([SS table][AA table])
    uint32_t firstPair;           // global pair index of wave 0
    uint16_t nPairs;              // live waves of this workgroup
    uint16_t rowsInTile;          // valid rows of this tile (<= 64R)
    uint32_t segLen;
    uint32_t tileBase;            // first query row of this tile
    uint32_t flags;               // bit 0: continues a previous tile (border in), bit 1: is continued (border out)
};

// One workgroup of a multi-query launch (k_sw2): up to 2 * (blockDim.x / 64) consecutive pairs of one query.
struct SwBlockDesc {
    uint32_t imgOff;              // dword offset of this query's LDS image inside SwArgs::profSS ([SS table][AA table])
    uint32_t firstPair;           // global pair index of wave 0 (targetIds / result arrays are concatenated over queries)
    uint16_t nPairs;              // live waves of this workgroup
    uint16_t rowsInTile;          // query length (<= 64R)
    uint32_t segLen;
};

__device__ __forceinline__ uint32_t wave_shr1(uint32_t x) {
    return __builtin_amdgcn_update_dpp(0u, x, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}

// One profile row of the LDS image (layout: swDwordIndex): 16-byte chunks at lane * 16, then the 8-byte / 4-byte remainders at their own lane
// strides.  `rowOff` is the byte offset of the row (target code * row bytes, possibly wave-varying); the lane-dependent parts of the addresses
// are loop invariants the caller keeps in registers (SwLaneBase), so every chunk costs ONE add (row offset + lane base) instead of
// base + row + lane.  (A layout with a single lane stride for all chunks -- planes of 8 bytes at R = 6 -- needs fewer adds still, but the
// compiler fuses its reads into ds_read2st64_b64, whose two halves hit the same banks: measured 9 % slower, dropped.)
template <int R>
struct SwLaneBase {
    const unsigned char *b16, *b8, *b4;
    __device__ __forceinline__ SwLaneBase(const unsigned char *table, int lane)
        : b16(table + lane * 16), b8(table + (R / 4) * 1024 + lane * 8), b4(table + (R / 4) * 1024 + (R % 4 == 3 ? 512 : 0) + lane * 4) {}
};
template <int R>
__device__ __forceinline__ void swLoadRow(const SwLaneBase<R> &lb, uint32_t rowOff, uint32_t (&P)[R]) {
    constexpr int full = (R / 4) * 4;
#pragma unroll
    for (int k = 0; k < R / 4; k++) {
        const uint4 v = *(const uint4 *) (lb.b16 + rowOff + k * 1024);
        P[4 * k] = v.x; P[4 * k + 1] = v.y; P[4 * k + 2] = v.z; P[4 * k + 3] = v.w;
    }
    constexpr int rem = R % 4;
    if constexpr (rem >= 2) {
        const uint2 v = *(const uint2 *) (lb.b8 + rowOff);
        P[full] = v.x; P[full + 1] = v.y;
        if constexpr (rem == 3) P[full + 2] = *(const uint32_t *) (lb.b4 + rowOff);
    } else if constexpr (rem == 1) {
        P[full] = *(const uint32_t *) (lb.b4 + rowOff);
    }
}

__device__ __forceinline__ uint64_t waveMaxU64(uint64_t k) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t lo = __shfl_xor((uint32_t) k, d), hi = __shfl_xor((uint32_t) (k >> 32), d);
        uint64_t o = ((uint64_t) hi << 32) | lo;
        k = o > k ? o : k;
    }
    return k;
}

template <int R, bool HAS_AA, typename A>
__global__ __launch_bounds__(512) void k_sw(SwArgs a) {
    constexpr int ROWB = swRowDwords(R) * 4;
    constexpr int TBL = kAlphabet * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t *profSS = a.profSS, *profAA = a.profAA;
    int rowsInTile = a.rowsInTile, segLen = a.segLen, tileBase = a.tileBase;
    bool tileIn = a.tileIn != 0, tileOut = a.tileOut != 0;
    int pairFirst = (int) (blockIdx.x * (blockDim.x >> 6)), pairEnd = a.nPairs;
    if (a.tblocks) {                                   // workgroup-uniform descriptor -> SGPRs
        const SwTileBlock bd = a.tblocks[blockIdx.x];
        profSS = a.profSS + __builtin_amdgcn_readfirstlane(bd.imgOff);
        profAA = profSS + TBL / 4;
        rowsInTile = __builtin_amdgcn_readfirstlane((int) bd.rowsInTile); segLen = __builtin_amdgcn_readfirstlane((int) bd.segLen);
        tileBase = __builtin_amdgcn_readfirstlane((int) bd.tileBase);
        const uint32_t fl = __builtin_amdgcn_readfirstlane(bd.flags);
        tileIn = (fl & 1u) != 0; tileOut = (fl & 2u) != 0;
        pairFirst = __builtin_amdgcn_readfirstlane((int) bd.firstPair); pairEnd = pairFirst + __builtin_amdgcn_readfirstlane((int) bd.nPairs);
    }
    {
        const uint4 *s3 = (const uint4 *) profSS;
        uint4 *d3 = (uint4 *) smem;
        for (int i = threadIdx.x; i < TBL / 16; i += blockDim.x) d3[i] = s3[i];
        if constexpr (HAS_AA) {
            const uint4 *sa = (const uint4 *) profAA;
            uint4 *da = (uint4 *) (smem + TBL);
            for (int i = threadIdx.x; i < TBL / 16; i += blockDim.x) da[i] = sa[i];
        }
    }
    __syncthreads();
    // The wavefront is a long dependent chain with little work per step: when a throughput kernel (the gapless scan
    // of another query) shares the SIMD, win the issue arbitration so this kernel's latency does not stretch.
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int waveInBlock = (int) (threadIdx.x >> 6);
    const int pair = __builtin_amdgcn_readfirstlane(pairFirst + waveInBlock);
    if (pair >= pairEnd) return;

    // wave-uniform pair parameters -> SGPRs, scalar loop control
    const uint32_t tid = __builtin_amdgcn_readfirstlane(a.targetIds[pair]);
    const int Lt = __builtin_amdgcn_readfirstlane(a.lengths[tid]);
    const uint64_t off64 = a.offsets[tid];
    const uint64_t off = ((uint64_t) __builtin_amdgcn_readfirstlane((uint32_t) (off64 >> 32)) << 32) |
                         (uint64_t) __builtin_amdgcn_readfirstlane((uint32_t) off64);
    const int nLanes = tileOut ? 64 : (rowsInTile + R - 1) / R;
    const int steps = Lt > 0 ? Lt + nLanes - 1 : 0;
    const bool laneActive = lane < nLanes;

    // segment-start masks for my rows
    uint32_t segmask[R];
#pragma unroll
    for (int r = 0; r < R; r++) segmask[r] = ((tileBase + lane * R + r) % segLen == 0) ? 0u : 0xffffffffu;

    uint32_t E[R], Hp[R], snap[R];
#pragma unroll
    for (int r = 0; r < R; r++) { E[r] = 0; Hp[r] = 0; snap[r] = 0; }
    uint32_t best = 0, bestcol = 0;
    uint32_t hOut = 0, fsegOut = 0, ffullOut = 0, hUpPrev = 0, tval = 0;
    uint32_t chunkCur = 0, chunkNxt = 0;
    uint32_t bh = 0, bfs = 0, bff = 0, bhN = 0, bfsN = 0, bffN = 0;   // border-in chunks
    uint32_t oh = 0, ofs = 0, off_ = 0;                                // border-out accumulators
    const size_t bBase = a.tblocks ? (size_t) __builtin_amdgcn_readfirstlane(a.borderBase[pair]) : (size_t) pair * a.borderStride;
    const SwLaneBase<R> lb3(smem, lane), lbA(smem + TBL, lane);

    auto loadChunk = [&](int s0) -> uint32_t {
        int col = s0 + lane;
        uint32_t v = 0;
        if (col < Lt) {
            uint32_t c3 = a.ss[off + col];
            v = c3 * (uint32_t) ROWB;
            if constexpr (HAS_AA) v |= ((uint32_t) a.aa[off + col] * (uint32_t) ROWB) << 16;
        }
        return v;
    };
    chunkNxt = loadChunk(0);
    if (tileIn && lane < Lt) {
        const uint32_t *p = a.borderIn + (bBase + lane) * 3;
        bhN = p[0]; bfsN = p[1]; bffN = p[2];
    }

    for (int s = 0; s < steps; s++) {
        if ((s & 63) == 0) {
            chunkCur = chunkNxt;
            chunkNxt = loadChunk(s + 64);
            if (tileIn) {
                bh = bhN; bfs = bfsN; bff = bffN;
                int col = s + 64 + lane;
                if (col < Lt) {
                    const uint32_t *p = a.borderIn + (bBase + col) * 3;
                    bhN = p[0]; bfsN = p[1]; bffN = p[2];
                }
            }
        }
        // ---- conveyor: everything moves one lane down per step ----
        uint32_t hUpNew = wave_shr1(hOut);
        uint32_t fsegIn = wave_shr1(fsegOut);
        uint32_t ffullIn = wave_shr1(ffullOut);
        tval = wave_shr1(tval);
        {
            const uint32_t tv = __builtin_amdgcn_readlane(chunkCur, s & 63);
            if (lane == 0) tval = tv;
            if (tileIn) {
                const uint32_t x0 = __builtin_amdgcn_readlane(bh, s & 63);
                const uint32_t x1 = __builtin_amdgcn_readlane(bfs, s & 63);
                const uint32_t x2 = __builtin_amdgcn_readlane(bff, s & 63);
                if (lane == 0) { hUpNew = x0; fsegIn = x1; ffullIn = x2; }
            }
        }
        const int col = s - lane;
        if (laneActive && col >= 0 && col < Lt) {
            uint32_t P3[R], PA[R];
            swLoadRow<R>(lb3, tval & 0xffffu, P3);
            if constexpr (HAS_AA) swLoadRow<R>(lbA, tval >> 16, PA);
            uint32_t diag = hUpPrev, fseg = fsegIn, ffull = ffullIn, cm = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint32_t sc = P3[r];
                if constexpr (HAS_AA) sc = A::add(PA[r], sc);
                uint32_t h = A::adds(diag, sc);
                h = A::max(h, E[r]);
                fseg &= segmask[r];
                h = A::max(h, fseg);
                const uint32_t t = A::subus(h, a.go);
                E[r] = A::max(A::subus(E[r], a.ge), t);
                const uint32_t hf = A::max(h, ffull);
                fseg = A::max(A::subus(fseg, a.ge), t);
                ffull = A::max(A::subus(ffull, a.ge), t);
                diag = Hp[r];
                Hp[r] = hf;
                cm = A::max(cm, hf);
            }
            hOut = Hp[R - 1]; fsegOut = fseg; ffullOut = ffull;
            const uint32_t nb = A::max(best, cm);
            if (nb != best) {
                // some lane of the wave sees a new maximum (most steps of the first half of a target): one v_bfi_b32 per
                // register.  The mask is made opaque so that the compiler keeps the bit-select form instead of
                // re-deriving it as two half-word compares + selects + a byte permute per register.
                uint32_t m = A::gtMask(cm, best);
                asm volatile("" : "+v"(m));
                const uint32_t cp = A::splat((uint32_t) col);
                bestcol = (m & cp) | (~m & bestcol);
#pragma unroll
                for (int r = 0; r < R; r++) snap[r] = (m & Hp[r]) | (~m & snap[r]);
                best = nb;
            }
        }
        hUpPrev = hUpNew;
        if (tileOut) {
            const int c63 = s - 63;
            if (c63 >= 0 && c63 < Lt) {
                const uint32_t x0 = __builtin_amdgcn_readlane(hOut, 63);
                const uint32_t x1 = __builtin_amdgcn_readlane(fsegOut, 63);
                const uint32_t x2 = __builtin_amdgcn_readlane(ffullOut, 63);
                const int slot = c63 & 63;
                if (lane == slot) { oh = x0; ofs = x1; off_ = x2; }
                if (slot == 63 || c63 == Lt - 1) {
                    if (lane <= slot) {
                        uint32_t *p = a.borderOut + (bBase + (size_t) (c63 - slot + lane)) * 3;
                        p[0] = oh; p[1] = ofs; p[2] = off_;
                    }
                }
            }
        }
    }

    // ---- per-lane candidate -> wave reduction with the reference's tie-breaks ----
    constexpr int NDIR = A::packed ? 2 : 1;
#pragma unroll
    for (int d = 0; d < NDIR; d++) {
        uint32_t b, c;
        if constexpr (A::packed) { b = (best >> (16 * d)) & 0xffffu; c = (bestcol >> (16 * d)) & 0xffffu; }
        else { b = best; c = bestcol; }
        int row = 0;
#pragma unroll
        for (int r = R - 1; r >= 0; r--) {
            uint32_t v;
            if constexpr (A::packed) v = (snap[r] >> (16 * d)) & 0xffffu; else v = snap[r];
            if (v == b) row = r;
        }
        const uint32_t q = (uint32_t) (tileBase + lane * R + row);
        uint64_t key = ((uint64_t) b << 32) | ((uint64_t) (0xffffu - (c & 0xffffu)) << 16) | (uint64_t) (0xffffu - (q & 0xffffu));
        key = waveMaxU64(key);
        if (tileIn) { const uint64_t pk = a.keys[(size_t) pair * 2 + d]; key = pk > key ? pk : key; }
        if (lane == 0) {
            if (tileOut) {
                a.keys[(size_t) pair * 2 + d] = key;
            } else {
                int32_t *res = (d == 0 ? a.res0 : a.res1) + (size_t) pair * 4;
                res[0] = (int32_t) (key >> 32);
                res[1] = (int32_t) (0xffffu - (uint32_t) (key & 0xffffu));
                res[2] = (int32_t) (0xffffu - (uint32_t) ((key >> 16) & 0xffffu));
                res[3] = A::packed ? 1 : 2;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// k_sw2 -- ONE direction of the structure SW for TWO targets per wave.
// structurealign needs the reversed-query score only for pairs whose forward score passes the coverage and e-value
// gates (structurealign.cpp:55-65); on real hit lists that is 3-15 % of the pairs.  So instead of packing (forward,
// reversed) of one pair into the two int16 halves (k_sw), this kernel packs the SAME direction of two pairs of one
// query: half the waves for the forward pass, and the reversed pass only over the survivors.  Recurrence, tie-breaks
// and result format are k_sw's; per target column the profile rows of both targets are read from the (fwd | rev)
// LDS image and the halves of the chosen direction are merged with one v_perm_b32 per register.  The shorter target
// of a wave is continued with code 21, an extra image row of INT16_MIN (3Di table) / 0 (AA table): its H, E, F can then
// only decay, never set a new maximum.  Single-tile queries in multi-query launches only; int16-saturated pairs are
// re-run by the caller through k_sw's int32 instantiation like before.
// ------------------------------------------------------------------------------------------------------------
constexpr int kSw2Rows = kAlphabet + 1;

template <int R, bool HAS_AA>
__global__ __launch_bounds__(1024) void k_sw2(SwArgs a) {
    using A = Pk16;
    constexpr int ROWB = swRowDwords(R) * 4;
    constexpr int TBL = kSw2Rows * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SwBlockDesc bd = a.blocks[blockIdx.x];
    // workgroup-uniform values -> SGPRs (scalar loop control, scalar address arithmetic)
    const uint32_t *profSS = a.profSS + __builtin_amdgcn_readfirstlane(bd.imgOff);
    const int rowsInTile = __builtin_amdgcn_readfirstlane((int) bd.rowsInTile), segLen = __builtin_amdgcn_readfirstlane((int) bd.segLen);
    const int pairBase = __builtin_amdgcn_readfirstlane((int) bd.firstPair), pairsHere = __builtin_amdgcn_readfirstlane((int) bd.nPairs);
    {
        const uint4 *s3 = (const uint4 *) profSS;
        uint4 *d3 = (uint4 *) smem;
        constexpr int n16 = (HAS_AA ? 2 : 1) * TBL / 16;
        for (int i = threadIdx.x; i < n16; i += blockDim.x) d3[i] = s3[i];
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int waveInBlock = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));     // wave-uniform: keep the pair bookkeeping scalar
    if (2 * waveInBlock >= pairsHere) return;
    const bool hasB = 2 * waveInBlock + 1 < pairsHere;
    const int pairA = __builtin_amdgcn_readfirstlane(pairBase + 2 * waveInBlock);

    const uint32_t tidA = __builtin_amdgcn_readfirstlane(a.targetIds[pairA]);
    const uint32_t tidB = __builtin_amdgcn_readfirstlane(a.targetIds[pairA + (hasB ? 1 : 0)]);
    const int LtA = __builtin_amdgcn_readfirstlane(a.lengths[tidA]);
    const int LtB = hasB ? __builtin_amdgcn_readfirstlane(a.lengths[tidB]) : 0;
    const uint64_t oA = a.offsets[tidA], oB = a.offsets[tidB];
    const uint64_t offA = ((uint64_t) __builtin_amdgcn_readfirstlane((uint32_t) (oA >> 32)) << 32) | (uint64_t) __builtin_amdgcn_readfirstlane((uint32_t) oA);
    const uint64_t offB = ((uint64_t) __builtin_amdgcn_readfirstlane((uint32_t) (oB >> 32)) << 32) | (uint64_t) __builtin_amdgcn_readfirstlane((uint32_t) oB);
    const int Lt = LtA > LtB ? LtA : LtB;          // the caller orders pairs longest first: normally LtA
    const int nLanes = (rowsInTile + R - 1) / R;
    const int steps = Lt > 0 ? Lt + nLanes - 1 : 0;
    // merge selector: {S0 = target B's dword (bytes 4..7), S1 = target A's dword (bytes 0..3)} -> (A.dir, B.dir)
    const uint32_t sel = a.dir ? 0x07060302u : 0x05040100u;

    uint32_t segmask[R];
#pragma unroll
    for (int r = 0; r < R; r++) segmask[r] = ((lane * R + r) % segLen == 0) ? 0u : 0xffffffffu;
    uint32_t E[R], Hp[R], snap[R];
#pragma unroll
    for (int r = 0; r < R; r++) { E[r] = 0; Hp[r] = 0; snap[r] = 0; }
    uint32_t best = 0, bestcol = 0;
    constexpr uint32_t kDeadOff = (uint32_t) kAlphabet * (uint32_t) ROWB;
    uint32_t hOut = 0, fsegOut = 0, ffullOut = 0, hUpPrev = 0, tval = kDeadOff | (kDeadOff << 16), tvalAA = kDeadOff | (kDeadOff << 16);
    uint32_t chunkCur = 0, chunkNxt = 0, chunkAACur = 0, chunkAANxt = 0;

    auto loadChunk = [&](int s0, uint32_t &vAA) -> uint32_t {
        const int col = s0 + lane;
        uint32_t v = kDeadOff | (kDeadOff << 16);
        vAA = v;
        if (col < LtA) {
            v = (v & 0xffff0000u) | ((uint32_t) a.ss[offA + col] * (uint32_t) ROWB);
            if constexpr (HAS_AA) vAA = (vAA & 0xffff0000u) | ((uint32_t) a.aa[offA + col] * (uint32_t) ROWB);
        }
        if (col < LtB) {
            v = (v & 0xffffu) | (((uint32_t) a.ss[offB + col] * (uint32_t) ROWB) << 16);
            if constexpr (HAS_AA) vAA = (vAA & 0xffffu) | (((uint32_t) a.aa[offB + col] * (uint32_t) ROWB) << 16);
        }
        return v;
    };
    chunkNxt = loadChunk(0, chunkAANxt);

    // two steps per loop iteration, each an inlined copy of the body: the diagonal registers (old H of a row = diag of the next) then
    // alternate between two register sets instead of being moved back into place at every step (5 v_mov per step at R = 6)
    const SwLaneBase<R> lb3(smem, lane), lbA(smem + TBL, lane);
    auto step = [&](const int s) {
        if ((s & 63) == 0) {
            chunkCur = chunkNxt; chunkAACur = chunkAANxt;
            chunkNxt = loadChunk(s + 64, chunkAANxt);
        }
        uint32_t hUpNew = wave_shr1(hOut);
        const uint32_t fsegIn = wave_shr1(fsegOut);
        const uint32_t ffullIn = wave_shr1(ffullOut);
        tval = wave_shr1(tval);
        {
            const uint32_t tv = __builtin_amdgcn_readlane(chunkCur, s & 63);
            if (lane == 0) tval = tv;
        }
        if constexpr (HAS_AA) {
            tvalAA = wave_shr1(tvalAA);
            const uint32_t tv = __builtin_amdgcn_readlane(chunkAACur, s & 63);
            if (lane == 0) tvalAA = tv;
        }
        const int col = s - lane;
        // No lane is masked off.  A lane that has not reached its first column yet still holds the "past the end" code it was initialised
        // with (score INT16_MIN: H, E, F stay 0), columns beyond a target's end read the same row (values can only decay), and rows beyond
        // the query -- inside the last lane or in whole lanes -- score 0 against everything: such a cell repeats the value of its diagonal
        // neighbour, which was computed a step earlier, so it never sets a NEW maximum.  Without the exec mask the new H of a row is
        // written in place instead of being merged into the persistent registers by a move per row.
        {
            uint32_t PA[R], PB[R];
            swLoadRow<R>(lb3, tval & 0xffffu, PA);
            swLoadRow<R>(lb3, tval >> 16, PB);
            if constexpr (HAS_AA) {
                uint32_t QA[R], QB[R];
                swLoadRow<R>(lbA, tvalAA & 0xffffu, QA);
                swLoadRow<R>(lbA, tvalAA >> 16, QB);
#pragma unroll
                for (int r = 0; r < R; r++) { PA[r] = A::add(QA[r], PA[r]); PB[r] = A::add(QB[r], PB[r]); }
            }
            uint32_t diag = hUpPrev, fseg = fsegIn, ffull = ffullIn, cm = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t sc = __builtin_amdgcn_perm(PB[r], PA[r], sel);
                uint32_t h = A::adds(diag, sc);
                h = A::max(h, E[r]);
                fseg &= segmask[r];
                h = A::max(h, fseg);
                const uint32_t t = A::subus(h, a.go);
                E[r] = A::max(A::subus(E[r], a.ge), t);
                const uint32_t hf = A::max(h, ffull);
                fseg = A::max(A::subus(fseg, a.ge), t);
                ffull = A::max(A::subus(ffull, a.ge), t);
                diag = Hp[r];
                Hp[r] = hf;
                cm = A::max(cm, hf);
            }
            hOut = Hp[R - 1]; fsegOut = fseg; ffullOut = ffull;
            const uint32_t nb = A::max(best, cm);
            if (nb != best) {
                uint32_t m = A::gtMask(cm, best);
                asm volatile("" : "+v"(m));
                const uint32_t cp = A::splat((uint32_t) col);
                bestcol = (m & cp) | (~m & bestcol);
#pragma unroll
                for (int r = 0; r < R; r++) snap[r] = (m & Hp[r]) | (~m & snap[r]);
                best = nb;
            }
        }
        hUpPrev = hUpNew;
    };
    {
        int s = 0;
        for (; s + 1 < steps; s += 2) { step(s); step(s + 1); }
        if (s < steps) step(s);
    }

#pragma unroll
    for (int d = 0; d < 2; d++) {
        const uint32_t b = (best >> (16 * d)) & 0xffffu, c = (bestcol >> (16 * d)) & 0xffffu;
        int row = 0;
#pragma unroll
        for (int r = R - 1; r >= 0; r--) {
            const uint32_t v = (snap[r] >> (16 * d)) & 0xffffu;
            if (v == b) row = r;
        }
        const uint32_t q = (uint32_t) (lane * R + row);
        uint64_t key = ((uint64_t) b << 32) | ((uint64_t) (0xffffu - (c & 0xffffu)) << 16) | (uint64_t) (0xffffu - (q & 0xffffu));
        key = waveMaxU64(key);
        if (lane == 0 && (d == 0 || hasB)) {
            int32_t *res = a.res0 + (size_t) (pairA + d) * 4;
            res[0] = (int32_t) (key >> 32);
            res[1] = (int32_t) (0xffffu - (uint32_t) (key & 0xffffu));
            res[2] = (int32_t) (0xffffu - (uint32_t) ((key >> 16) & 0xffffu));
            res[3] = 1;
        }
    }
}

} // namespace fs
