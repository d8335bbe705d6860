// k_sw3.hpp -- the batch form of the structure Smith-Waterman (score + end position), third layout: gfx950.
//
// Same recurrence, tie-breaks and result format as k_sw / k_sw2 (k_sw.hpp; reference F/src/commons/StructureSmithWaterman.cpp:
// 1093-1322 sw_sse2_word with the striped kernel's segment-local E quirk), one direction of TWO targets of one query in the int16
// halves of every register (k_sw2's packing).  What changes is the shape of the wavefront and where its inputs come from:
//   * HL lanes per target pair instead of the whole wave: HL = 32 puts two target pairs (four targets) into one wave64, each lane owning
//     R <= 16 consecutive query rows.  Against 64 lanes x R/2 rows that halves the wavefront's fill / drain (HL - 1 steps per target
//     instead of 63), halves the per-step bookkeeping per row, and cuts the row padding (rows are padded to HL * R: 352 instead of 384
//     rows for a 350-residue query).  HL = 64 (one pair per wave) carries queries of 513..1024 rows in one piece.
//     Round 6: HL = 16 (four pairs = eight targets per wave, R <= 24 rows per lane) for queries of up to 384 rows against targets of up to
//     FSGPU_SW3_MID columns: per DP cell it halves the fill / drain again (15 steps), halves the per-step bookkeeping (ring, DPP moves, best
//     tracking: ~18 instructions next to 15 R) and pads to 16 R rows (336 for a 330-residue query, 352 with 32 lanes); the lane that starts
//     a pair takes zeros from the DPP row shift itself (row_shr:1 with bound_ctrl) instead of two v_and.
//   * the target codes come from a ring in LDS (2 HL columns per target pair, per column the LDS row offsets of target A and target B in
//     the 3Di table [and in the AA table]) that the wave refills 16 columns at a time, instead of riding a register conveyor of
//     v_mov_dpp + v_readlane + v_cndmask per step and table; a lane fetches the entry of its NEXT column while it works on this one.
//   * the LDS images are built ON THE DEVICE (k_sw3_image) from the query's codes, its position biases and the two 21 x 21 matrices:
//     profile[a][i] = mat[a][q_i] + bias_i is all a structurealign profile is (StructureSmithWaterman.cpp:1566-1640), so the host ships
//     6 L bytes per query instead of building and copying 2 x 22 x 64 R dwords.
// Everything that crosses a lane (H of the row above, the two F chains) still moves by v_mov_b32_dpp wave_shr:1; the first lane of the
// second pair (lane 32) drops what lane 31 hands it (two v_and; its segment-local F is cleared by the segment mask of query row 0).
#pragma once
#include "k_sw.hpp"

namespace fs {

// ---- LDS image of one table and ONE direction for HL lanes x R rows per lane ----
// A pass runs one direction, so an image holds that direction's scores only, as int16 pairs: dword j of lane l = (row l*R + 2j) | (row l*R +
// 2j + 1) << 16 -- D = ceil(R / 2) dwords per lane and profile row, half the LDS footprint and half the LDS reads of a (forward | reversed)
// image; the (target A | target B) register of a row is one v_perm_b32 of the two targets' dwords either way (even rows take the low
// halves, odd rows the high halves), and the AA table is added on the packed row pairs.
// The D dwords are fetched in chunks of 4 (a remainder of 3 is fetched as a chunk of 4 whose last dword is unused: same LDS footprint as a
// 2 + 1 split, one address and two reads less), a remainder of 2 as one 8-byte read, a remainder of 1 as one 4-byte read.  A 4-dword chunk
// k is a plane of HL lanes x 16 B, the 2-dword remainder a plane of HL x 8 B, the 1-dword remainder a plane of 64 x 4 B (for HL = 32 it
// holds two copies: lanes 32..63 of the wave read the second one, a ds_read_b32 services all 64 lanes at once; for HL = 16 four copies, and
// the 2-dword plane two: lanes 16..31 and 48..63 read the second).  Every plane is a multiple of 256 B and so is the row: the bank of an
// access depends on the lane only, although every lane group reads a different profile row.
__host__ __device__ constexpr int sw3Dw(int R) { return (R + 1) / 2; }
__host__ __device__ constexpr int sw3Planes16(int R) { return sw3Dw(R) / 4 + ((sw3Dw(R) % 4) == 3 ? 1 : 0); }
__host__ __device__ constexpr int sw3Plane8Lanes(int HL) { return HL < 32 ? 32 : HL; }      // lanes (copies included) of the 2-dword plane
__host__ __device__ constexpr int sw3RowBytes(int R, int HL) { return sw3Planes16(R) * HL * 16 + ((sw3Dw(R) % 4) == 2 ? sw3Plane8Lanes(HL) * 8 : 0) + ((sw3Dw(R) % 4) == 1 ? 256 : 0); }
// dword index inside one profile row for (lane l of its target pair, dword j < D); copies of the remainder planes: sw3RemCopies / sw3RemCopyStride
__host__ __device__ constexpr int sw3DwordIndex(int R, int HL, int l, int j) {
    const int in16 = sw3Planes16(R) * 4;
    if (j < in16) return (j / 4) * HL * 4 + l * 4 + (j % 4);
    const int off = sw3Planes16(R) * HL * 4;
    if ((sw3Dw(R) % 4) == 2) return off + l * 2 + (j - in16);
    return off + l;
}
// the remainder plane of a row (dwords j >= 4 * sw3Planes16) is stored sw3RemCopies times, sw3RemCopyStride dwords apart
__host__ __device__ constexpr int sw3RemCopies(int R, int HL) { return (sw3Dw(R) % 4) == 2 ? sw3Plane8Lanes(HL) / HL : (sw3Dw(R) % 4) == 1 ? 64 / HL : 1; }
__host__ __device__ constexpr int sw3RemCopyStride(int R, int HL) { return (sw3Dw(R) % 4) == 2 ? HL * 2 : HL; }
constexpr int kSw3MaxR = 16;           // rows per lane of the 32- and 64-lane shapes
constexpr int kSw3MaxR16 = 24;         // of the 16-lane shape
__host__ __device__ constexpr int sw3MaxR(int HL) { return HL == 16 ? kSw3MaxR16 : kSw3MaxR; }
constexpr int kSw3Chunk = 16;          // target columns per ring refill
// columns of a target pair held in LDS: the refill at step s (a multiple of 16) overwrites the slots of columns s + 16 - RING .. s + 31 - RING while
// lane HL - 1 still reads column s + 2 - HL: RING >= HL + 29
__host__ __device__ constexpr int sw3RingCols(int HL) { return HL < 32 ? 64 : 2 * HL; }
__host__ __device__ constexpr int sw3TableBytes(int R, int HL) { return kSw2Rows * sw3RowBytes(R, HL); }
__host__ __device__ constexpr int sw3ImageBytes(int R, int HL, bool hasAA) { return (hasAA ? 2 : 1) * sw3TableBytes(R, HL); }   // one direction
__host__ __device__ constexpr int sw3RingBytes(int HL, bool hasAA) { return sw3RingCols(HL) * (hasAA ? 16 : 8); }
// LDS offset of the first ring (rings are aligned to their size: the position wraps by and / or)
__host__ __device__ constexpr int sw3RingBase(int R, int HL, bool hasAA) {
    const int img = sw3ImageBytes(R, HL, hasAA), rb = sw3RingBytes(HL, hasAA);
    return (img + rb - 1) / rb * rb;
}
// dynamic LDS of a workgroup of `waves` waves
__host__ __device__ constexpr int sw3LdsBytes(int R, int HL, bool hasAA, int waves) {
    return sw3RingBase(R, HL, hasAA) + waves * (64 / HL) * sw3RingBytes(HL, hasAA);
}

struct Sw3Args {
    const uint8_t *aa;            // plain target AA codes (unmasked), may be NULL when !HAS_AA
    const uint8_t *ss;            // plain target 3Di codes (unmasked)
    const uint64_t *offsets;
    const int32_t *lengths;
    const uint32_t *targetIds;    // pairs of all queries of the call, concatenated
    const uint32_t *img;          // images of all queries ([forward: 3Di table, AA table][reversed: 3Di table, AA table] each), dwords
    const SwBlockDesc *blocks;    // workgroup b serves blocks[b]: up to (waves * 128 / HL) consecutive pairs of one query
    uint32_t go, ge;              // splat
    int dir;                      // 0 = forward-query image, 1 = reversed-query image
    int32_t *res0;                // fsgpu_swres[pairs] as int32 x 4
};

template <int R, int HL>
struct Sw3LaneBase {
    uint32_t b16, brem;            // LDS byte addresses: 16-byte planes, the 8- or 4-byte remainder plane
    __device__ __forceinline__ Sw3LaneBase(uint32_t table, int l, int lane64)
        : b16(table + l * 16), brem(table + sw3Planes16(R) * HL * 16 + ((sw3Dw(R) % 4) == 2 ? (lane64 & (sw3Plane8Lanes(HL) - 1)) * 8 : lane64 * 4)) {}
};
// rowOff: byte offset of the profile row inside its table; P: the lane's D = ceil(R / 2) row-pair dwords
template <int R, int HL>
__device__ __forceinline__ void sw3LoadRow(const unsigned char *smem, const Sw3LaneBase<R, HL> &lb, uint32_t rowOff, uint32_t (&P)[sw3Dw(R)]) {
    constexpr int D = sw3Dw(R), n16 = sw3Planes16(R);
    if constexpr (n16 > 0) {
        const unsigned char *p = smem + (lb.b16 + rowOff);
#pragma unroll
        for (int k = 0; k < n16; k++) {
            const uint4 v = *(const uint4 *) (p + k * HL * 16);
            P[4 * k] = v.x;
            if (4 * k + 1 < D) P[4 * k + 1] = v.y;
            if (4 * k + 2 < D) P[4 * k + 2] = v.z;
            if (4 * k + 3 < D) P[4 * k + 3] = v.w;
        }
    }
    if constexpr ((D % 4) == 2) {
        const uint2 v = *(const uint2 *) (smem + (lb.brem + rowOff));
        P[D - 2] = v.x; P[D - 1] = v.y;
    } else if constexpr ((D % 4) == 1) {
        P[D - 1] = *(const uint32_t *) (smem + (lb.brem + rowOff));
    }
}

__device__ __forceinline__ uint32_t row_shr1(uint32_t x) {
    return __builtin_amdgcn_update_dpp(0u, x, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
}

template <int HL>
__device__ __forceinline__ uint64_t groupMaxU64(uint64_t k) {
#pragma unroll
    for (int d = 1; d < HL; d <<= 1) {
        uint32_t lo = __shfl_xor((uint32_t) k, d), hi = __shfl_xor((uint32_t) (k >> 32), d);
        uint64_t o = ((uint64_t) hi << 32) | lo;
        k = o > k ? o : k;
    }
    return k;
}

// the work of one workgroup for R rows per lane (the kernel below dispatches on the workgroup's R)
template <int R, bool HAS_AA, int HL>
__device__ __forceinline__ void sw3Body(const Sw3Args &a, const SwBlockDesc &bd, unsigned char *smem) {
    using A = Pk16;
    static_assert(HL == 16 || HL == 32 || HL == 64, "lanes per target pair");
    constexpr int ROWB = sw3RowBytes(R, HL);
    constexpr int TBL = kSw2Rows * ROWB;
    constexpr int IMG = (HAS_AA ? 2 : 1) * TBL;       // one direction
    constexpr int D = sw3Dw(R);
    constexpr int GPW = 64 / HL;                       // target pairs (groups of HL lanes) per wave
    constexpr int RING = sw3RingCols(HL);              // columns
    constexpr int EB = HAS_AA ? 16 : 8;                // bytes per ring entry: row offsets of (A, B) in the 3Di table [, (A, B) in the AA table]
    constexpr int RINGB = RING * EB;
    constexpr uint32_t kDeadOff = (uint32_t) kAlphabet * (uint32_t) ROWB;
    const uint32_t *imgSrc = a.img + __builtin_amdgcn_readfirstlane(bd.imgOff) + (a.dir ? IMG / 4 : 0);
    const int rows = __builtin_amdgcn_readfirstlane((int) bd.rowsInTile), segLen = __builtin_amdgcn_readfirstlane((int) bd.segLen);
    const int pairBase = __builtin_amdgcn_readfirstlane((int) bd.firstPair), pairsHere = __builtin_amdgcn_readfirstlane((int) bd.nPairs);
    {
        const uint4 *s3 = (const uint4 *) imgSrc;
        uint4 *d3 = (uint4 *) smem;
        for (int i = threadIdx.x; i < IMG / 16; i += blockDim.x) d3[i] = s3[i];
    }
    const int lane = threadIdx.x & 63;
    const int l = lane & (HL - 1);
    const int grp = lane / HL;
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    // LDS byte address of this target pair's ring; aligned to its size, so a position wraps by (pos & (RINGB - 1)) | ring
    const uint32_t ring = (uint32_t) sw3RingBase(R, HL, HAS_AA) + (uint32_t) ((wave * GPW + grp) * RINGB);
    // every column slot starts as "past the end" (the lanes behind lane 0 read columns -1, -2, ... during the fill)
#pragma unroll
    for (int k = 0; k < RING / HL; k++) {
        const uint32_t at = ring + (uint32_t) ((l + k * HL) * EB);
        if constexpr (HAS_AA) *(uint4 *) (smem + at) = make_uint4(kDeadOff, kDeadOff, kDeadOff, kDeadOff);
        else *(uint2 *) (smem + at) = make_uint2(kDeadOff, kDeadOff);
    }
    __syncthreads();
    const int pairsW = pairsHere - wave * 2 * GPW;     // pairs of this wave (scalar)
    if (pairsW <= 0) return;
    const int pA = pairBase + wave * 2 * GPW + 2 * grp;
    const bool hasA = 2 * grp < pairsW, hasB = 2 * grp + 1 < pairsW;
    const uint32_t tidA = hasA ? a.targetIds[pA] : 0u, tidB = hasB ? a.targetIds[pA + 1] : 0u;
    const int LtA = hasA ? a.lengths[tidA] : 0, LtB = hasB ? a.lengths[tidB] : 0;
    const uint64_t offA = a.offsets[tidA], offB = a.offsets[tidB];
    int LtW = LtA > LtB ? LtA : LtB;                   // longest target of the wave -> scalar loop bound
    if constexpr (GPW == 4) { const int o = __shfl_xor(LtW, 16); LtW = o > LtW ? o : LtW; }
    if constexpr (GPW >= 2) { const int o = __shfl_xor(LtW, 32); LtW = o > LtW ? o : LtW; }
    LtW = __builtin_amdgcn_readfirstlane(LtW);
    // The wavefront is a long dependent chain: the waves with the longest targets are the critical path of the launch, they win the issue
    // arbitration of their SIMD; every wave of this kernel stays above a co-running throughput kernel (the gapless scan of another query).
    if (LtW > 1024) __builtin_amdgcn_s_setprio(3);
    else if (LtW > 512) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(1);
    const int nLanes = (rows + R - 1) / R;
    const int steps = LtW > 0 ? LtW + nLanes - 1 : 0;
    // merge selectors: {S0 = target B's dword (bytes 4..7), S1 = target A's dword (bytes 0..3)} -> (A | B << 16) of the even / odd row of a pair
    constexpr uint32_t selEven = 0x05040100u, selOdd = 0x07060302u;

    // The striped reference kernel's F is local to a segment of segLen = ceil(L / 16) query rows (k_sw.hpp).  With 16 lanes per target pair
    // R = ceil(L / 16) IS segLen: a lane's rows are exactly one segment, its segment-local F never leaves the lane -- no mask per row, no
    // DPP move, 13 instead of 14 instructions per register row.
    constexpr bool LANESEG = HL == 16;
    uint32_t segmask[LANESEG ? 1 : R];
    if constexpr (!LANESEG) {
#pragma unroll
        for (int r = 0; r < R; r++) segmask[r] = ((l * R + r) % segLen == 0) ? 0u : 0xffffffffu;
    }
    uint32_t E[R], Hp[R], snap[R];
#pragma unroll
    for (int r = 0; r < R; r++) { E[r] = 0; Hp[r] = 0; snap[r] = 0; }
    uint32_t best = 0, bestcol = 0;
    uint32_t hOut = 0, fsegOut = 0, ffullOut = 0, hUpPrev = 0;
    const uint32_t inMask = (HL == 32 && lane == 32) ? 0u : 0xffffffffu;     // lane 32 starts a target pair: nothing comes in from lane 31

    // ---- target-code ring: the first kSw3Chunk lanes of a pair fetch one column each, kSw3Chunk steps ahead of its first use ----
    uint32_t n3A = kDeadOff, n3B = kDeadOff, nAA = kDeadOff, nAB = kDeadOff;          // the chunk in flight (registers)
    auto loadChunk = [&](int c0) {
        n3A = n3B = nAA = nAB = kDeadOff;
        const int col = c0 + l;
        if (l < kSw3Chunk) {
            if (col < LtA) {
                n3A = (uint32_t) a.ss[offA + col] * (uint32_t) ROWB;
                if constexpr (HAS_AA) nAA = (uint32_t) a.aa[offA + col] * (uint32_t) ROWB;
            }
            if (col < LtB) {
                n3B = (uint32_t) a.ss[offB + col] * (uint32_t) ROWB;
                if constexpr (HAS_AA) nAB = (uint32_t) a.aa[offB + col] * (uint32_t) ROWB;
            }
        }
    };
    auto storeChunk = [&](int c0) {
        if (l < kSw3Chunk) {
            const uint32_t at = ring + (uint32_t) (((c0 + l) & (RING - 1)) * EB);
            if constexpr (HAS_AA) *(uint4 *) (smem + at) = make_uint4(n3A, n3B, nAA, nAB);
            else *(uint2 *) (smem + at) = make_uint2(n3A, n3B);
        }
        // the ring is shared by the lanes of this wave only: LDS executes a wave's accesses in order, the compiler must keep them in order too
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    loadChunk(0); storeChunk(0);
    loadChunk(kSw3Chunk);
    uint32_t ringAt = ring + (uint32_t) (((0 - l) & (RING - 1)) * EB);       // LDS address of my current column's entry
    uint32_t t3A, t3B, tAA = 0, tAB = 0;                                      // entry of the column of the next step
    auto readEntry = [&]() {
        if constexpr (HAS_AA) { const uint4 e = *(const uint4 *) (smem + ringAt); t3A = e.x; t3B = e.y; tAA = e.z; tAB = e.w; }
        else { const uint2 e = *(const uint2 *) (smem + ringAt); t3A = e.x; t3B = e.y; }
    };
    readEntry();

    const Sw3LaneBase<R, HL> lb3(0u, l, lane), lbA((uint32_t) TBL, l, lane);
    auto step = [&](const int s) {
        if ((s & (kSw3Chunk - 1)) == 0) {
            storeChunk(s + kSw3Chunk);
            loadChunk(s + 2 * kSw3Chunk);
        }
        // what the lane above hands down; the first lane of a pair takes zeros (16 lanes: a DPP row is a pair, the shift itself fills them in)
        uint32_t hUpNew = HL == 16 ? row_shr1(hOut) : wave_shr1(hOut);
        uint32_t fsegIn = 0;
        if constexpr (!LANESEG) fsegIn = wave_shr1(fsegOut);
        uint32_t ffullIn = HL == 16 ? row_shr1(ffullOut) : wave_shr1(ffullOut);
        if constexpr (HL == 32) { hUpNew &= inMask; ffullIn &= inMask; }
        const uint32_t r3A = t3A, r3B = t3B, rAA = tAA, rAB = tAB;
        // the entry of the column this lane works on in the NEXT step
        ringAt = ((ringAt + EB) & (uint32_t) (RINGB - 1)) | ring;
        readEntry();
        const int col = s - l;
        // No lane is masked off (see k_sw2): a lane before its first column or past its target's end reads the "past the end" row, rows
        // beyond the query score 0 -- neither can set a new maximum.
        {
            uint32_t PA[D], PB[D];
            sw3LoadRow<R, HL>(smem, lb3, r3A, PA);
            sw3LoadRow<R, HL>(smem, lb3, r3B, PB);
            if constexpr (HAS_AA) {
                uint32_t QA[D], QB[D];
                sw3LoadRow<R, HL>(smem, lbA, rAA, QA);
                sw3LoadRow<R, HL>(smem, lbA, rAB, QB);
#pragma unroll
                for (int j = 0; j < D; j++) { PA[j] = A::add(QA[j], PA[j]); PB[j] = A::add(QB[j], PB[j]); }
            }
            uint32_t diag = hUpPrev, fseg = fsegIn, ffull = ffullIn, cm = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t sc = __builtin_amdgcn_perm(PB[r / 2], PA[r / 2], (r & 1) ? selOdd : selEven);
                uint32_t h = A::adds(diag, sc);
                h = A::max(h, E[r]);
                if constexpr (LANESEG) {
                    if (r > 0) h = A::max(h, fseg);
                } else {
                    fseg &= segmask[r];
                    h = A::max(h, fseg);
                }
                const uint32_t t = A::subus(h, a.go);
                E[r] = A::max(A::subus(E[r], a.ge), t);
                const uint32_t hf = A::max(h, ffull);
                fseg = (LANESEG && r == 0) ? t : A::max(A::subus(fseg, a.ge), t);
                ffull = A::max(A::subus(ffull, a.ge), t);
                diag = Hp[r];
                Hp[r] = hf;
                cm = r == 0 ? hf : A::max(cm, hf);      // scores are >= 0
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
        const uint32_t q = (uint32_t) (l * R + row);
        uint64_t key = ((uint64_t) b << 32) | ((uint64_t) (0xffffu - (c & 0xffffu)) << 16) | (uint64_t) (0xffffu - (q & 0xffffu));
        key = groupMaxU64<HL>(key);
        if (l == 0 && (d == 0 ? hasA : hasB)) {
            int32_t *res = a.res0 + (size_t) (pA + d) * 4;
            res[0] = (int32_t) (key >> 32);
            res[1] = (int32_t) (0xffffu - (uint32_t) (key & 0xffffu));
            res[2] = (int32_t) (0xffffu - (uint32_t) ((key >> 16) & 0xffffu));
            res[3] = 1;
        }
    }
}

// One launch serves every query of a batch whose rows-per-lane count lies in [RLO, RLO + 8): a workgroup reads its query's length from
// its descriptor and runs the body instantiated for R = ceil(L / HL).  (One kernel per R made a batch of mixed lengths a dozen small
// launches, each with its own long-target tail; register allocation is that of R = RLO + 7.)
template <bool HAS_AA, int HL, int RLO>
__global__ __launch_bounds__(512, (HL == 16 && RLO == 17) ? 3 : 1) void k_sw3(Sw3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SwBlockDesc bd = a.blocks[blockIdx.x];
    const int R = __builtin_amdgcn_readfirstlane(((int) bd.rowsInTile + HL - 1) / HL);
    switch (R - RLO) {
        case 0: sw3Body<RLO + 0, HAS_AA, HL>(a, bd, smem); break;
        case 1: sw3Body<RLO + 1, HAS_AA, HL>(a, bd, smem); break;
        case 2: sw3Body<RLO + 2, HAS_AA, HL>(a, bd, smem); break;
        case 3: sw3Body<RLO + 3, HAS_AA, HL>(a, bd, smem); break;
        case 4: sw3Body<RLO + 4, HAS_AA, HL>(a, bd, smem); break;
        case 5: sw3Body<RLO + 5, HAS_AA, HL>(a, bd, smem); break;
        case 6: sw3Body<RLO + 6, HAS_AA, HL>(a, bd, smem); break;
        case 7: sw3Body<RLO + 7, HAS_AA, HL>(a, bd, smem); break;
        default: break;
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_sw3_image -- the LDS images of a call's queries, built from the compact query data.
// data of query i at qs[i].dataOff: [q3Di L][qAA L][cb3Di_fwd L][cbAA_fwd L][cb3Di_rev L][cbAA_rev L] (bytes; the rev biases are indexed by
// the position in the reversed query).  Score of (table, letter a, query row i): forward image mat[a][q_i] + cb_fwd_i, reversed image
// mat[a][q_{L-1-i}] + cb_rev_i, 0 for rows beyond the query; row 21 ("past the end of the target") = INT16_MIN in the 3Di table, 0 in the
// AA table (their sum must not wrap).  Output per query: [forward: 3Di table, AA table][reversed: 3Di table, AA table].
// ------------------------------------------------------------------------------------------------------------
struct Sw3ImgQuery {
    uint32_t imgOff;              // dwords
    uint32_t dataOff;             // bytes
    uint32_t L;
    uint16_t R, HL;
};

#ifdef FS_SW3_DEFINE_IMAGE_KERNEL        // one definition: fsgpu_sw3.hip, 3Di-only build
__global__ __launch_bounds__(256) void k_sw3_image(const Sw3ImgQuery *qs, const uint8_t *data, const int8_t *mat3, const int8_t *matA, uint32_t *img, int hasAA) {
    __shared__ int8_t m[2][kAlphabet * kAlphabet];
    for (int i = threadIdx.x; i < kAlphabet * kAlphabet; i += blockDim.x) { m[0][i] = mat3[i]; m[1][i] = hasAA ? matA[i] : (int8_t) 0; }
    __syncthreads();
    const Sw3ImgQuery q = qs[blockIdx.y];
    const int R = q.R, HL = q.HL, L = (int) q.L, D = sw3Dw(R), nt = hasAA ? 2 : 1;
    const int rowDw = sw3RowBytes(R, HL) / 4;
    const int perTable = kSw2Rows * HL * D;
    const int total = 2 * nt * perTable;
    const uint8_t *d = data + q.dataOff;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int dt = idx / perTable;                 // direction * nt + table
        const int dir = dt / nt, tbl = dt - dir * nt;
        int rem = idx - dt * perTable;
        const int a = rem / (HL * D);
        rem -= a * HL * D;
        const int l = rem / D, j = rem - l * D;
        uint32_t v = 0;
        if (a == kAlphabet) v = tbl == 0 ? 0x80008000u : 0u;
        else {
            const uint8_t *codes = d + (size_t) tbl * L;
            const int8_t *cb = (const int8_t *) d + (size_t) (2 + 2 * dir + tbl) * L;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int r = 2 * j + h, row = l * R + r;
                if (r < R && row < L) {
                    const int sc = (int) m[tbl][a * kAlphabet + codes[dir ? L - 1 - row : row]] + (int) cb[row];
                    v |= (uint32_t) (uint16_t) (int16_t) sc << (16 * h);
                }
            }
        }
        uint32_t *dst = img + q.imgOff + (size_t) dt * kSw2Rows * rowDw + (size_t) a * rowDw;
        const int di = sw3DwordIndex(R, HL, l, j);
        dst[di] = v;
        if (j >= sw3Planes16(R) * 4) {                                       // the copies of the remainder plane
            const int nc = sw3RemCopies(R, HL), st = sw3RemCopyStride(R, HL);
            for (int c = 1; c < nc; c++) dst[di + c * st] = v;
        }
    }
}
#endif

} // namespace fs
