// block_aligner.cpp -- C++ restatement of the adaptive-block X-drop aligner that structurealign uses for start
// positions and the backtrace (reference: the Rust crate M/lib/block-aligner, AVX2 configuration, L = 16).
//
// Bit-exact CIGARs depend on saturation corner cases, on tie-breaks between the gap tables and on the adaptive
// shift / grow / shrink trajectory, so this is written as a lane-exact emulation of the crate's vector code:
// every helper below models one of its SIMD helpers on a 16 x int16 value.  File:line citations refer to
// M/lib/block-aligner/src/.
//   align_core        scan_block.rs:120-630     place_block(_3di)   scan_block.rs:1140-1280, 1302-1443
//   Trace / cigar     scan_block.rs:1726-2007   PaddedBytes         scan_block.rs:2149-2245
//   AAMatrix/PosBias  scores.rs:44-160,703-740  Cigar               cigar.rs:24-60
//   vector helpers    avx2.rs                   C entry points      ffi.rs:32-502
// Rust is not available in this image; the crate's own unit-test vectors (scan_block.rs:2337-2413) are ported in
// tests/test_block_aligner.py, and tests/test_oracle_vs_ref.py checks it inside the compiled reference
// (StructureSmithWaterman::alignStartPosBacktraceBlock accepts a trace only if its score equals the SW score).
#include "block_aligner_abi.h"

#include <algorithm>
#include <cassert>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int L = 16;                 // avx2.rs:11
constexpr int16_t ZERO = 1 << 14;     // avx2.rs:15
constexpr int16_t MIN = 0;            // avx2.rs:16
constexpr size_t STEP = 8;            // scan_block.rs:813
constexpr size_t X_DROP_ITER = 2;     // scan_block.rs:814
constexpr bool SHRINK = true;         // scan_block.rs:815
// SHRINK_SUFFIX_LEN = STEP / 4 = 2 (scan_block.rs:816) is folded into suffixHmax2
constexpr uint8_t AA_NULL = 26;       // b'A' + 26 - b'A'

// One 16 x int16 vector.  On x86-64 with AVX2 (the configuration the reference itself is built for) every helper is
// the very instruction the crate's avx2.rs issues; elsewhere the scalar statements below define the same lane values.
#if defined(__AVX2__)
#include <immintrin.h>
struct V {
    __m256i m;
};
inline V set1(int16_t x) { return V{_mm256_set1_epi16(x)}; }
inline V load(const int16_t *p) { return V{_mm256_loadu_si256((const __m256i *) p)}; }
inline void store(int16_t *p, const V &a) { _mm256_storeu_si256((__m256i *) p, a.m); }
inline int16_t lane(const V &a, int i) { alignas(32) int16_t t[16]; _mm256_store_si256((__m256i *) t, a.m); return t[i]; }
inline V withLane0(const V &a, int16_t x) { return V{_mm256_insert_epi16(a.m, x, 0)}; }
inline V adds(const V &a, const V &b) { return V{_mm256_adds_epi16(a.m, b.m)}; }
inline V subs(const V &a, const V &b) { return V{_mm256_subs_epi16(a.m, b.m)}; }
inline V vmax(const V &a, const V &b) { return V{_mm256_max_epi16(a.m, b.m)}; }
inline V cmpeq(const V &a, const V &b) { return V{_mm256_cmpeq_epi16(a.m, b.m)}; }
inline V blend(const V &a, const V &b, const V &mask) { return V{_mm256_blendv_epi8(a.m, b.m, mask.m)}; }
// simd_sl_i16!(a, b, 1) (avx2.rs:92-109)
inline V sl1(const V &a, const V &b) { return V{_mm256_alignr_epi8(a.m, _mm256_permute2x128_si256(a.m, b.m, 0x03), 14)}; }
// simd_step (avx2.rs:131-135)
inline V step8(const V &a, const V &b) { return V{_mm256_permute2x128_si256(a.m, b.m, 0x03)}; }
template <int N> inline V sllz(const V &a) { return V{_mm256_slli_si256(a.m, N * 2)}; }   // simd_sllz_i16! (avx2.rs:143-154)
template <int B> inline V slli16(const V &a) { return V{_mm256_slli_epi16(a.m, B)}; }
inline V broadcasthi(const V &a) { return V{_mm256_permute4x64_epi64(_mm256_shufflehi_epi16(a.m, 0xFF), 0xFF)}; }   // avx2.rs:156-161
inline int16_t hmax(const V &a) {                                                           // avx2.rs:176-184
    __m256i v2 = _mm256_max_epi16(a.m, _mm256_srli_si256(a.m, 2));
    v2 = _mm256_max_epi16(v2, _mm256_srli_si256(v2, 4));
    v2 = _mm256_max_epi16(v2, _mm256_srli_si256(v2, 8));
    v2 = _mm256_max_epi16(v2, _mm256_permute2x128_si256(v2, v2, 0x03));
    return (int16_t) _mm256_extract_epi16(v2, 0);
}
struct ScanConsts { V gapExtendAll, lane; };
inline ScanConsts prefixScanConsts(const V &gap) {                                          // avx2.rs:283-297
    __m256i shift1 = _mm256_adds_epi16(_mm256_slli_si256(gap.m, 2), gap.m);
    __m256i shift2 = _mm256_adds_epi16(_mm256_slli_si256(shift1, 4), shift1);
    __m256i shift4 = _mm256_adds_epi16(_mm256_slli_si256(shift2, 8), shift2);
    __m256i correct1 = _mm256_srli_si256(_mm256_shufflehi_epi16(shift4, 0xFF), 8);
    correct1 = _mm256_permute4x64_epi64(correct1, 0x05);
    correct1 = _mm256_adds_epi16(correct1, shift4);
    ScanConsts c; c.gapExtendAll = V{correct1}; c.lane = V{shift4};
    return c;
}
inline V prefixScan(const V &Rmax, const V &gapCost, const V &gapCostLane) {                // avx2.rs:299-317
    __m256i shift1 = _mm256_adds_epi16(_mm256_slli_si256(Rmax.m, 2), gapCost.m);
    shift1 = _mm256_max_epi16(Rmax.m, shift1);
    __m256i shift2 = _mm256_adds_epi16(_mm256_slli_si256(shift1, 4), _mm256_slli_epi16(gapCost.m, 1));
    shift2 = _mm256_max_epi16(shift1, shift2);
    __m256i shift4 = _mm256_adds_epi16(_mm256_slli_si256(shift2, 8), _mm256_slli_epi16(gapCost.m, 2));
    shift4 = _mm256_max_epi16(shift2, shift4);
    __m256i correct1 = _mm256_permute4x64_epi64(_mm256_shufflehi_epi16(shift4, 0xFF), 0x50);
    correct1 = _mm256_adds_epi16(correct1, gapCostLane.m);
    return V{_mm256_max_epi16(shift4, correct1)};
}
// simd_movemask_i8(simd_blend_i8(lo, hi, 0xFF00)): bit 2k <- lo lane k, bit 2k+1 <- hi lane k
inline uint32_t mask2(const V &lo, const V &hi) {
    return (uint32_t) _mm256_movemask_epi8(_mm256_blendv_epi8(lo.m, hi.m, _mm256_set1_epi16((int16_t) 0xFF00)));
}
// halfsimd_lookup2_i16 (avx2.rs:319-326): 32-entry int8 row looked up with two pshufb + blend, sign extended
inline V lookup32(const int8_t *row, const uint8_t *q) {
    const __m128i lut1 = _mm_loadu_si128((const __m128i *) row), lut2 = _mm_loadu_si128((const __m128i *) (row + 16));
    const __m128i v = _mm_loadu_si128((const __m128i *) q);
    const __m128i a = _mm_shuffle_epi8(lut1, v), b = _mm_shuffle_epi8(lut2, v);
    const __m128i c = _mm_blendv_epi8(a, b, _mm_slli_epi16(v, 3));
    return V{_mm256_cvtepi8_epi16(c)};
}
#else
struct V {
    int16_t v[L];
};
inline int16_t sat16(int32_t x) { return (int16_t) (x > 32767 ? 32767 : (x < -32768 ? -32768 : x)); }
inline V set1(int16_t x) { V r; for (int i = 0; i < L; i++) r.v[i] = x; return r; }
inline V load(const int16_t *p) { V r; memcpy(r.v, p, sizeof(r.v)); return r; }
inline void store(int16_t *p, const V &a) { memcpy(p, a.v, sizeof(a.v)); }
inline int16_t lane(const V &a, int i) { return a.v[i]; }
inline V withLane0(const V &a, int16_t x) { V r = a; r.v[0] = x; return r; }
inline V adds(const V &a, const V &b) { V r; for (int i = 0; i < L; i++) r.v[i] = sat16((int32_t) a.v[i] + b.v[i]); return r; }
inline V subs(const V &a, const V &b) { V r; for (int i = 0; i < L; i++) r.v[i] = sat16((int32_t) a.v[i] - b.v[i]); return r; }
inline V vmax(const V &a, const V &b) { V r; for (int i = 0; i < L; i++) r.v[i] = a.v[i] > b.v[i] ? a.v[i] : b.v[i]; return r; }
inline V cmpeq(const V &a, const V &b) { V r; for (int i = 0; i < L; i++) r.v[i] = a.v[i] == b.v[i] ? (int16_t) -1 : (int16_t) 0; return r; }
inline V blend(const V &a, const V &b, const V &mask) { V r; for (int i = 0; i < L; i++) r.v[i] = mask.v[i] ? b.v[i] : a.v[i]; return r; }
inline V sl1(const V &a, const V &b) { V r; r.v[0] = b.v[L - 1]; for (int i = 1; i < L; i++) r.v[i] = a.v[i - 1]; return r; }
inline V step8(const V &a, const V &b) { V r; for (int i = 0; i < 8; i++) { r.v[i] = b.v[8 + i]; r.v[8 + i] = a.v[i]; } return r; }
// _mm256_slli_si256 by N lanes: shifts inside each 128-bit half, zero fill
template <int N> inline V sllz(const V &a) {
    V r;
    for (int h = 0; h < 2; h++)
        for (int k = 0; k < 8; k++) r.v[h * 8 + k] = k >= N ? a.v[h * 8 + k - N] : (int16_t) 0;
    return r;
}
template <int B> inline V slli16(const V &a) { V r; for (int i = 0; i < L; i++) r.v[i] = (int16_t) ((uint16_t) a.v[i] << B); return r; }
inline V broadcasthi(const V &a) { return set1(a.v[L - 1]); }
inline int16_t hmax(const V &a) { int16_t m = a.v[0]; for (int i = 1; i < L; i++) m = std::max(m, a.v[i]); return m; }
struct ScanConsts { V gapExtendAll, lane; };
inline ScanConsts prefixScanConsts(const V &gap) {
    V shift1 = adds(sllz<1>(gap), gap);
    V shift2 = adds(sllz<2>(shift1), shift1);
    V shift4 = adds(sllz<4>(shift2), shift2);
    V correct1;
    for (int k = 0; k < 8; k++) { correct1.v[k] = 0; correct1.v[8 + k] = shift4.v[7]; }
    correct1 = adds(correct1, shift4);
    ScanConsts c; c.gapExtendAll = correct1; c.lane = shift4;
    return c;
}
inline V prefixScan(const V &Rmax, const V &gapCost, const V &gapCostLane) {
    V shift1 = vmax(Rmax, adds(sllz<1>(Rmax), gapCost));
    V shift2 = vmax(shift1, adds(sllz<2>(shift1), slli16<1>(gapCost)));
    V shift4 = vmax(shift2, adds(sllz<4>(shift2), slli16<2>(gapCost)));
    V correct1;
    for (int k = 0; k < 4; k++) { correct1.v[k] = shift4.v[k]; correct1.v[4 + k] = shift4.v[k]; }
    for (int k = 0; k < 8; k++) correct1.v[8 + k] = shift4.v[7];
    correct1 = adds(correct1, gapCostLane);
    return vmax(shift4, correct1);
}
inline uint32_t mask2(const V &lo, const V &hi) {
    uint32_t m = 0;
    for (int k = 0; k < L; k++) {
        if (lo.v[k]) m |= 1u << (2 * k);
        if (hi.v[k]) m |= 1u << (2 * k + 1);
    }
    return m;
}
inline V lookup32(const int8_t *row, const uint8_t *q) {
    V r;
    for (int k = 0; k < L; k++) {
        const uint8_t b = q[k];
        r.v[k] = (b & 0x80) ? 0 : row[(b & 0x0f) + ((b & 0x10) ? 16 : 0)];
    }
    return r;
}
#endif
// simd_prefix_hmax_i16!(v, STEP = 8): maximum of lanes 0..7; the shifted-in zeros only touch lanes that are not read
inline int16_t prefixHmax8(const int16_t *p) { int16_t m = p[0]; for (int i = 1; i < 8; i++) m = std::max(m, p[i]); return m; }
// simd_suffix_hmax_i16!(v, 2): max of the two top lanes (avx2.rs:232-256)
inline int16_t suffixHmax2(const int16_t *p) { return std::max(p[L - 1], p[L - 2]); }

} // namespace

// ---- opaque ABI types ----------------------------------------------------------------------------------------
struct AAMatrix { int8_t scores[27 * 32]; };
struct PaddedBytes { std::vector<uint8_t> s; size_t len; };
struct PosBias { std::vector<int16_t> bias; size_t len; };
struct Cigar { std::vector<OpLen> s; size_t idx; };

namespace {

inline uint8_t upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t) (c - 32) : c; }
inline uint8_t convertChar(uint8_t c) { c = upper(c); assert(c >= 'A' && c <= 'A' + 26); return (uint8_t) (c - 'A'); }

// AAMatrix::get_scores (scores.rs:126-140): the 32-entry row of c, looked up per query byte, sign extended
inline V getScores(const AAMatrix *m, uint8_t c, const uint8_t *q) { return lookup32(m->scores + (size_t) c * 32, q); }

struct Trace {
    std::vector<uint32_t> trace, trace2;
    std::vector<uint64_t> right;
    std::vector<uint32_t> blockStart;
    std::vector<uint16_t> blockSize;
    size_t traceIdx = 0, blockIdx = 0, ckptTraceIdx = 0, ckptBlockIdx = 0, queryLen = 0, referenceLen = 0;
    void init(size_t q, size_t r, size_t maxSize) {
        const size_t len = q + r;
        trace.assign((maxSize / L) * (len + maxSize * 2), 0);
        trace2.assign((maxSize / L) * (len + maxSize * 2), 0);
        right.assign((len + 63) / 64 + 1, 0);
        blockStart.assign(len * 2 + 4, 0);
        blockSize.assign(len * 2 + 4, 0);
        queryLen = q; referenceLen = r;
    }
    void clear(size_t q, size_t r) {
        std::fill(right.begin(), right.end(), 0);
        traceIdx = blockIdx = ckptTraceIdx = ckptBlockIdx = 0;
        queryLen = q; referenceLen = r;
    }
    inline void addTrace(uint32_t t, uint32_t t2) { trace[traceIdx] = t; trace2[traceIdx] = t2; traceIdx++; }
    inline void addBlock(size_t i, size_t j, size_t width, size_t height, bool r) {
        blockStart[blockIdx * 2] = (uint32_t) i;
        blockStart[blockIdx * 2 + 1] = (uint32_t) j;
        blockSize[blockIdx * 2] = (uint16_t) height;
        blockSize[blockIdx * 2 + 1] = (uint16_t) width;
        const size_t a = blockIdx / 64, b = blockIdx % 64;
        right[a] = (right[a] & ~(1ull << b)) | ((uint64_t) r << b);
        blockIdx++;
    }
    inline void saveCkpt() { ckptTraceIdx = traceIdx; ckptBlockIdx = blockIdx; }
    inline void restoreCkpt() { traceIdx = ckptTraceIdx; blockIdx = ckptBlockIdx; }
};

enum Dir { DirRight, DirDown, DirGrow };

struct Seq {                       // PaddedBytes3di / plain PaddedBytes view
    const PaddedBytes *bytes;
    const PaddedBytes *bytes3di;   // may be NULL (single-matrix alignment)
    const PosBias *bias;           // may be NULL
    size_t len() const { return bytes->len; }
};

struct Block {
    bool traceOn, xdrop;
    AlignResult res{0, 0, 0};
    Trace trace;
    std::vector<int16_t> Dcol, Ccol, Drow, Rrow, DcolCk, CcolCk, DrowCk, RrowCk, tmp1, tmp2;
    size_t queryLenCap = 0, referenceLenCap = 0, maxSizeCap = 0;

    Block(size_t q, size_t r, size_t maxSize, bool t, bool x) : traceOn(t), xdrop(x) {
        assert(maxSize && (maxSize & (maxSize - 1)) == 0);
        if (t) trace.init(q, r, maxSize); else trace.init(0, 0, 0);
        for (auto *v : {&Dcol, &Ccol, &Drow, &Rrow, &DcolCk, &CcolCk, &DrowCk, &RrowCk}) v->assign(maxSize + L, MIN);
        tmp1.assign(L, MIN); tmp2.assign(L, MIN);
        queryLenCap = q; referenceLenCap = r; maxSizeCap = maxSize;
    }
    void clear(size_t q, size_t r, size_t maxSize) {
        assert(q + r <= queryLenCap + referenceLenCap && maxSize <= maxSizeCap);
        trace.clear(q, r);
        for (auto *v : {&Dcol, &Ccol, &Drow, &Rrow, &DcolCk, &CcolCk, &DrowCk, &RrowCk}) std::fill(v->begin(), v->begin() + maxSize, MIN);
        std::fill(tmp1.begin(), tmp1.end(), MIN);
        std::fill(tmp2.begin(), tmp2.end(), MIN);
    }

    struct PB { V Dmax, argI, argJ; };

    // place_block / place_block_3di (scan_block.rs:1140-1280 / 1302-1443); `query` is the sequence along the vector
    // dimension, `reference` the one walked one residue per outer iteration (the caller swaps them for down shifts)
    PB placeBlock(const Seq &query, const Seq &reference, const AAMatrix *m, const AAMatrix *m3di, const Gaps &gaps, size_t startI,
                  size_t startJ, size_t width, size_t height, int16_t *DcolP, int16_t *CcolP, int16_t *DrowP, int16_t *RrowP, V Dcorner,
                  bool /*right*/) {
        const V gapOpen = set1(gaps.open), gapExtend = set1(gaps.extend);
        const ScanConsts sc = prefixScanConsts(gapExtend);
        PB out; out.Dmax = set1(MIN); out.argI = set1(0); out.argJ = set1(0);
        if (width == 0 || height == 0) return out;
        const V openMinusExt = subs(gapOpen, gapExtend);
        for (size_t j = 0; j < width; j++) {
            V R01 = set1(MIN), D11 = set1(MIN), R11 = set1(MIN), prevTraceR = set1(0);
            const uint8_t c = reference.bytes->s[startJ + j];
            const uint8_t c3 = reference.bytes3di ? reference.bytes3di->s[startJ + j] : 0;
            const V refBias = set1(reference.bias ? reference.bias->bias[startJ + j] : 0);
            for (size_t i = 0; i < height; i += L) {
                const V D10 = load(DcolP + i), C10 = load(CcolP + i);
                const V D00 = sl1(D10, Dcorner);
                Dcorner = D10;
                V scores = getScores(m, c, query.bytes->s.data() + startI + i);
                if (m3di) {
                    const V s3 = getScores(m3di, c3, query.bytes3di->s.data() + startI + i);
                    const V qBias = load(query.bias->bias.data() + startI + i);
                    scores = adds(adds(scores, s3), adds(refBias, qBias));
                }
                D11 = adds(D00, scores);
                if (startI + i == 0 && startJ + j == 0) D11 = withLane0(D11, ZERO);
                const V C11open = adds(D10, gapOpen);
                const V C11 = vmax(adds(C10, gapExtend), C11open);
                D11 = vmax(D11, C11);
                const V D11open = adds(D11, openMinusExt);
                R11 = prefixScan(D11open, gapExtend, sc.lane);
                R11 = vmax(R11, adds(broadcasthi(R01), sc.gapExtendAll));
                D11 = vmax(D11, R11);
                R01 = R11;
                if (traceOn) {
                    const uint32_t t = mask2(cmpeq(D11, C11), cmpeq(D11, R11));
                    const V tempTraceR = cmpeq(R11, D11open);
                    const V traceR = sl1(tempTraceR, prevTraceR);
                    const uint32_t t2 = mask2(cmpeq(C11, C11open), traceR);
                    prevTraceR = tempTraceR;
                    trace.addTrace(t, t2);
                }
                out.Dmax = vmax(out.Dmax, D11);
                if (xdrop) {
                    const V mk = cmpeq(out.Dmax, D11);
                    out.argI = blend(out.argI, set1((int16_t) i), mk);
                    out.argJ = blend(out.argJ, set1((int16_t) j), mk);
                }
                store(DcolP + i, D11);
                store(CcolP + i, C11);
            }
            Dcorner = set1(MIN);
            DrowP[j] = lane(D11, L - 1);
            RrowP[j] = lane(R11, L - 1);
            if (!xdrop && startI + height > query.len() && startJ + j >= reference.len()) {
                if (traceOn) trace.traceIdx += (width - 1 - j) * (height / L);
                break;
            }
        }
        return out;
    }

    static void justOffset(size_t blockSize, int16_t *b1, int16_t *b2, const V &offAdd) {
        for (size_t i = 0; i < blockSize; i += L) { store(b1 + i, adds(load(b1 + i), offAdd)); store(b2 + i, adds(load(b2 + i), offAdd)); }
    }
    // shift_and_offset (scan_block.rs:1096-1123): drop the first STEP entries, append temp buffers, re-base
    static V shiftAndOffset(size_t blockSize, int16_t *b1, int16_t *b2, const int16_t *t1, const int16_t *t2, const V &offAdd) {
        V curr1 = adds(load(b1), offAdd);
        const V Dcorner = set1(lane(curr1, STEP - 1));
        V curr2 = adds(load(b2), offAdd);
        size_t i = 0;
        while (i < blockSize - L) {
            const V next1 = adds(load(b1 + i + L), offAdd), next2 = adds(load(b2 + i + L), offAdd);
            store(b1 + i, step8(next1, curr1));
            store(b2 + i, step8(next2, curr2));
            curr1 = next1; curr2 = next2;
            i += L;
        }
        store(b1 + blockSize - L, step8(load(t1), curr1));
        store(b2 + blockSize - L, step8(load(t2), curr2));
        return Dcorner;
    }
    static int16_t clamp16(int32_t x) { return (int16_t) (x > 32767 ? 32767 : (x < -32768 ? -32768 : x)); }

    // align_core (scan_block.rs:120-630)
    void align(const Seq &query, const Seq &reference, const AAMatrix *m, const AAMatrix *m3di, Gaps gaps, size_t minSize, size_t maxSize,
               int32_t xDropThr) {
        if (minSize < (size_t) L) minSize = L;
        if (maxSize < (size_t) L) maxSize = L;
        clear(query.len(), reference.len(), maxSize);
        size_t si = 0, sj = 0;
        int32_t bestMax = 0;
        size_t bestArgI = 0, bestArgJ = 0;
        Dir prevDir = DirGrow, dir = DirGrow;
        size_t prevSize = 0, blockSize = minSize;
        int32_t off = 0, prevOff, offMax = 0;
        size_t yDropIter = 0, xDropIter = 0;
        size_t iCkpt = si, jCkpt = sj;
        int32_t offCkpt = 0;
        V Dcorner = set1(MIN);
        auto copyToCkpt = [&](size_t n) {
            memcpy(DcolCk.data(), Dcol.data(), n * 2); memcpy(CcolCk.data(), Ccol.data(), n * 2);
            memcpy(DrowCk.data(), Drow.data(), n * 2); memcpy(RrowCk.data(), Rrow.data(), n * 2);
        };
        for (;;) {
            prevOff = off;
            V growDmax = set1(MIN), growArgI = set1(0), growArgJ = set1(0);
            PB pb;
            int16_t rightMax, downMax;
            if (dir == DirRight) {
                off = offMax;
                const V offAdd = set1(clamp16(prevOff - off));
                if (traceOn) trace.addBlock(si, sj + blockSize - STEP, STEP, blockSize, true);
                justOffset(blockSize, Dcol.data(), Ccol.data(), offAdd);
                pb = placeBlock(query, reference, m, m3di, gaps, si, sj + blockSize - STEP, STEP, blockSize, Dcol.data(), Ccol.data(), tmp1.data(),
                                tmp2.data(), prevDir == DirDown ? adds(Dcorner, offAdd) : set1(MIN), true);
                rightMax = prefixHmax8(Dcol.data());
                Dcorner = shiftAndOffset(blockSize, Drow.data(), Rrow.data(), tmp1.data(), tmp2.data(), offAdd);
                downMax = prefixHmax8(Drow.data());
            } else if (dir == DirDown) {
                off = offMax;
                const V offAdd = set1(clamp16(prevOff - off));
                if (traceOn) trace.addBlock(si + blockSize - STEP, sj, blockSize, STEP, false);
                justOffset(blockSize, Drow.data(), Rrow.data(), offAdd);
                pb = placeBlock(reference, query, m, m3di, gaps, sj, si + blockSize - STEP, STEP, blockSize, Drow.data(), Rrow.data(), tmp1.data(),
                                tmp2.data(), prevDir == DirRight ? adds(Dcorner, offAdd) : set1(MIN), false);
                downMax = prefixHmax8(Drow.data());
                Dcorner = shiftAndOffset(blockSize, Dcol.data(), Ccol.data(), tmp1.data(), tmp2.data(), offAdd);
                rightMax = prefixHmax8(Dcol.data());
            } else {
                Dcorner = set1(MIN);
                const size_t growStep = blockSize - prevSize;
                if (traceOn) trace.addBlock(si + prevSize, sj, prevSize, growStep, false);
                const PB p1 = placeBlock(reference, query, m, m3di, gaps, sj, si + prevSize, growStep, prevSize, Drow.data(), Rrow.data(),
                                         Dcol.data() + prevSize, Ccol.data() + prevSize, set1(MIN), false);
                if (traceOn) trace.addBlock(si, sj + prevSize, growStep, blockSize, true);
                pb = placeBlock(query, reference, m, m3di, gaps, si, sj + prevSize, growStep, blockSize, Dcol.data(), Ccol.data(),
                                Drow.data() + prevSize, Rrow.data() + prevSize, set1(MIN), true);
                rightMax = prefixHmax8(Dcol.data());
                downMax = prefixHmax8(Drow.data());
                growDmax = p1.Dmax; growArgI = p1.argI; growArgJ = p1.argJ;
                copyToCkpt(blockSize);
                if (traceOn) trace.saveCkpt();
            }
            prevDir = dir;
            const int16_t DmaxMax = hmax(pb.Dmax), growMax = hmax(growDmax);
            const int16_t mx = std::max(DmaxMax, growMax);
            offMax = off + (int32_t) mx - (int32_t) ZERO;
            yDropIter++;
            bool growNoMax = dir == DirGrow;
            if (offMax > bestMax) {
                if (xdrop) {
                    size_t bestI = 0, bestJ = 0;
                    const bool grow = dir == DirGrow && DmaxMax < growMax;
                    const int16_t currMax = grow ? growMax : DmaxMax;
                    const V &cd = grow ? growDmax : pb.Dmax;
                    const V &ci = grow ? growArgI : pb.argI;
                    const V &cj = grow ? growArgJ : pb.argJ;
                    alignas(32) int16_t cdv[L], civ[L], cjv[L];
                    store(cdv, cd); store(civ, ci); store(cjv, cj);
                    for (int ln = 0; ln < L; ln++) {
                        if (cdv[ln] != currMax) continue;
                        const size_t idxI = (size_t) (uint16_t) civ[ln], idxJ = (size_t) (uint16_t) cjv[ln];
                        const size_t r = idxI + ln, c = (blockSize - STEP) + idxJ;
                        size_t gi, gj;
                        if (grow) { gi = si + prevSize + idxJ; gj = sj + idxI + ln; }
                        else if (dir == DirRight) { gi = si + r; gj = sj + c; }
                        else if (dir == DirDown) { gi = si + c; gj = sj + r; }
                        else { gi = si + idxI + ln; gj = sj + prevSize + idxJ; }
                        const bool better = (gj != bestJ) ? (gj > bestJ) : (gi > bestI);
                        if (better) { bestI = gi; bestJ = gj; }
                    }
                    bestArgI = bestI; bestArgJ = bestJ;
                }
                if (blockSize < maxSize) {
                    iCkpt = si; jCkpt = sj; offCkpt = off;
                    copyToCkpt(blockSize);
                    if (traceOn) trace.saveCkpt();
                    growNoMax = false;
                }
                bestMax = offMax;
                yDropIter = 0;
            }
            if (xdrop) {
                if (offMax < bestMax - xDropThr) {
                    if (xDropIter < X_DROP_ITER - 1) xDropIter++;
                    else break;
                } else {
                    xDropIter = 0;
                }
            }
            if (si + blockSize > query.len() && sj + blockSize > reference.len()) break;
            if (sj + blockSize > reference.len()) { si += STEP; dir = DirDown; continue; }
            if (si + blockSize > query.len()) { sj += STEP; dir = DirRight; continue; }
            const size_t nextSize = blockSize * 2;
            if (nextSize <= maxSize) {
                if (yDropIter > (blockSize / STEP) - 1 || growNoMax) {
                    prevSize = blockSize;
                    blockSize = nextSize;
                    dir = DirGrow;
                    si = iCkpt; sj = jCkpt; off = offCkpt;
                    memcpy(Dcol.data(), DcolCk.data(), prevSize * 2); memcpy(Ccol.data(), CcolCk.data(), prevSize * 2);
                    memcpy(Drow.data(), DrowCk.data(), prevSize * 2); memcpy(Rrow.data(), RrowCk.data(), prevSize * 2);
                    if (traceOn) trace.restoreCkpt();
                    yDropIter = 0;
                    continue;
                }
            }
            if (SHRINK && blockSize > minSize && yDropIter == 0) {
                const int16_t shrinkMax = std::max(suffixHmax2(Drow.data() + blockSize - L), suffixHmax2(Dcol.data() + blockSize - L));
                if (shrinkMax >= mx) {
                    prevDir = DirGrow;
                    blockSize /= 2;
                    for (size_t i = 0; i < blockSize; i += L) {
                        memcpy(Dcol.data() + i, Dcol.data() + i + blockSize, L * 2); memcpy(Ccol.data() + i, Ccol.data() + i + blockSize, L * 2);
                        memcpy(Drow.data() + i, Drow.data() + i + blockSize, L * 2); memcpy(Rrow.data() + i, Rrow.data() + i + blockSize, L * 2);
                    }
                    si += blockSize; sj += blockSize;
                    iCkpt = si; jCkpt = sj; offCkpt = off;
                    copyToCkpt(blockSize);
                    rightMax = prefixHmax8(Dcol.data());
                    downMax = prefixHmax8(Drow.data());
                    if (traceOn) trace.saveCkpt();
                    yDropIter = 0;
                }
            }
            if (downMax > rightMax) { si += STEP; dir = DirDown; }
            else { sj += STEP; dir = DirRight; }
        }
        if (xdrop) {
            res.score = bestMax; res.query_idx = bestArgI; res.reference_idx = bestArgJ;
        } else {
            int32_t score;
            if (dir == DirRight || dir == DirGrow) score = off + (int32_t) Dcol[query.len() - si] - (int32_t) ZERO;
            else score = off + (int32_t) Drow[reference.len() - sj] - (int32_t) ZERO;
            res.score = score; res.query_idx = query.len(); res.reference_idx = reference.len();
        }
    }

    // Trace::cigar_core (scan_block.rs:1844-2007)
    void cigar(size_t i, size_t j, const PaddedBytes *q, const PaddedBytes *r, bool eq, Cigar *cg) const {
        assert(i <= trace.queryLen && j <= trace.referenceLen);
        const size_t need = i + j + 5;
        if (cg->s.size() < need) cg->s.resize(need);
        for (size_t k = 0; k < need; k++) { cg->s[k].op = BA_Sentinel; cg->s[k].len = 0; }
        cg->idx = 1;
        enum Table { TD = 0, TC = 1, TR = 2 };
        struct Ent { uint8_t op; uint8_t di, dj; uint8_t table; };
        static Ent lut[2][64];
        static bool lutInit = false;
        if (!lutInit) {
            for (int rt = 0; rt < 2; rt++)
                for (int t = 0; t < 4; t++)
                    for (int t2 = 0; t2 < 4; t2++)
                        for (int tb = 0; tb < 3; tb++) {
                            Ent e;
                            const bool t2b0 = t2 & 1, t2b1 = t2 & 2;
                            if (rt == 1) {
                                if (tb == TC) e = t2b0 ? Ent{BA_D, 0, 1, TD} : Ent{BA_D, 0, 1, TC};
                                else if (tb == TR) e = t2b1 ? Ent{BA_I, 1, 0, TD} : Ent{BA_I, 1, 0, TR};
                                else if (t == 0) e = Ent{BA_M, 1, 1, TD};
                                else if (t == 1 || t == 3) e = t2b0 ? Ent{BA_D, 0, 1, TD} : Ent{BA_D, 0, 1, TC};
                                else e = t2b1 ? Ent{BA_I, 1, 0, TD} : Ent{BA_I, 1, 0, TR};
                            } else {
                                if (tb == TR) e = t2b0 ? Ent{BA_I, 1, 0, TD} : Ent{BA_I, 1, 0, TR};
                                else if (tb == TC) e = t2b1 ? Ent{BA_D, 0, 1, TD} : Ent{BA_D, 0, 1, TC};
                                else if (t == 0) e = Ent{BA_M, 1, 1, TD};
                                else if (t == 1 || t == 3) e = t2b0 ? Ent{BA_I, 1, 0, TD} : Ent{BA_I, 1, 0, TR};
                                else e = t2b1 ? Ent{BA_D, 0, 1, TD} : Ent{BA_D, 0, 1, TC};
                            }
                            lut[rt][(t << 4) | (t2 << 2) | tb] = e;
                        }
            lutInit = true;
        }
        size_t blockIdx = trace.blockIdx, traceIdx = trace.traceIdx;
        size_t blockI = 0, blockJ = 0, blockW = 0, blockH = 0;
        int rightFlag = 0;
        int table = TD;
        auto add = [&](uint8_t op) {
            const size_t addn = op != cg->s[cg->idx - 1].op;
            cg->idx += addn;
            cg->s[cg->idx - 1].op = op;
            cg->s[cg->idx - 1].len += 1;
        };
        while (i > 0 || j > 0) {
            for (;;) {
                blockIdx--;
                blockI = trace.blockStart[blockIdx * 2];
                blockJ = trace.blockStart[blockIdx * 2 + 1];
                blockH = trace.blockSize[blockIdx * 2];
                blockW = trace.blockSize[blockIdx * 2 + 1];
                traceIdx -= blockW * blockH / L;
                if (i >= blockI && j >= blockJ) {
                    rightFlag = (int) ((trace.right[blockIdx / 64] >> (blockIdx % 64)) & 1);
                    break;
                }
            }
            const Ent *lt = lut[rightFlag];
            while (i >= blockI && j >= blockJ && (i > 0 || j > 0)) {
                const size_t ci = i - blockI, cj = j - blockJ;
                size_t idx, sh;
                if (rightFlag) { idx = traceIdx + ci / L + cj * (blockH / L); sh = (ci % L) * 2; }
                else { idx = traceIdx + cj / L + ci * (blockW / L); sh = (cj % L) * 2; }
                const unsigned t = (trace.trace[idx] >> sh) & 3u, t2 = (trace.trace2[idx] >> sh) & 3u;
                const Ent &e = lt[(t << 4) | (t2 << 2) | (unsigned) table];
                uint8_t op = e.op;
                if (eq && op == BA_M) op = (q->s[i] == r->s[j]) ? BA_Eq : BA_X;
                i -= e.di; j -= e.dj; table = e.table;
                add(op);
            }
        }
    }
};

} // namespace

extern "C" {

AAMatrix *block_new_simple_aamatrix(int8_t matchScore, int8_t mismatchScore) {
    AAMatrix *m = new AAMatrix();
    for (int i = 0; i < 27 * 32; i++) m->scores[i] = INT8_MIN;
    for (int i = 0; i < 26; i++)
        for (int j = 0; j < 26; j++) m->scores[i * 32 + j] = i == j ? matchScore : mismatchScore;
    return m;
}
void block_set_aamatrix(AAMatrix *m, uint8_t a, uint8_t b, int8_t score) {
    a = upper(a); b = upper(b);
    assert(a >= 'A' && a <= 'Z' + 1 && b >= 'A' && b <= 'Z' + 1);
    m->scores[(size_t) (a - 'A') * 32 + (b - 'A')] = score;
    m->scores[(size_t) (b - 'A') * 32 + (a - 'A')] = score;
}
void block_set_aamatrix_num(AAMatrix *m, int8_t a, int8_t b, int8_t score) {
    m->scores[(size_t) (uint8_t) a * 32 + (uint8_t) b] = score;
    m->scores[(size_t) (uint8_t) b * 32 + (uint8_t) a] = score;
}
void block_free_aamatrix(AAMatrix *m) { delete m; }
const int8_t *block_aamatrix_scores(const AAMatrix *m) { return m->scores; }

Cigar *block_new_cigar(uintptr_t q, uintptr_t r) {
    Cigar *c = new Cigar();
    c->s.assign(q + r + 5, OpLen{BA_Sentinel, 0});
    c->idx = 1;
    return c;
}
OpLen block_get_cigar(const Cigar *c, uintptr_t i) { return c->s[c->idx - 1 - i]; }
uintptr_t block_len_cigar(const Cigar *c) { return c->idx - 1; }
void block_free_cigar(Cigar *c) { delete c; }

PaddedBytes *block_new_padded_aa(uintptr_t len, uintptr_t maxSize) {
    PaddedBytes *p = new PaddedBytes();
    p->s.assign(1 + len + maxSize + L, AA_NULL);
    p->len = len;
    return p;
}
void block_set_bytes_padded_aa(PaddedBytes *p, const uint8_t *s, uintptr_t len, uintptr_t maxSize) {
    if (p->s.size() < 1 + len + maxSize + L) p->s.resize(1 + len + maxSize + L, AA_NULL);
    p->s[0] = AA_NULL;
    for (size_t i = 0; i < len; i++) p->s[1 + i] = convertChar(s[i]);
    std::fill(p->s.begin() + 1 + len, p->s.begin() + 1 + len + maxSize, AA_NULL);
    p->len = len;
}
void block_free_padded_aa(PaddedBytes *p) { delete p; }

PosBias *block_new_pos_bias(uintptr_t len, uintptr_t maxSize) {
    PosBias *b = new PosBias();
    b->bias.assign(len + maxSize + 1 + L, 0);
    b->len = len;
    return b;
}
void block_set_pos_bias(PosBias *b, const int16_t *v, uintptr_t len) {
    if (b->bias.size() < len + 1 + L) b->bias.resize(len + 1 + L, 0);
    std::fill(b->bias.begin(), b->bias.end(), 0);        // PosBias::set_biases zero-fills the whole vector (scores.rs:721-725)
    memcpy(b->bias.data() + 1, v, len * sizeof(int16_t));
    b->len = len;
}
void block_free_pos_bias(PosBias *b) { delete b; }

BlockHandle block_new_aa_trace_xdrop(uintptr_t q, uintptr_t r, uintptr_t maxSize) { return new Block(q, r, maxSize, true, true); }
void block_align_3di_aa_trace_xdrop(BlockHandle b, const PaddedBytes *q, const PaddedBytes *q3, const PosBias *qb, const PaddedBytes *r,
                                    const PaddedBytes *r3, const PosBias *rb, const AAMatrix *m, const AAMatrix *m3, Gaps g, SizeRange s, int32_t x) {
    assert(q->len == q3->len && q->len == qb->len && r->len == r3->len && r->len == rb->len);
    assert(g.open < 0 && g.extend < 0 && g.open < g.extend && x >= 0);
    Seq qs{q, q3, qb}, rs{r, r3, rb};
    ((Block *) b)->align(qs, rs, m, m3, g, s.min, s.max, x);
}
void block_align_aa_trace_xdrop(BlockHandle b, const PaddedBytes *q, const PaddedBytes *r, const AAMatrix *m, Gaps g, SizeRange s, int32_t x) {
    Seq qs{q, nullptr, nullptr}, rs{r, nullptr, nullptr};
    ((Block *) b)->align(qs, rs, m, nullptr, g, s.min, s.max, x);
}
AlignResult block_res_aa_trace_xdrop(BlockHandle b) { return ((Block *) b)->res; }
void block_cigar_aa_trace_xdrop(BlockHandle b, uintptr_t qi, uintptr_t ri, Cigar *c) { ((Block *) b)->cigar(qi, ri, nullptr, nullptr, false, c); }
void block_cigar_eq_aa_trace_xdrop(BlockHandle b, const PaddedBytes *q, const PaddedBytes *r, uintptr_t qi, uintptr_t ri, Cigar *c) {
    ((Block *) b)->cigar(qi, ri, q, r, true, c);
}
void block_free_aa_trace_xdrop(BlockHandle b) { delete (Block *) b; }

BlockHandle block_new_aa_trace(uintptr_t q, uintptr_t r, uintptr_t maxSize) { return new Block(q, r, maxSize, true, false); }
void block_align_aa_trace(BlockHandle b, const PaddedBytes *q, const PaddedBytes *r, const AAMatrix *m, Gaps g, SizeRange s, int32_t x) {
    Seq qs{q, nullptr, nullptr}, rs{r, nullptr, nullptr};
    ((Block *) b)->align(qs, rs, m, nullptr, g, s.min, s.max, x);
}
AlignResult block_res_aa_trace(BlockHandle b) { return ((Block *) b)->res; }
void block_cigar_aa_trace(BlockHandle b, uintptr_t qi, uintptr_t ri, Cigar *c) { ((Block *) b)->cigar(qi, ri, nullptr, nullptr, false, c); }
void block_cigar_eq_aa_trace(BlockHandle b, const PaddedBytes *q, const PaddedBytes *r, uintptr_t qi, uintptr_t ri, Cigar *c) {
    ((Block *) b)->cigar(qi, ri, q, r, true, c);
}
void block_free_aa_trace(BlockHandle b) { delete (Block *) b; }

// The regions the last alignment computed, in the order the trace stack holds them (the crate's Trace::blocks(), scan_block.rs:2009-2030;
// a Rust API the crate's C header does not export): out[5 k ..] = {row, column, height, width, right} of region k.  Returns their number.
// For tests that hold the block TRAJECTORY against an independent model (tests/ba_model.py).
uintptr_t block_trace_blocks(BlockHandle b, uint32_t *out, uintptr_t cap) {
    const Trace &t = ((Block *) b)->trace;
    for (size_t k = 0; k < t.blockIdx && k < cap; k++) {
        out[5 * k] = t.blockStart[2 * k]; out[5 * k + 1] = t.blockStart[2 * k + 1];
        out[5 * k + 2] = t.blockSize[2 * k]; out[5 * k + 3] = t.blockSize[2 * k + 1];
        out[5 * k + 4] = (uint32_t) ((t.right[k / 64] >> (k % 64)) & 1u);
    }
    return t.blockIdx;
}

} // extern "C"
