// k_btrace.hpp -- start position + backtrace of accepted hits ON THE DEVICE (round 5): the adaptive-block X-drop aligner structurealign
// takes them from (StructureSmithWaterman::alignStartPosBacktraceBlock, F/src/commons/StructureSmithWaterman.cpp:369-537, over the Rust crate
// M/lib/block-aligner: align_core scan_block.rs:120-630, place_block_3di :1302-1443, Trace / cigar :1726-2007, AVX2 configuration L = 16).
//
// The host restatement (host/block_aligner.cpp) is a lane-exact emulation of the crate's 16 x int16 vector code, because CIGARs depend on its
// saturation corners, tie-breaks and block trajectory.  Here a 16-lane DPP ROW RUNS ONE ALIGNMENT -- four alignments per wave (round 6; round 5
// ran one per wave with the four rows computing the same thing, and the kernel is bound by instruction issue: ~390 k wave instructions per
// alignment) -- and the row IS the crate's vector: lane k of a row holds lane k of every vector as a sign-extended int, each vector helper of the
// host file is the same expression on row shuffles (simd_sl_i16 = row shift by one, simd_step = rotate by 8, a lane broadcast = row_newbcast, the
// 128-bit-half quirks of the prefix scan as written there).  What align_core decides per block (direction, grow / shrink, x-drop, checkpoints) is
// uniform inside a row and differs between the rows of a wave: the rows share the instruction stream where they agree (right and down steps run
// the same code on swapped operands) and take turns where they do not; the host hands the tasks over sorted by size so that the four alignments of
// a wave are of similar length.  Column / row state lives in LDS (2.4 KB per row), the padded sequences, the trace words and the block list of an
// alignment in global scratch.  The block may grow to kBtMaxBlock rows; an alignment that wants a larger block, or does not reach the SW score with starting sizes
// 32 .. kBtMaxBlock, is handed back (status 0) and takes the host path -- same answers either way, which is what tests/test_btrace_gpu.py holds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fs {

constexpr int kBtL = 16;                 // avx2.rs:11
constexpr int kBtZero = 1 << 14;         // avx2.rs:15
constexpr int kBtMin = 0;                // avx2.rs:16
constexpr int kBtStep = 8;               // scan_block.rs:813
constexpr int kBtXDropIter = 2;          // scan_block.rs:814
constexpr int kBtMaxBlock = 128;         // largest block size of the first pass (16 alignments per workgroup)
constexpr int kBtMaxBlock2 = 512;        // ... of the second pass over what the first handed back (4 alignments per workgroup; the crate: 4096)
constexpr int kBtPad = kBtMaxBlock2 + 2 * kBtL + 16;     // readable bytes behind a sequence
constexpr int kBtNull = 26;              // AA_NULL: b'A' + 26 - b'A'
constexpr int kBtRows = 16;               // alignments (rows of 16 lanes) per workgroup of the first pass
constexpr int kBtRows2 = 4;               // ... of the second pass

struct BtTask {                          // one accepted hit
    uint32_t query, target;              // index into the call's queries / target id in the resident DB
    int32_t qEnd, dbEnd, score;          // end cell and score of the forward SW
    uint32_t pad;
    uint64_t seqOff, traceOff, blockOff, btOff;   // this task's slices of the scratch buffers (bytes / words / blocks / bytes)
};
struct BtQuery { uint32_t off, L; };     // forward codes AA [L], 3Di [L], bias int16 [L] (cbAA + cbSS) at off / off + L / off + 2 L (bias 2-byte aligned)
struct BtRes { int32_t status, qStart, dbStart, identicalAA, btLen, blockSizes; };   // status: 0 = host path, 1 = ok, 2 = SW score not reproduced (start -1, no backtrace)

struct BtArgs {
    const BtTask *tasks; int nTasks;
    const BtQuery *queries; const uint8_t *qdata;
    const uint8_t *dbAA, *dbSS; const uint64_t *dbOff; const int32_t *dbLen;
    const int8_t *tblAA, *tblSS;         // [27][32] block-aligner matrices, letter-indexed
    const uint8_t *letAA, *letSS;        // [21] code -> letter index (letter - 'A')
    int gapOpen, gapExtend;              // negative (Gaps of the crate)
    uint8_t *seq; uint32_t *trace; uint4 *blocks; char *bt; BtRes *res;
};

// ---- the crate's vector helpers on a 16-lane row (host/block_aligner.cpp, scalar definitions) ----
__device__ __forceinline__ int btSat(int x) { return min(max(x, -32768), 32767); }
__device__ __forceinline__ int btAdds(int a, int b) { return btSat(a + b); }
__device__ __forceinline__ int btSubs(int a, int b) { return btSat(a - b); }
// Row shuffles as DPP modifiers (a 16-lane DPP row is the vector; ds_bpermute costs an LDS round trip per shuffle and a block step is a chain of a dozen):
// row_shr:n = 0x110 + n (lane i takes lane i - n, 0 where there is none), row_ror:n = 0x120 + n, row_newbcast:k = 0x150 + k (every lane of a row takes
// the row's lane k).
template <int CTRL> __device__ __forceinline__ int btDpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int K> __device__ __forceinline__ int btBcast(int v) { return btDpp<0x150 + K>(v); }
__device__ __forceinline__ int btSl1(int a, int b, int ln) { const int t = btDpp<0x111>(a), c = btBcast<kBtL - 1>(b); return ln == 0 ? c : t; }
__device__ __forceinline__ int btStep8(int a, int b, int ln) { const int xa = btDpp<0x128>(a), xb = btDpp<0x128>(b); return ln < 8 ? xb : xa; }
template <int N> __device__ __forceinline__ int btSllz(int a, int ln) { const int t = btDpp<0x110 + N>(a); return (ln & 7) >= N ? t : 0; }
template <int B> __device__ __forceinline__ int btSlli16(int a) { return (int) (int16_t) (uint16_t) ((uint32_t) a << B); }
__device__ __forceinline__ int btHmax(int a) {          // all-reduce by rotations: every lane ends with the row's maximum
    a = max(a, btDpp<0x121>(a)); a = max(a, btDpp<0x122>(a)); a = max(a, btDpp<0x124>(a)); a = max(a, btDpp<0x128>(a));
    return a;
}
struct BtScanConsts { int gapExtendAll, lane; };
__device__ __forceinline__ BtScanConsts btPrefixScanConsts(int gap, int ln) {
    const int shift1 = btAdds(btSllz<1>(gap, ln), gap);
    const int shift2 = btAdds(btSllz<2>(shift1, ln), shift1);
    const int shift4 = btAdds(btSllz<4>(shift2, ln), shift2);
    const int s7 = btBcast<7>(shift4);
    BtScanConsts c;
    c.gapExtendAll = btAdds(ln < 8 ? 0 : s7, shift4);
    c.lane = shift4;
    return c;
}
__device__ __forceinline__ int btPrefixScan(int Rmax, int gapCost, int gapCostLane, int ln) {
    const int shift1 = max(Rmax, btAdds(btSllz<1>(Rmax, ln), gapCost));
    const int shift2 = max(shift1, btAdds(btSllz<2>(shift1, ln), btSlli16<1>(gapCost)));
    const int shift4 = max(shift2, btAdds(btSllz<4>(shift2, ln), btSlli16<2>(gapCost)));
    const int lowq = ln < 4 ? shift4 : btDpp<0x114>(shift4), s7 = btBcast<7>(shift4);          // lanes 0..7 take lane (ln & 3)
    const int correct1 = btAdds(ln < 8 ? lowq : s7, gapCostLane);
    return max(shift4, correct1);
}
__device__ __forceinline__ uint32_t btSpread16(uint32_t x) {
    x &= 0xffffu;
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    return x;
}
// bit 2k <- lo lane k, bit 2k + 1 <- hi lane k of this lane's row (sh = 16 x row of the wave; the rows that are not here with this one do not matter)
__device__ __forceinline__ uint32_t btMask2(bool lo, bool hi, int sh) {
    const uint32_t a = (uint32_t) (__ballot(lo) >> sh), b = (uint32_t) (__ballot(hi) >> sh);
    return btSpread16(a) | (btSpread16(b) << 1);
}

struct BtSeq { const uint8_t *aa, *ss; const int16_t *bias; int len; };      // padded: index 0 = AA_NULL, residues at 1 .. len, AA_NULL behind

template <int MAXB>
struct BtState {                       // per-row LDS
    static constexpr int kMaxBlock = MAXB;
    int16_t Dcol[MAXB + kBtL], Ccol[MAXB + kBtL], Drow[MAXB + kBtL], Rrow[MAXB + kBtL];
    int16_t DcolCk[MAXB + kBtL], CcolCk[MAXB + kBtL], DrowCk[MAXB + kBtL], RrowCk[MAXB + kBtL];
    int16_t tmp1[kBtL], tmp2[kBtL];
    // the letters and biases of the block being placed (btPlaceBlock): the sequences live in global scratch, a vector step must not wait for them
    int16_t wqb[MAXB], wrb[MAXB / 2];                          // (height <= MAXB; the reference window is refilled every MAXB / 2 columns)
    uint8_t wqa[MAXB], wqs[MAXB], wra[MAXB / 2], wrs[MAXB / 2];
};

struct BtTrace {                       // row-uniform values + the task's global arrays
    uint32_t *trace, *trace2;
    uint4 *blocks;                     // {i, j, height | width << 16, right}
    uint32_t traceIdx, blockIdx, ckptTraceIdx, ckptBlockIdx;
    uint32_t traceCap;                 // words of each of the two trace arrays
};

struct BtPB { int Dmax, argI, argJ; };

__device__ __forceinline__ void btWaveSync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// place_block_3di (scan_block.rs:1302-1443); `query` runs along the vector dimension
template <class ST>
__device__ __forceinline__ BtPB btPlaceBlock(const BtSeq &query, const BtSeq &reference, const int8_t *tblAA, const int8_t *tblSS, int gapOpen, int gapExtend,
                                             int startI, int startJ, int width, int height, int16_t *DcolP, int16_t *CcolP, int16_t *DrowP, int16_t *RrowP,
                                             int Dcorner, BtTrace &tr, ST &S, int ln, int sh) {
    const BtScanConsts sc = btPrefixScanConsts(gapExtend, ln);
    BtPB out; out.Dmax = kBtMin; out.argI = 0; out.argJ = 0;
    if (width == 0 || height == 0) return out;
    const int openMinusExt = btSubs(gapOpen, gapExtend);
    // the block's letters into LDS: every global read of the block is in flight at once (width, height <= kBtMaxBlock)
    for (int i = ln; i < height; i += kBtL) { S.wqa[i] = query.aa[startI + i]; S.wqs[i] = query.ss[startI + i]; S.wqb[i] = query.bias ? query.bias[startI + i] : (int16_t) 0; }
    btWaveSync();
    constexpr int kW = ST::kMaxBlock / 2;           // columns per fill of the reference window (the very first block of an attempt is minSize <= MAXB wide)
    for (int j = 0; j < width; j++) {
        if ((j & (kW - 1)) == 0) {
            for (int jj = ln; jj < kW && j + jj < width; jj += kBtL) {
                S.wra[jj] = reference.aa[startJ + j + jj]; S.wrs[jj] = reference.ss[startJ + j + jj]; S.wrb[jj] = reference.bias ? reference.bias[startJ + j + jj] : (int16_t) 0;
            }
            btWaveSync();
        }
        int R01 = kBtMin, D11 = kBtMin, R11 = kBtMin;
        bool prevTraceR = false;
        const int c = S.wra[j & (kW - 1)], c3 = S.wrs[j & (kW - 1)];
        const int refBias = S.wrb[j & (kW - 1)];
        for (int i = 0; i < height; i += kBtL) {
            const int D10 = DcolP[i + ln], C10 = CcolP[i + ln];
            const int D00 = btSl1(D10, Dcorner, ln);
            Dcorner = D10;
            int scores = tblAA[c * 32 + (S.wqa[i + ln] & 31)];
            {
                const int s3 = tblSS[c3 * 32 + (S.wqs[i + ln] & 31)];
                const int qBias = S.wqb[i + ln];
                scores = btAdds(btAdds(scores, s3), btAdds(refBias, qBias));
            }
            D11 = btAdds(D00, scores);
            if (startI + i == 0 && startJ + j == 0 && ln == 0) D11 = kBtZero;
            const int C11open = btAdds(D10, gapOpen);
            const int C11 = max(btAdds(C10, gapExtend), C11open);
            D11 = max(D11, C11);
            const int D11open = btAdds(D11, openMinusExt);
            R11 = btPrefixScan(D11open, gapExtend, sc.lane, ln);
            R11 = max(R11, btAdds(btBcast<kBtL - 1>(R01), sc.gapExtendAll));
            D11 = max(D11, R11);
            R01 = R11;
            {
                const uint32_t t = btMask2(D11 == C11, D11 == R11, sh);
                const bool tempTraceR = R11 == D11open;
                // traceR = simd_sl_i16(tempTraceR, prevTraceR, 1)
                const int up = btDpp<0x111>((int) tempTraceR), last = btBcast<kBtL - 1>((int) prevTraceR);
                const bool traceR = (ln == 0 ? last : up) != 0;
                const uint32_t t2 = btMask2(C11 == C11open, traceR, sh);
                prevTraceR = tempTraceR;
                if (ln == 0) { tr.trace[tr.traceIdx] = t; tr.trace2[tr.traceIdx] = t2; }
                tr.traceIdx++;
            }
            out.Dmax = max(out.Dmax, D11);
            if (out.Dmax == D11) { out.argI = i; out.argJ = j; }
            // (LDS executes a wave's accesses in order and these go through one base pointer: no barrier needed between the loads above and the stores)
            DcolP[i + ln] = (int16_t) D11;
            CcolP[i + ln] = (int16_t) C11;
        }
        Dcorner = kBtMin;
        if (ln == kBtL - 1) { DrowP[j] = (int16_t) D11; RrowP[j] = (int16_t) R11; }
        btWaveSync();
    }
    return out;
}

__device__ __forceinline__ void btJustOffset(int blockSize, int16_t *b1, int16_t *b2, int offAdd, int ln) {
    for (int i = 0; i < blockSize; i += kBtL) { b1[i + ln] = (int16_t) btAdds(b1[i + ln], offAdd); b2[i + ln] = (int16_t) btAdds(b2[i + ln], offAdd); }
    btWaveSync();
}
// shift_and_offset (scan_block.rs:1096-1123)
__device__ __forceinline__ int btShiftAndOffset(int blockSize, int16_t *b1, int16_t *b2, const int16_t *t1, const int16_t *t2, int offAdd, int ln) {
    int curr1 = btAdds(b1[ln], offAdd);
    const int Dcorner = btBcast<kBtStep - 1>(curr1);
    int curr2 = btAdds(b2[ln], offAdd);
    int i = 0;
    while (i < blockSize - kBtL) {
        const int next1 = btAdds(b1[i + kBtL + ln], offAdd), next2 = btAdds(b2[i + kBtL + ln], offAdd);
        btWaveSync();
        b1[i + ln] = (int16_t) btStep8(next1, curr1, ln);
        b2[i + ln] = (int16_t) btStep8(next2, curr2, ln);
        btWaveSync();
        curr1 = next1; curr2 = next2;
        i += kBtL;
    }
    const int a1 = t1[ln], a2 = t2[ln];
    btWaveSync();
    b1[blockSize - kBtL + ln] = (int16_t) btStep8(a1, curr1, ln);
    b2[blockSize - kBtL + ln] = (int16_t) btStep8(a2, curr2, ln);
    btWaveSync();
    return Dcorner;
}
__device__ __forceinline__ int btPrefixHmax8(const int16_t *p, int ln) { return btHmax(ln < 8 ? (int) p[ln] : -32768); }
__device__ __forceinline__ int btSuffixHmax2(const int16_t *p) { return max((int) p[kBtL - 1], (int) p[kBtL - 2]); }
__device__ __forceinline__ void btCopy(int16_t *dst, const int16_t *src, int n, int ln) {
    for (int i = 0; i < n; i += kBtL) dst[i + ln] = src[i + ln];
}
__device__ __forceinline__ int btClamp16(int x) { return btSat(x); }

enum { kBtRight = 0, kBtDown = 1, kBtGrow = 2 };

// align_core (scan_block.rs:120-630) with trace and x-drop.  Returns false when the block wants to grow beyond kBtMaxBlock.
template <class ST>
__device__ __forceinline__ bool btAlign(ST &S, BtTrace &tr, const BtSeq &query, const BtSeq &reference, const int8_t *tblAA, const int8_t *tblSS, int gapOpen, int gapExtend,
                                        int minSize, int maxSize, int xDropThr, int ln, int sh, int &resScore, int &resQ, int &resR) {
    // clear
    tr.traceIdx = tr.blockIdx = tr.ckptTraceIdx = tr.ckptBlockIdx = 0;
    constexpr int MAXB = ST::kMaxBlock;
    for (int i = 0; i < MAXB + kBtL; i += kBtL) {
        S.Dcol[i + ln] = kBtMin; S.Ccol[i + ln] = kBtMin; S.Drow[i + ln] = kBtMin; S.Rrow[i + ln] = kBtMin;
        S.DcolCk[i + ln] = kBtMin; S.CcolCk[i + ln] = kBtMin; S.DrowCk[i + ln] = kBtMin; S.RrowCk[i + ln] = kBtMin;
    }
    S.tmp1[ln] = kBtMin; S.tmp2[ln] = kBtMin;
    btWaveSync();
    int si = 0, sj = 0;
    int bestMax = 0, bestArgI = 0, bestArgJ = 0;
    int prevDir = kBtGrow, dir = kBtGrow;
    int prevSize = 0, blockSize = minSize;
    int off = 0, prevOff, offMax = 0;
    int yDropIter = 0, xDropIter = 0;
    int iCkpt = si, jCkpt = sj, offCkpt = 0;
    int Dcorner = kBtMin;
    const int qLen = query.len, rLen = reference.len;
    const uint32_t blockCap = (uint32_t) (qLen + rLen + 12);
    auto addBlock = [&](int i, int j, int width, int height, int right) {
        if (ln == 0 && tr.blockIdx < blockCap) tr.blocks[tr.blockIdx] = make_uint4((uint32_t) i, (uint32_t) j, (uint32_t) height | ((uint32_t) width << 16), (uint32_t) right);
        tr.blockIdx++;
    };
    auto copyToCkpt = [&](int n) {
        btCopy(S.DcolCk, S.Dcol, n, ln); btCopy(S.CcolCk, S.Ccol, n, ln); btCopy(S.DrowCk, S.Drow, n, ln); btCopy(S.RrowCk, S.Rrow, n, ln);
        btWaveSync();
    };
    for (;;) {
        if (tr.blockIdx + 2 >= blockCap) return false;            // cannot happen while the walk follows the crate (a block per 8 rows or columns); the host path answers
        if (tr.traceIdx + (uint32_t) (3 * MAXB * MAXB / (4 * kBtL)) > tr.traceCap) return false;   // the largest step (a grow to MAXB: two blocks) must fit the task's slice
        prevOff = off;
        int growDmax = kBtMin, growArgI = 0, growArgJ = 0;
        BtPB pb;
        int rightMax, downMax;
        if (dir != kBtGrow) {
            // a step to the right and a step down are the same code on swapped operands (scan_block.rs:237-330): the rows of a wave that step in
            // different directions stay in one instruction stream
            const bool rt = dir == kBtRight;
            int16_t *c1 = rt ? S.Dcol : S.Drow, *c2 = rt ? S.Ccol : S.Rrow, *o1 = rt ? S.Drow : S.Dcol, *o2 = rt ? S.Rrow : S.Ccol;
            BtSeq sa, sb;
            sa.aa = rt ? query.aa : reference.aa; sa.ss = rt ? query.ss : reference.ss; sa.bias = rt ? query.bias : reference.bias; sa.len = rt ? query.len : reference.len;
            sb.aa = rt ? reference.aa : query.aa; sb.ss = rt ? reference.ss : query.ss; sb.bias = rt ? reference.bias : query.bias; sb.len = rt ? reference.len : query.len;
            off = offMax;
            const int offAdd = btClamp16(prevOff - off);
            addBlock(rt ? si : si + blockSize - kBtStep, rt ? sj + blockSize - kBtStep : sj, rt ? kBtStep : blockSize, rt ? blockSize : kBtStep, rt ? 1 : 0);
            btJustOffset(blockSize, c1, c2, offAdd, ln);
            pb = btPlaceBlock(sa, sb, tblAA, tblSS, gapOpen, gapExtend, rt ? si : sj, (rt ? sj : si) + blockSize - kBtStep, kBtStep, blockSize, c1, c2, S.tmp1, S.tmp2,
                              prevDir == (rt ? kBtDown : kBtRight) ? btAdds(Dcorner, offAdd) : kBtMin, tr, S, ln, sh);
            const int m1 = btPrefixHmax8(c1, ln);
            Dcorner = btShiftAndOffset(blockSize, o1, o2, S.tmp1, S.tmp2, offAdd, ln);
            const int m2 = btPrefixHmax8(o1, ln);
            rightMax = rt ? m1 : m2; downMax = rt ? m2 : m1;
        } else {
            Dcorner = kBtMin;
            const int growStep = blockSize - prevSize;
            addBlock(si + prevSize, sj, prevSize, growStep, 0);
            const BtPB p1 = btPlaceBlock(reference, query, tblAA, tblSS, gapOpen, gapExtend, sj, si + prevSize, growStep, prevSize, S.Drow, S.Rrow, S.Dcol + prevSize,
                                         S.Ccol + prevSize, kBtMin, tr, S, ln, sh);
            addBlock(si, sj + prevSize, growStep, blockSize, 1);
            pb = btPlaceBlock(query, reference, tblAA, tblSS, gapOpen, gapExtend, si, sj + prevSize, growStep, blockSize, S.Dcol, S.Ccol, S.Drow + prevSize,
                              S.Rrow + prevSize, kBtMin, tr, S, ln, sh);
            rightMax = btPrefixHmax8(S.Dcol, ln);
            downMax = btPrefixHmax8(S.Drow, ln);
            growDmax = p1.Dmax; growArgI = p1.argI; growArgJ = p1.argJ;
            copyToCkpt(blockSize);
            tr.ckptTraceIdx = tr.traceIdx; tr.ckptBlockIdx = tr.blockIdx;
        }
        prevDir = dir;
        const int DmaxMax = btHmax(pb.Dmax), growMax = btHmax(growDmax);
        const int mx = max(DmaxMax, growMax);
        offMax = off + mx - kBtZero;
        yDropIter++;
        bool growNoMax = dir == kBtGrow;
        if (offMax > bestMax) {
            {
                const bool grow = dir == kBtGrow && DmaxMax < growMax;
                const int currMax = grow ? growMax : DmaxMax;
                const int cd = grow ? growDmax : pb.Dmax, ci = grow ? growArgI : pb.argI, cj = grow ? growArgJ : pb.argJ;
                // per lane: the cell of its maximum; between lanes the largest column, then the largest row (the sequential "better" of the crate
                // is a strict lexicographic maximum starting from (0, 0))
                unsigned long long key = 0;
                if (cd == currMax) {
                    const int idxI = (int) (uint16_t) ci, idxJ = (int) (uint16_t) cj;
                    const int r = idxI + ln, c = (blockSize - kBtStep) + idxJ;
                    int gi, gj;
                    if (grow) { gi = si + prevSize + idxJ; gj = sj + idxI + ln; }
                    else if (dir == kBtRight) { gi = si + r; gj = sj + c; }
                    else if (dir == kBtDown) { gi = si + c; gj = sj + r; }
                    else { gi = si + idxI + ln; gj = sj + prevSize + idxJ; }
                    key = ((unsigned long long) (uint32_t) gj << 32) | (uint32_t) gi;
                }
#define FS_BT_KEYMAX(CTRL)                                                                                                        \
                {                                                                                                                 \
                    const uint32_t lo = (uint32_t) btDpp<CTRL>((int) (uint32_t) key), hi = (uint32_t) btDpp<CTRL>((int) (uint32_t) (key >> 32)); \
                    const unsigned long long o = ((unsigned long long) hi << 32) | lo;                                           \
                    key = o > key ? o : key;                                                                                      \
                }
                FS_BT_KEYMAX(0x121) FS_BT_KEYMAX(0x122) FS_BT_KEYMAX(0x124) FS_BT_KEYMAX(0x128)
#undef FS_BT_KEYMAX
                bestArgI = (int) (uint32_t) key; bestArgJ = (int) (uint32_t) (key >> 32);
            }
            if (blockSize < maxSize) {
                iCkpt = si; jCkpt = sj; offCkpt = off;
                copyToCkpt(blockSize);
                tr.ckptTraceIdx = tr.traceIdx; tr.ckptBlockIdx = tr.blockIdx;
                growNoMax = false;
            }
            bestMax = offMax;
            yDropIter = 0;
        }
        if (offMax < bestMax - xDropThr) {
            if (xDropIter < kBtXDropIter - 1) xDropIter++;
            else break;
        } else {
            xDropIter = 0;
        }
        if (si + blockSize > qLen && sj + blockSize > rLen) break;
        if (sj + blockSize > rLen) { si += kBtStep; dir = kBtDown; continue; }
        if (si + blockSize > qLen) { sj += kBtStep; dir = kBtRight; continue; }
        const int nextSize = blockSize * 2;
        if (nextSize <= maxSize) {
            if (yDropIter > (blockSize / kBtStep) - 1 || growNoMax) {
                if (nextSize > MAXB) return false;                 // the next pass / the host path continues where this pass's LDS ends
                prevSize = blockSize;
                blockSize = nextSize;
                dir = kBtGrow;
                si = iCkpt; sj = jCkpt; off = offCkpt;
                btCopy(S.Dcol, S.DcolCk, prevSize, ln); btCopy(S.Ccol, S.CcolCk, prevSize, ln); btCopy(S.Drow, S.DrowCk, prevSize, ln); btCopy(S.Rrow, S.RrowCk, prevSize, ln);
                btWaveSync();
                tr.traceIdx = tr.ckptTraceIdx; tr.blockIdx = tr.ckptBlockIdx;
                yDropIter = 0;
                continue;
            }
        }
        if (blockSize > minSize && yDropIter == 0) {              // SHRINK
            const int shrinkMax = max(btSuffixHmax2(S.Drow + blockSize - kBtL), btSuffixHmax2(S.Dcol + blockSize - kBtL));
            if (shrinkMax >= mx) {
                prevDir = kBtGrow;
                blockSize /= 2;
                for (int i = 0; i < blockSize; i += kBtL) {
                    const int a = S.Dcol[i + blockSize + ln], b = S.Ccol[i + blockSize + ln], c = S.Drow[i + blockSize + ln], d = S.Rrow[i + blockSize + ln];
                    btWaveSync();
                    S.Dcol[i + ln] = (int16_t) a; S.Ccol[i + ln] = (int16_t) b; S.Drow[i + ln] = (int16_t) c; S.Rrow[i + ln] = (int16_t) d;
                    btWaveSync();
                }
                si += blockSize; sj += blockSize;
                iCkpt = si; jCkpt = sj; offCkpt = off;
                copyToCkpt(blockSize);
                rightMax = btPrefixHmax8(S.Dcol, ln);
                downMax = btPrefixHmax8(S.Drow, ln);
                tr.ckptTraceIdx = tr.traceIdx; tr.ckptBlockIdx = tr.blockIdx;
                yDropIter = 0;
            }
        }
        if (downMax > rightMax) { si += kBtStep; dir = kBtDown; }
        else { sj += kBtStep; dir = kBtRight; }
    }
    resScore = bestMax; resQ = bestArgI; resR = bestArgJ;
    return true;
}

// Trace::cigar_core (scan_block.rs:1844-2007) from cell (i, j) back to (0, 0): writes one character per operation in the order the steps are taken
// (= the forward order of the ORIGINAL alignment: the aligned sequences are the reversed prefixes) and counts identical amino acids under M.
__device__ __forceinline__ int btCigar(const BtTrace &tr, int i, int j, const uint8_t *qAA, const uint8_t *rAA, char *bt, int &identical, bool writer) {
    uint32_t blockIdx = tr.blockIdx, traceIdx = tr.traceIdx;
    int blockI = 0, blockJ = 0, blockW = 0, blockH = 0, rightFlag = 0;
    int table = 0;                                       // 0 = D, 1 = C, 2 = R
    int n = 0;
    identical = 0;
    const int limit = i + j + 4;                         // a walk takes at most i + j steps: anything longer is a broken trace, not an answer
    while (i > 0 || j > 0) {
        for (;;) {
            if (blockIdx == 0) return -1;
            blockIdx--;
            const uint4 b = tr.blocks[blockIdx];
            blockI = (int) b.x; blockJ = (int) b.y; blockH = (int) (b.z & 0xffffu); blockW = (int) (b.z >> 16);
            traceIdx -= (uint32_t) (blockW * blockH / kBtL);
            if (i >= blockI && j >= blockJ) { rightFlag = (int) b.w; break; }
        }
        while (i >= blockI && j >= blockJ && (i > 0 || j > 0)) {
            const int ci = i - blockI, cj = j - blockJ;
            uint32_t idx, sh;
            if (rightFlag) { idx = traceIdx + (uint32_t) (ci / kBtL + cj * (blockH / kBtL)); sh = (uint32_t) (ci % kBtL) * 2; }
            else { idx = traceIdx + (uint32_t) (cj / kBtL + ci * (blockW / kBtL)); sh = (uint32_t) (cj % kBtL) * 2; }
            const uint32_t t = (tr.trace[idx] >> sh) & 3u, t2 = (tr.trace2[idx] >> sh) & 3u;
            const bool t2b0 = t2 & 1u, t2b1 = t2 & 2u;
            // the crate's OP_LUT (scan_block.rs:1870-1925): first operand of every pair is the table the walk is in
            int op, di, dj, nt;                          // op: 0 = M, 1 = I, 2 = D
            if (rightFlag) {
                if (table == 1) { op = 2; di = 0; dj = 1; nt = t2b0 ? 0 : 1; }
                else if (table == 2) { op = 1; di = 1; dj = 0; nt = t2b1 ? 0 : 2; }
                else if (t == 0) { op = 0; di = 1; dj = 1; nt = 0; }
                else if (t == 1 || t == 3) { op = 2; di = 0; dj = 1; nt = t2b0 ? 0 : 1; }
                else { op = 1; di = 1; dj = 0; nt = t2b1 ? 0 : 2; }
            } else {
                if (table == 2) { op = 1; di = 1; dj = 0; nt = t2b0 ? 0 : 2; }
                else if (table == 1) { op = 2; di = 0; dj = 1; nt = t2b1 ? 0 : 1; }
                else if (t == 0) { op = 0; di = 1; dj = 1; nt = 0; }
                else if (t == 1 || t == 3) { op = 1; di = 1; dj = 0; nt = t2b0 ? 0 : 2; }
                else { op = 2; di = 0; dj = 1; nt = t2b1 ? 0 : 1; }
            }
            if (op == 0 && qAA[i] == rAA[j]) identical++;   // padded index i <-> reversed-prefix residue i - 1 of both sequences
            if (n >= limit) return -1;
            if (writer) bt[n] = op == 0 ? 'M' : op == 1 ? 'I' : 'D';
            n++;
            i -= di; j -= dj; table = nt;
        }
    }
    return n;
}

// alignStartPosBacktraceBlock (StructureSmithWaterman.cpp:369-537) for a batch of accepted hits: one 16-lane row per task
// SPREAD = false: the ROWS alignments of a workgroup sit four to a wave (throughput form: rows that disagree about the next piece of code take turns).
// SPREAD = true: ONE alignment per wave, in its first 16 lanes (latency form, for calls with fewer alignments than the device has wave slots and for the
// second pass: what an alignment waits for is then its own dependent chain only -- round 5's shape, 2.7 ms per 1 600 alignments against ~9 ms four to a wave).
template <int MAXB, int ROWS, bool SPREAD = false>
__global__ __launch_bounds__(SPREAD ? ROWS * 64 : ROWS * kBtL) void k_block_backtrace(BtArgs a) {
    __shared__ BtState<MAXB> states[ROWS];
    __shared__ int8_t tAA[27 * 32], tSS[27 * 32];
    for (int i = threadIdx.x; i < 27 * 32; i += blockDim.x) { tAA[i] = a.tblAA[i]; tSS[i] = a.tblSS[i]; }
    __syncthreads();
    if (SPREAD && (threadIdx.x & 63u) >= (unsigned) kBtL) return;          // (no workgroup barrier below this line)
    const int row = SPREAD ? (int) (threadIdx.x >> 6) : (int) (threadIdx.x >> 4), ln = threadIdx.x & (kBtL - 1);
    const int sh = SPREAD ? 0 : (int) (threadIdx.x & 48u);                 // sh: first lane of the row inside its wave
    BtState<MAXB> &S = states[row];
    for (int task = blockIdx.x * ROWS + row; task < a.nTasks; task += gridDim.x * ROWS) {
        const BtTask tk = a.tasks[task];
        const BtQuery q = a.queries[tk.query];
        const int qn = tk.qEnd + 1, tn = tk.dbEnd + 1;
        // ---- the padded, reversed prefixes: letter indices, AA_NULL in front and behind; the query's position bias (cbAA + cbSS) ----
        const size_t qStride = ((size_t) 1 + qn + kBtPad + 15) & ~(size_t) 15, tStride = ((size_t) 1 + tn + kBtPad + 15) & ~(size_t) 15;
        const uint8_t *qa = a.qdata + q.off, *q3 = qa + q.L;
        const int16_t *qb = (const int16_t *) (a.qdata + q.off + 2 * (size_t) q.L);      // q.off is a multiple of 16
        const uint8_t *ta = a.dbAA + a.dbOff[tk.target], *t3 = a.dbSS + a.dbOff[tk.target];
        BtTrace tr;
        const size_t traceWords = (size_t) (MAXB / kBtL) * ((size_t) qn + tn + 2 * MAXB);
        tr.trace = a.trace + tk.traceOff; tr.trace2 = tr.trace + traceWords; tr.traceCap = (uint32_t) traceWords;
        tr.blocks = a.blocks + tk.blockOff;
        BtRes r;
        r.status = 0; r.qStart = -1; r.dbStart = -1; r.identicalAA = 0; r.btLen = 0; r.blockSizes = 0;
        uint8_t *pq = a.seq + tk.seqOff, *pq3 = pq + qStride, *pt = pq3 + qStride, *pt3 = pt + tStride;
        int16_t *pb = (int16_t *) (pt3 + tStride);
        for (int i = ln; i < (int) qStride; i += kBtL) {
            const int k = i - 1;                             // reversed-prefix index
            const bool in = k >= 0 && k < qn;
            pq[i] = in ? a.letAA[min((int) qa[tk.qEnd - k], 20)] : (uint8_t) kBtNull;
            pq3[i] = in ? a.letSS[min((int) q3[tk.qEnd - k], 20)] : (uint8_t) kBtNull;
            pb[i] = in ? qb[tk.qEnd - k] : (int16_t) 0;
        }
        for (int i = ln; i < (int) tStride; i += kBtL) {
            const int k = i - 1;
            const bool in = k >= 0 && k < tn;
            pt[i] = in ? a.letAA[min((int) ta[tk.dbEnd - k], 20)] : (uint8_t) kBtNull;
            pt3[i] = in ? a.letSS[min((int) t3[tk.dbEnd - k], 20)] : (uint8_t) kBtNull;
        }
        __threadfence_block();
        btWaveSync();
        const BtSeq qs{pq, pq3, pb, qn}, rs{pt, pt3, nullptr, tn};
        int score = -1000000000, rq = -1, rr = -1, sizes = 0;
        bool onDevice = true;
        for (int minSize = 32; minSize <= MAXB && score < tk.score; minSize *= 2) {
            const int xDrop = -(minSize * a.gapExtend + a.gapOpen);
            onDevice = btAlign(S, tr, qs, rs, tAA, tSS, a.gapOpen, a.gapExtend, minSize, 4096, xDrop, ln, sh, score, rq, rr);
            sizes++;
            if (!onDevice) break;
        }
        r.blockSizes = sizes;
        if (onDevice && score >= tk.score) {
            // reached (a larger starting size is never tried once the score is there); like the host: a score that differs from the SW score
            // leaves the hit without start position, except at int16 saturation
            if (!(score != tk.score && !(tk.score == 32767 && score >= tk.score))) {
                __threadfence_block();
                btWaveSync();
                int ident = 0;
                const int n = btCigar(tr, rq, rr, pq, pt, a.bt + tk.btOff, ident, ln == 0);
                if (n >= 0) { r.status = 1; r.qStart = (tk.qEnd + 1) - rq; r.dbStart = (tk.dbEnd + 1) - rr; r.identicalAA = ident; r.btLen = n; }
            } else r.status = 2;
        }
        if (ln == 0) a.res[task] = r;
        btWaveSync();
    }
}

} // namespace fs
