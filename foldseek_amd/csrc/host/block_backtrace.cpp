// block_backtrace.cpp -- start position + backtrace of an accepted hit, the way structurealign obtains them:
// StructureSmithWaterman::alignStartPosBacktraceBlock (reference F/src/commons/StructureSmithWaterman.cpp:369-537).
// The end position (qEnd, dbEnd) and score come from the device SW kernel; here the reversed prefixes
// query[0..qEnd] and target[0..dbEnd] are aligned from their ends with the adaptive-block X-drop aligner until
// the block size is large enough to reproduce the SW score (32, 64, ..., 4096).
#include "hostlib.h"
#include "block_aligner_abi.h"

#include <algorithm>
#include <cstring>

namespace fsh {

namespace {
struct Scratch {
    BlockHandle block = nullptr;
    size_t capQ = 0, capT = 0;
    // per-thread reusable inputs: the reference allocates these per call (StructureSmithWaterman.cpp:387-449)
    PaddedBytes *pqAA = nullptr, *pq3 = nullptr, *ptAA = nullptr, *pt3 = nullptr;
    PosBias *pqB = nullptr, *ptB = nullptr;
    AAMatrix *maa = nullptr, *m3 = nullptr;
    uint64_t maaKey = 0, m3Key = 0;                   // content hash of the fsh::Matrix the cached block matrices hold
    ~Scratch() {
        if (block) block_free_aa_trace_xdrop(block);
        if (pqAA) { block_free_padded_aa(pqAA); block_free_padded_aa(pq3); block_free_padded_aa(ptAA); block_free_padded_aa(pt3); }
        if (pqB) { block_free_pos_bias(pqB); block_free_pos_bias(ptB); }
        if (maa) { block_free_aamatrix(maa); block_free_aamatrix(m3); }
    }
};
thread_local Scratch g_scratch;
uint64_t matrixHash(const Matrix &m) {
    uint64_t h = 1469598103934665603ull;
    for (short v : m.sub) { h ^= (uint16_t) v; h *= 1099511628211ull; }
    for (char c : m.letters) { h ^= (uint8_t) c; h *= 1099511628211ull; }
    return h | 1;
}
constexpr size_t MAX_SIZE = 4096;
}

void blockBacktrace(const Matrix &mAA, const Matrix &m3Di, const uint8_t *qAA, const uint8_t *q3Di, const int8_t *cbAA, const int8_t *cbSS,
                    int Lq, const uint8_t *tAA, const uint8_t *t3Di, int /*Lt*/, int qEnd, int dbEnd, int targetScore, int gapOpen,
                    int gapExtend, BlockAlnOut &out) {
    out = BlockAlnOut();
    Gaps gaps;
    gaps.open = (int8_t) -gapOpen;
    gaps.extend = (int8_t) -gapExtend;
    const size_t queryAlnLen = (size_t) qEnd + 1, targetAlnLen = (size_t) dbEnd + 1;
    Scratch &sc = g_scratch;
    if (!sc.block || sc.capQ < queryAlnLen || sc.capT < targetAlnLen) {
        if (sc.block) block_free_aa_trace_xdrop(sc.block);
        sc.capQ = std::max<size_t>(queryAlnLen, sc.capQ) + 64;
        sc.capT = std::max<size_t>(targetAlnLen, sc.capT) + 64;
        sc.block = block_new_aa_trace_xdrop(sc.capQ, sc.capT, MAX_SIZE);
    }
    // reversed query prefix = query_*_rev_sequence[queryStartPos ..], queryStartPos = Lq - (qEnd + 1) (:394-403)
    std::string qAAs(queryAlnLen, 'X'), q3s(queryAlnLen, 'X'), tAAs(targetAlnLen, 'X'), t3s(targetAlnLen, 'X');
    std::vector<int16_t> qBias(queryAlnLen), tBias(targetAlnLen, 0);
    for (size_t i = 0; i < queryAlnLen; i++) {
        const int src = qEnd - (int) i;                 // rev[queryStartPos + i] = fwd[Lq - 1 - (Lq - qEnd - 1 + i)]
        qAAs[i] = mAA.letters[qAA[src]];
        q3s[i] = m3Di.letters[q3Di[src]];
        qBias[i] = (int16_t) (cbAA[src] + cbSS[src]);   // composition_bias_*_rev at the same reversed index
    }
    for (size_t i = 0; i < targetAlnLen; i++) {
        tAAs[i] = mAA.letters[tAA[dbEnd - (int) i]];
        t3s[i] = m3Di.letters[t3Di[dbEnd - (int) i]];
    }
    (void) Lq;
    if (!sc.pqAA) {
        sc.pqAA = block_new_padded_aa(queryAlnLen, MAX_SIZE); sc.pq3 = block_new_padded_aa(queryAlnLen, MAX_SIZE);
        sc.ptAA = block_new_padded_aa(targetAlnLen, MAX_SIZE); sc.pt3 = block_new_padded_aa(targetAlnLen, MAX_SIZE);
        sc.pqB = block_new_pos_bias(queryAlnLen, MAX_SIZE); sc.ptB = block_new_pos_bias(targetAlnLen, MAX_SIZE);
    }
    PaddedBytes *pqAA = sc.pqAA, *pq3 = sc.pq3, *ptAA = sc.ptAA, *pt3 = sc.pt3;
    PosBias *pqB = sc.pqB, *ptB = sc.ptB;
    block_set_bytes_padded_aa(pqAA, (const uint8_t *) qAAs.data(), queryAlnLen, MAX_SIZE);
    block_set_bytes_padded_aa(pq3, (const uint8_t *) q3s.data(), queryAlnLen, MAX_SIZE);
    block_set_pos_bias(pqB, qBias.data(), queryAlnLen);
    block_set_bytes_padded_aa(ptAA, (const uint8_t *) tAAs.data(), targetAlnLen, MAX_SIZE);
    block_set_bytes_padded_aa(pt3, (const uint8_t *) t3s.data(), targetAlnLen, MAX_SIZE);
    block_set_pos_bias(ptB, tBias.data(), targetAlnLen);
    // matrices keyed by letter, filled from the short substitution scores cast to int8 (:428-447)
    if (!sc.maa) { sc.maa = block_new_simple_aamatrix(1, -1); sc.m3 = block_new_simple_aamatrix(1, -1); }
    AAMatrix *maa = sc.maa, *m3 = sc.m3;
    const uint64_t hA = matrixHash(mAA), h3 = matrixHash(m3Di);
    if (sc.maaKey != hA) {
        for (int a = 0; a < mAA.n; a++)
            for (int b = 0; b < mAA.n; b++) block_set_aamatrix(maa, (uint8_t) mAA.letters[a], (uint8_t) mAA.letters[b], (int8_t) mAA.sub[a * mAA.n + b]);
        sc.maaKey = hA;
    }
    if (sc.m3Key != h3) {
        for (int a = 0; a < m3Di.n; a++)
            for (int b = 0; b < m3Di.n; b++) block_set_aamatrix(m3, (uint8_t) m3Di.letters[a], (uint8_t) m3Di.letters[b], (int8_t) m3Di.sub[a * m3Di.n + b]);
        sc.m3Key = h3;
    }

    AlignResult res;
    res.score = -1000000000; res.query_idx = (uintptr_t) -1; res.reference_idx = (uintptr_t) -1;
    size_t minSize = 32;
    while (minSize <= MAX_SIZE && res.score < targetScore) {
        SizeRange range; range.min = minSize; range.max = MAX_SIZE;
        const int32_t xDrop = -((int32_t) minSize * gaps.extend + gaps.open);
        block_align_3di_aa_trace_xdrop(sc.block, pqAA, pq3, pqB, ptAA, pt3, ptB, maa, m3, gaps, range, xDrop);
        res = block_res_aa_trace_xdrop(sc.block);
        minSize *= 2;
    }
    if (!(res.score != targetScore && !(targetScore == INT16_MAX && res.score >= targetScore))) {
        Cigar *cigar = block_new_cigar(queryAlnLen, targetAlnLen);
        block_cigar_aa_trace_xdrop(sc.block, res.query_idx, res.reference_idx, cigar);
        const size_t n = block_len_cigar(cigar);
        size_t queryPos = 0, targetPos = 0;
        unsigned int aaIds = 0;
        for (size_t i = 0; i < n; i++) {
            const OpLen o = block_get_cigar(cigar, i);
            if (o.op == BA_M) {
                for (size_t j = 0; j < o.len; j++)
                    if (qAAs[queryPos + j] == tAAs[targetPos + j]) aaIds++;
                queryPos += o.len; targetPos += o.len;
                out.backtrace.append(o.len, 'M');
            } else if (o.op == BA_I) {
                queryPos += o.len;
                out.backtrace.append(o.len, 'I');
            } else if (o.op == BA_D) {
                targetPos += o.len;
                out.backtrace.append(o.len, 'D');
            }
        }
        std::reverse(out.backtrace.begin(), out.backtrace.end());
        out.identicalAA = aaIds;
        out.qStart = (qEnd + 1) - (int) queryPos;
        out.dbStart = (dbEnd + 1) - (int) targetPos;
        out.ok = true;
        block_free_cigar(cigar);
    }
}

} // namespace fsh
