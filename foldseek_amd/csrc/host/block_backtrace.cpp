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
    ~Scratch() { if (block) block_free_aa_trace_xdrop(block); }
};
thread_local Scratch g_scratch;
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
    PaddedBytes *pqAA = block_new_padded_aa(queryAlnLen, MAX_SIZE), *pq3 = block_new_padded_aa(queryAlnLen, MAX_SIZE);
    PaddedBytes *ptAA = block_new_padded_aa(targetAlnLen, MAX_SIZE), *pt3 = block_new_padded_aa(targetAlnLen, MAX_SIZE);
    PosBias *pqB = block_new_pos_bias(queryAlnLen, MAX_SIZE), *ptB = block_new_pos_bias(targetAlnLen, MAX_SIZE);
    block_set_bytes_padded_aa(pqAA, (const uint8_t *) qAAs.data(), queryAlnLen, MAX_SIZE);
    block_set_bytes_padded_aa(pq3, (const uint8_t *) q3s.data(), queryAlnLen, MAX_SIZE);
    block_set_pos_bias(pqB, qBias.data(), queryAlnLen);
    block_set_bytes_padded_aa(ptAA, (const uint8_t *) tAAs.data(), targetAlnLen, MAX_SIZE);
    block_set_bytes_padded_aa(pt3, (const uint8_t *) t3s.data(), targetAlnLen, MAX_SIZE);
    block_set_pos_bias(ptB, tBias.data(), targetAlnLen);
    // matrices keyed by letter, filled from the short substitution scores cast to int8 (:428-447)
    AAMatrix *maa = block_new_simple_aamatrix(1, -1), *m3 = block_new_simple_aamatrix(1, -1);
    for (int a = 0; a < mAA.n; a++)
        for (int b = 0; b < mAA.n; b++) block_set_aamatrix(maa, (uint8_t) mAA.letters[a], (uint8_t) mAA.letters[b], (int8_t) mAA.sub[a * mAA.n + b]);
    for (int a = 0; a < m3Di.n; a++)
        for (int b = 0; b < m3Di.n; b++) block_set_aamatrix(m3, (uint8_t) m3Di.letters[a], (uint8_t) m3Di.letters[b], (int8_t) m3Di.sub[a * m3Di.n + b]);

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
    block_free_padded_aa(pqAA); block_free_padded_aa(pq3); block_free_padded_aa(ptAA); block_free_padded_aa(pt3);
    block_free_pos_bias(pqB); block_free_pos_bias(ptB);
    block_free_aamatrix(maa); block_free_aamatrix(m3);
}

} // namespace fsh
