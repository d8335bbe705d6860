// banded_backtrace.cpp -- start position + backtrace the way StructureSmithWaterman::alignStartPosBacktrace obtains them
// (reference F/src/commons/StructureSmithWaterman.cpp:540-739: reverse striped pass with early termination, then banded_sw
// :1723-1957 with band doubling, then computerBacktrace :746-773).  In the reference this path serves profile (HMM) queries
// and the (dead) fall-back after a failed block alignment (structurealign.cpp:83-100); here it is the sequence-query form
// of the same three steps:
//   1. the reverse pass = the SAME affine SW kernels on the reversed query prefix [0, qEnd] / reversed target prefix
//      [0, dbEnd] (fsgpu_sw_batch_seqs): the first target column that reaches the forward score is the start column
//      (the reference breaks its column loop there, :1280), the smallest query row holding it the start row;
//   2. a scalar banded affine DP over the rectangle [qStart, qEnd] x [dbStart, dbEnd] with a 3-state direction matrix, band
//      |dbLen - qLen| + 1 doubled until the score is reached -- on the host, like the block aligner, for accepted hits only;
//   3. expansion of the path into the M / I / D string + the identity count.
#include "hostlib.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace fsh {

// banded_sw (:1723-1957) for substitution-matrix queries.  q* / cb* start at qStart, t* at dbStart.  Returns false when the
// trace-back meets an impossible direction code (the reference prints "Trace back error" and returns no cigar).
bool bandedBacktrace(const Matrix &mAA, const Matrix &m3Di, const uint8_t *qAA, const uint8_t *q3Di, const int8_t *cbAA, const int8_t *cbSS,
                     int qLen, const uint8_t *tAA, const uint8_t *t3Di, int dbLen, int score, int gapOpen, int gapExtend, std::string &path) {
    path.clear();
    if (qLen <= 0 || dbLen <= 0) return false;
    int band = std::abs(dbLen - qLen) + 1;
    std::vector<int32_t> hPrev, ePrev, hCur;
    std::vector<int8_t> dir;
    int64_t width = 0, widthD = 0;
    int best = 0;
    // column of the scoring matrix -> slot in one band line; (state p, column) -> slot in one direction line
    auto slotU = [](int w, int i, int j) { int x = i - w; x = x > 0 ? x : 0; return j - x + 1; };
    auto slotD = [](int w, int i, int j, int p) { int x = i - w; x = x > 0 ? x : 0; return (j - x) * 3 + p; };
    const int nA = mAA.n, n3 = m3Di.n;
    do {
        width = (int64_t) band * 2 + 3; widthD = (int64_t) band * 2 + 1;
        hPrev.assign((size_t) width + 2, 0); ePrev.assign((size_t) width + 2, 0); hCur.assign((size_t) width + 2, 0);
        dir.assign((size_t) (widthD * qLen * 3 + 8), 0);
        best = 0;
        for (int i = 0; i < qLen; i++) {
            int beg = std::max(0, i - band), end = std::min(dbLen - 1, i + band);
            const int edge = (int) std::min<int64_t>(end + 1, width - 1);
            int f = 0, u = 0;
            hPrev[0] = ePrev[0] = hPrev[edge] = ePrev[edge] = hCur[0] = 0;
            int8_t *line = dir.data() + widthD * i * 3;
            const int rowScoreBias = (int) cbAA[i] + (int) cbSS[i];
            for (int j = beg; j <= end; j++) {
                u = slotU(band, i, j);
                const int e = slotU(band, i - 1, j), b = slotU(band, i, j - 1), d = slotU(band, i - 1, j - 1);
                const int de = slotD(band, i, j, 0), df = slotD(band, i, j, 1), dh = slotD(band, i, j, 2);
                int t1 = i == 0 ? -gapOpen : hPrev[e] - gapOpen;
                int t2 = i == 0 ? -gapExtend : ePrev[e] - gapExtend;
                ePrev[u] = t1 > t2 ? t1 : t2;
                line[de] = t1 > t2 ? 3 : 2;
                t1 = hCur[b] - gapOpen;
                t2 = f - gapExtend;
                f = t1 > t2 ? t1 : t2;
                line[df] = t1 > t2 ? 5 : 4;
                const int e1 = ePrev[u] > 0 ? ePrev[u] : 0, f1 = f > 0 ? f : 0;
                t1 = e1 > f1 ? e1 : f1;
                t2 = hPrev[d] + (int) mAA.tiny[(size_t) qAA[i] * nA + tAA[j]] + (int) m3Di.tiny[(size_t) q3Di[i] * n3 + t3Di[j]] + rowScoreBias;
                hCur[u] = t1 > t2 ? t1 : t2;
                if (hCur[u] > best) best = hCur[u];
                line[dh] = t1 <= t2 ? (int8_t) 1 : (e1 > f1 ? line[de] : line[df]);
            }
            for (int j = 1; j <= u; j++) hPrev[j] = hCur[j];
        }
        band *= 2;
    } while (best < score && band / 2 < std::max(qLen, dbLen));    // the reference doubles without a bound; a band of max(qLen, dbLen) IS the rectangle
    band /= 2;
    if (best < score) return false;                                // a score the rectangle cannot reach (the reference would keep doubling until memory runs out)
    // trace back from the end cell in the H state
    int i = qLen - 1, j = dbLen - 1, state = 2;
    char op = 'M';
    const int8_t *line = dir.data() + widthD * i * 3;
    std::string rev;
    while (i > 0 || j > 0) {
        if (i < 0 || j < 0) return false;
        const int code = line[slotD(band, i, j, state)];
        switch (code) {
            case 1: --i; --j; state = 2; line -= widthD * 3; op = 'M'; break;
            case 2: --i; state = 0; line -= widthD * 3; op = 'I'; break;
            case 3: --i; state = 2; line -= widthD * 3; op = 'I'; break;
            case 4: --j; state = 1; op = 'D'; break;
            case 5: --j; state = 2; op = 'D'; break;
            default: return false;
        }
        rev.push_back(op);
    }
    // the cell (0, 0) itself: the reference closes the cigar with the pending run and, unless that run is M, one more 'M'
    rev.push_back('M');
    path.assign(rev.rbegin(), rev.rend());
    return true;
}

} // namespace fsh
