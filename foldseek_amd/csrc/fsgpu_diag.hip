// fsgpu_diag.hip -- single-diagonal rescoring of (query, target, diagonal) triples: the device half of Foldseek's
// `structurerescorediagonal` module.
//
// Reference: ungappedAlignment + ungappedAlignStructure, F/src/strucclustutils/structurerescorediagonal.cpp:23-102.
// Per pair: a Kadane scan of sub3Di[q3][t3] + subAA[qA][tA] along ONE diagonal of the forward query and one of the
// reversed query; score = fwd - rev, start/end of the forward run with the reference's tie rules (reset when the running
// sum is <= 0, a new maximum only when strictly greater).  2 x min(Lq, Lt) matrix look-ups per pair over data that is
// already resident: bytes per pair = 2 (3Di + AA) x diagonal length of the target + the query's share -- an HBM/L2
// streaming kernel, no DP state to speak of.  One lane per pair: the linclust hand-off produces a handful of pairs per
// query and millions of queries (config C5), so pairs, not positions, are the parallel axis.
//
// The reference's reverse pass for NEGATIVE diagonals passes (qRev3Di, qAA, qRevAA + dist, tAA + dist) where
// (qRev3Di, qRevAA, t3Di + dist, tAA + dist) is meant (:96-99).  It is reproduced literally -- the 3Di score reads the
// reversed query's AMINO ACIDS as 3Di states -- wherever the indices stay inside the query (dist + len <= Lq, i.e. the
// target is not longer than the query).  Beyond that the reference reads past the query into whatever its per-thread
// buffer holds from earlier queries; such pairs get status FSGPU_DIAG_UNDEFINED and no scores.
#include "fsgpu_ctx.h"

namespace {

struct DiagArgs {
    const uint8_t *tAA, *tSS;
    const uint64_t *tOff;
    const int32_t *tLen;
    const uint8_t *qAA, *qSS;
    const uint64_t *qOff;
    const int32_t *qLen;
    const int16_t *mats;          // [2][21*21]: 3Di, AA
    const fsgpu_diag_pair *pairs;
    fsgpu_diag_res *out;
    int64_t n;
    uint32_t nq;
    uint64_t nt;
};

// ungappedAlignment (:23-48): returns score; start / end through references
template <typename F>
__device__ __forceinline__ int kadane(int len, F cell, int &startPos, int &endPos) {
    int maxScore = 0, maxEndPos = 0, maxStartPos = 0, minPos = -1, score = 0;
    for (int pos = 0; pos < len; pos++) {
        score += cell(pos);
        const bool isMin = score <= 0;
        score = isMin ? 0 : score;
        minPos = isMin ? pos : minPos;
        const bool isNew = score > maxScore;
        maxEndPos = isNew ? pos : maxEndPos;
        maxStartPos = isNew ? minPos + 1 : maxStartPos;
        maxScore = isNew ? score : maxScore;
    }
    startPos = maxStartPos; endPos = maxEndPos;
    return maxScore;
}

__global__ __launch_bounds__(256) void k_diag_rescore(DiagArgs a) {
    __shared__ int16_t m3[kAlphabet * kAlphabet], mA[kAlphabet * kAlphabet];
    for (int i = threadIdx.x; i < kAlphabet * kAlphabet; i += blockDim.x) { m3[i] = a.mats[i]; mA[i] = a.mats[kAlphabet * kAlphabet + i]; }
    __syncthreads();
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const fsgpu_diag_pair p = a.pairs[i];
    fsgpu_diag_res r;
    r.score = 0; r.startPos = -1; r.endPos = -1; r.revScore = 0; r.diagonalLen = 0; r.identicalAA = 0; r.status = FSGPU_DIAG_NO_OVERLAP; r.reserved = 0;
    if (p.query >= a.nq || p.target >= a.nt) { r.status = FSGPU_DIAG_BAD_ID; a.out[i] = r; return; }
    const int Lq = a.qLen[p.query], Lt = a.tLen[p.target];
    const uint8_t *q3 = a.qSS + a.qOff[p.query], *qA = a.qAA + a.qOff[p.query];
    const uint8_t *t3 = a.tSS + a.tOff[p.target], *tA = a.tAA + a.tOff[p.target];
    const int diagonal = p.diagonal;
    const int dist = diagonal < 0 ? -diagonal : diagonal;
    int s = 0, e = 0;
    if (diagonal >= 0 && dist < Lq) {
        const int len = min(Lt, Lq - dist);
        r.diagonalLen = len;
        r.score = kadane(len, [&](int pos) { return (int) m3[q3[dist + pos] * kAlphabet + t3[pos]] + (int) mA[qA[dist + pos] * kAlphabet + tA[pos]]; }, s, e);
        r.startPos = s; r.endPos = e;
        int rs, re;   // reversed query: qRev[k] = q[Lq - 1 - k]
        r.revScore = kadane(len, [&](int pos) { const int k = Lq - 1 - (dist + pos); return (int) m3[q3[k] * kAlphabet + t3[pos]] + (int) mA[qA[k] * kAlphabet + tA[pos]]; }, rs, re);
        int id = 0;
        for (int pos = s; pos <= e; pos++) id += qA[dist + pos] == tA[pos];
        r.identicalAA = id;
        r.status = FSGPU_DIAG_OK;
    } else if (diagonal < 0 && dist < Lt) {
        const int len = min(Lt - dist, Lq);
        r.diagonalLen = len;
        r.score = kadane(len, [&](int pos) { return (int) m3[q3[pos] * kAlphabet + t3[dist + pos]] + (int) mA[qA[pos] * kAlphabet + tA[dist + pos]]; }, s, e);
        r.startPos = s; r.endPos = e;
        int id = 0;
        for (int pos = s; pos <= e; pos++) id += qA[pos] == tA[dist + pos];
        r.identicalAA = id;
        if (dist + len <= Lq) {
            // (:96-99) seq3Di1 = qRev3Di, seqAA1 = qAA (forward!), seq3Di2 = qRevAA + dist, seqAA2 = tAA + dist
            int rs, re;
            r.revScore = kadane(len, [&](int pos) { return (int) m3[q3[Lq - 1 - pos] * kAlphabet + qA[Lq - 1 - (dist + pos)]] + (int) mA[qA[pos] * kAlphabet + tA[dist + pos]]; }, rs, re);
            r.status = FSGPU_DIAG_OK;
        } else {
            r.status = FSGPU_DIAG_UNDEFINED;
        }
    }
    a.out[i] = r;
}

} // namespace

extern "C" int fsgpu_diag_rescore(fsgpu_ctx *ctx, const uint8_t *qAA, const uint8_t *q3Di, const uint64_t *qOffsets, const int32_t *qLengths, int nq,
                                  const int16_t *mat3Di, const int16_t *matAA, const fsgpu_diag_pair *pairs, int64_t n, fsgpu_diag_res *out) {
    if (!ctx) return FSGPU_E_ARG;
    if (!qAA || !q3Di || !qOffsets || !qLengths || nq < 0 || !mat3Di || !matAA || n < 0 || (n > 0 && (!pairs || !out))) { ctx->err = "fsgpu_diag_rescore: bad argument"; return FSGPU_E_ARG; }
    if (!ctx->db || ctx->db->n == 0) { ctx->err = "no database loaded"; return FSGPU_E_NODB; }
    if (!ctx->db->hasAA) { ctx->err = "fsgpu_diag_rescore needs the AA half of the database"; return FSGPU_E_NODB; }
    if (ctx->sw.pending || ctx->gaplessPending) { ctx->err = "fsgpu_diag_rescore: a scan or SW batch of this context is still in flight (it shares the scratch buffers)"; return FSGPU_E_ARG; }
    if (n == 0) return FSGPU_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t qBytes = qOffsets[nq];
    for (int i = 0; i < nq; i++)
        if (qLengths[i] < 0 || qOffsets[i] + (uint64_t) qLengths[i] > qBytes) { ctx->err = "fsgpu_diag_rescore: bad query layout"; return FSGPU_E_ARG; }
    int rc;
    // scratch: reuse the alt-alignment staging buffers for the queries, the SW id / result buffers for pairs / results
    if ((rc = ensure(ctx, ctx->ovSS, qBytes + 16)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ovAA, qBytes + 16)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ovOff, (size_t) (nq + 1) * 8)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ovLen, (size_t) nq * 4 + 16)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->img, (size_t) 2 * kAlphabet * kAlphabet * 2)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->tids, (size_t) n * sizeof(fsgpu_diag_pair))) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->res0, (size_t) n * sizeof(fsgpu_diag_res))) != FSGPU_OK) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(ctx->ovSS.p, q3Di, qBytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->ovAA.p, qAA, qBytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->ovOff.p, qOffsets, (size_t) (nq + 1) * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->ovLen.p, qLengths, (size_t) nq * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->img.p, mat3Di, (size_t) kAlphabet * kAlphabet * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy((char *) ctx->img.p + (size_t) kAlphabet * kAlphabet * 2, matAA, (size_t) kAlphabet * kAlphabet * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->tids.p, pairs, (size_t) n * sizeof(fsgpu_diag_pair), hipMemcpyHostToDevice));
    DiagArgs a;
    a.tAA = ctx->db->alnAA; a.tSS = ctx->db->aln3di; a.tOff = ctx->db->dOffsets; a.tLen = ctx->db->dLengths;
    a.qAA = (const uint8_t *) ctx->ovAA.p; a.qSS = (const uint8_t *) ctx->ovSS.p; a.qOff = (const uint64_t *) ctx->ovOff.p; a.qLen = (const int32_t *) ctx->ovLen.p;
    a.mats = (const int16_t *) ctx->img.p; a.pairs = (const fsgpu_diag_pair *) ctx->tids.p; a.out = (fsgpu_diag_res *) ctx->res0.p;
    a.n = n; a.nq = (uint32_t) nq; a.nt = ctx->db->n;
    HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
    hipLaunchKernelGGL(k_diag_rescore, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->evValid[1] = true;
    HIPCHK(hipMemcpyAsync(out, ctx->res0.p, (size_t) n * sizeof(fsgpu_diag_res), hipMemcpyDeviceToHost, ctx->stream));
    return syncStream(ctx);
}
