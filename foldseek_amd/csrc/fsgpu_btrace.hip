// fsgpu_btrace.hip -- C ABI of the device block-aligner backtrace (include/fsgpu.h: fsgpu_block_backtrace; kernel: k_btrace.hpp).
// One call = the accepted hits of one alignment batch: start position, identical-residue count and backtrace of every hit whose block stays within the
// device's LDS budget; the others come back with status 0 and take the host path (host/block_backtrace.cpp).  No CPU fallback in here.
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>

#include "fsgpu_ctx.h"
#include "k_btrace.hpp"

namespace {
inline size_t up16(size_t x) { return (x + 15) & ~(size_t) 15; }
}

extern "C" int fsgpu_block_backtrace(fsgpu_ctx *ctx, const int8_t *tblAA, const int8_t *tbl3Di, const uint8_t *letterAA, const uint8_t *letter3Di,
                                     const fsgpu_bt_query *queries, int nq, const fsgpu_bt_task *tasks, int nt, int gapOpen, int gapExtend,
                                     fsgpu_bt_res *res, const char **btBase) {
    if (!ctx || !tblAA || !tbl3Di || !letterAA || !letter3Di || nq < 0 || nt < 0 || (nt > 0 && (!queries || !tasks || !res || !btBase))) return FSGPU_E_ARG;
    if (btBase) *btBase = nullptr;
    if (nt == 0) return FSGPU_OK;
    if (!ctx->db || ctx->db->n == 0) { ctx->err = "no database loaded"; return FSGPU_E_NODB; }
    if (!ctx->db->hasAA) { ctx->err = "fsgpu_block_backtrace: the database was loaded without AA sequences"; return FSGPU_E_NODB; }
    if (!(gapOpen > gapExtend && gapExtend >= 1 && gapOpen <= 127)) { ctx->err = "fsgpu_block_backtrace: gap costs must satisfy 127 >= gapOpen > gapExtend >= 1"; return FSGPU_E_UNSUPPORTED; }
    HIPCHK(hipSetDevice(ctx->device));
    // ---- query blob: [AA L][3Di L][bias int16 L] per query, 2-byte aligned ----
    std::vector<BtQuery> hq(nq);
    size_t qBytes = 0;
    for (int i = 0; i < nq; i++) {
        if (queries[i].L <= 0 || queries[i].L > FSGPU_MAX_SEQ_LEN || !queries[i].qAA || !queries[i].q3Di || !queries[i].cbAA || !queries[i].cbSS) { ctx->err = "fsgpu_block_backtrace: bad query"; return FSGPU_E_ARG; }
        hq[i].off = (uint32_t) qBytes; hq[i].L = (uint32_t) queries[i].L;
        qBytes += up16((size_t) queries[i].L * 4);
        if (qBytes >= (1ull << 32)) { ctx->err = "fsgpu_block_backtrace: query data of one call exceeds 4 GiB"; return FSGPU_E_NOMEM; }
    }
    // ---- per-task slices of the scratch buffers ----
    std::vector<BtTask> ht(nt);
    size_t seqBytes = 0, traceWords = 0, blocks = 0, btBytes = 0;
    const std::vector<int32_t> &len = ctx->db->hLengths;
    for (int t = 0; t < nt; t++) {
        const fsgpu_bt_task &k = tasks[t];
        if (k.query >= (uint32_t) nq || k.target >= ctx->db->n || k.qEnd < 0 || k.qEnd >= queries[k.query].L || k.dbEnd < 0 || k.dbEnd >= len[k.target]) {
            ctx->err = "fsgpu_block_backtrace: task out of range"; return FSGPU_E_ARG;
        }
        const size_t qn = (size_t) k.qEnd + 1, tn = (size_t) k.dbEnd + 1;
        const size_t qStride = up16(1 + qn + kBtPad), tStride = up16(1 + tn + kBtPad);
        BtTask &d = ht[t];
        d.query = k.query; d.target = k.target; d.qEnd = k.qEnd; d.dbEnd = k.dbEnd; d.score = k.score; d.pad = 0;
        d.seqOff = seqBytes; d.traceOff = traceWords; d.blockOff = blocks; d.btOff = btBytes;
        seqBytes += 2 * qStride + 2 * tStride + 2 * qStride;
        traceWords += 2 * (size_t) (kBtMaxBlock / kBtL) * (qn + tn + 2 * kBtMaxBlock);
        blocks += qn + tn + 16;
        btBytes += up16(qn + tn + 8);
    }
    int rc;
    if ((rc = ensure(ctx, ctx->btSeq, seqBytes)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->btTrace, traceWords * 4)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->btBlocks, blocks * sizeof(uint4))) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->btOut, btBytes + (size_t) nt * sizeof(BtRes))) != FSGPU_OK) return rc;
    const size_t tblOff = up16(qBytes), taskOff = tblOff + 2048, qdOff = up16(taskOff + (size_t) nt * sizeof(BtTask)), inBytes = qdOff + (size_t) nq * sizeof(BtQuery);
    if ((rc = ensure(ctx, ctx->btIn, inBytes)) != FSGPU_OK) return rc;
    if ((rc = ensurePinned(ctx, ctx->hBtIn, inBytes)) != FSGPU_OK) return rc;
    if ((rc = ensurePinned(ctx, ctx->hBtOut, btBytes + (size_t) nt * sizeof(BtRes))) != FSGPU_OK) return rc;
    unsigned char *hb = (unsigned char *) ctx->hBtIn.p;
    for (int i = 0; i < nq; i++) {
        unsigned char *d = hb + hq[i].off;
        const int L = queries[i].L;
        memcpy(d, queries[i].qAA, L); memcpy(d + L, queries[i].q3Di, L);
        int16_t *b = (int16_t *) (d + 2 * (size_t) L);
        for (int k = 0; k < L; k++) b[k] = (int16_t) ((int) queries[i].cbAA[k] + (int) queries[i].cbSS[k]);
    }
    memcpy(hb + tblOff, tblAA, 27 * 32); memcpy(hb + tblOff + 27 * 32, tbl3Di, 27 * 32);
    memset(hb + tblOff + 2 * 27 * 32, 0, 64);
    memcpy(hb + tblOff + 2 * 27 * 32, letterAA, 21); memcpy(hb + tblOff + 2 * 27 * 32 + 32, letter3Di, 21);
    memcpy(hb + taskOff, ht.data(), (size_t) nt * sizeof(BtTask));
    memcpy(hb + qdOff, hq.data(), (size_t) nq * sizeof(BtQuery));
    hipStream_t st = ctx->stream;
    HIPCHK(hipMemcpyAsync(ctx->btIn.p, hb, inBytes, hipMemcpyHostToDevice, st));
    BtArgs a;
    const unsigned char *db = (const unsigned char *) ctx->btIn.p;
    a.tasks = (const BtTask *) (db + taskOff); a.nTasks = nt;
    a.queries = (const BtQuery *) (db + qdOff); a.qdata = db;
    a.dbAA = ctx->db->alnAA; a.dbSS = ctx->db->aln3di; a.dbOff = ctx->db->dOffsets; a.dbLen = ctx->db->dLengths;
    a.tblAA = (const int8_t *) (db + tblOff); a.tblSS = (const int8_t *) (db + tblOff + 27 * 32);
    a.letAA = db + tblOff + 2 * 27 * 32; a.letSS = a.letAA + 32;
    a.gapOpen = -gapOpen; a.gapExtend = -gapExtend;
    a.seq = (uint8_t *) ctx->btSeq.p; a.trace = (uint32_t *) ctx->btTrace.p; a.blocks = (uint4 *) ctx->btBlocks.p;
    a.bt = (char *) ctx->btOut.p; a.res = (BtRes *) ((char *) ctx->btOut.p + btBytes);
    const unsigned grid = (unsigned) std::min<size_t>(((size_t) nt + 3) / 4, (size_t) ctx->numCU * 8);
    hipLaunchKernelGGL(k_block_backtrace, dim3(grid), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ctx->hBtOut.p, ctx->btOut.p, btBytes + (size_t) nt * sizeof(BtRes), hipMemcpyDeviceToHost, st));
    if ((rc = syncStream(ctx)) != FSGPU_OK) return rc;
    const BtRes *hr = (const BtRes *) ((const char *) ctx->hBtOut.p + btBytes);
    for (int t = 0; t < nt; t++) {
        res[t].status = hr[t].status; res[t].qStart = hr[t].qStart; res[t].dbStart = hr[t].dbStart; res[t].identicalAA = hr[t].identicalAA;
        res[t].btLen = hr[t].btLen; res[t].blockSizes = hr[t].blockSizes; res[t].btOff = ht[t].btOff;
    }
    *btBase = (const char *) ctx->hBtOut.p;
    return FSGPU_OK;
}
