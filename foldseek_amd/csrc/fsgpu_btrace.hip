// fsgpu_btrace.hip -- C ABI of the device block-aligner backtrace (include/fsgpu.h: fsgpu_block_backtrace; kernel: k_btrace.hpp).
// One call = the accepted hits of one alignment batch: start position, identical-residue count and backtrace of every hit whose block stays within the
// device's LDS budget; the others come back with status 0 and take the host path (host/block_backtrace.cpp).  No CPU fallback in here.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "fsgpu_ctx.h"
#include "k_btrace.hpp"
#include "../../include/fshost.h"

namespace {
inline size_t up16(size_t x) { return (x + 15) & ~(size_t) 15; }
// scratch of a call: out of device memory is an answer (FSGPU_E_NOMEM with the thread's HIP error state cleared: the caller's host path takes the batch and
// the next launch of this thread must not find a stale error), not a failure of the search
int ensureBt(fsgpu_ctx *ctx, DevBuf &b, size_t bytes) {
    if (b.cap >= bytes && b.p) return FSGPU_OK;
    if (b.p) { HIPCHK(hipStreamSynchronize(ctx->stream)); (void) hipFree(b.p); b.p = nullptr; b.cap = 0; }
    const size_t wants[2] = {std::max(bytes + bytes / 4, (size_t) 4096), bytes};
    for (size_t want : wants) {
        const hipError_t e = hipMalloc(&b.p, want);
        if (e == hipSuccess) { b.cap = want; return FSGPU_OK; }
        b.p = nullptr;
        (void) hipGetLastError();
        if (e != hipErrorOutOfMemory) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); return FSGPU_E_HIP; }
    }
    ctx->err = "fsgpu_block_backtrace: out of device memory for the scratch of one call (" + std::to_string(bytes >> 20) + " MiB requested)";
    return FSGPU_E_NOMEM;
}
// anything enqueued on the stream is finished before an error return hands the buffers back to the caller (who falls back to the host path at once)
struct BtDrain {
    hipStream_t st; bool armed = false;
    ~BtDrain() { if (armed) { (void) hipStreamSynchronize(st); (void) hipGetLastError(); } }
};
}

extern "C" int fsgpu_block_backtrace_footprint(fsgpu_ctx *ctx, int workgroupsPerCU) {
    if (!ctx || workgroupsPerCU < 0 || workgroupsPerCU > 16) return FSGPU_E_ARG;
    ctx->btWgPerCU = workgroupsPerCU;
    return FSGPU_OK;
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
    const std::vector<int32_t> &len = ctx->db->hLengths;
    size_t btBytes = 0;
    std::vector<uint64_t> btOff(nt);
    for (int t = 0; t < nt; t++) {
        const fsgpu_bt_task &k = tasks[t];
        if (k.query >= (uint32_t) nq || k.target >= ctx->db->n || k.qEnd < 0 || k.qEnd >= queries[k.query].L || k.dbEnd < 0 || k.dbEnd >= len[k.target]) {
            ctx->err = "fsgpu_block_backtrace: task out of range"; return FSGPU_E_ARG;
        }
        btOff[t] = btBytes;
        btBytes += up16((size_t) k.qEnd + 1 + (size_t) k.dbEnd + 1 + 8);
        res[t].status = 0; res[t].qStart = -1; res[t].dbStart = -1; res[t].identicalAA = 0; res[t].btLen = 0; res[t].blockSizes = 0; res[t].btOff = btOff[t];
    }
    int rc;
    if ((rc = ensureBt(ctx, ctx->btOut, btBytes + (size_t) nt * sizeof(BtRes))) != FSGPU_OK) return rc;
    if ((rc = ensurePinned(ctx, ctx->hBtOut, btBytes + (size_t) nt * sizeof(BtRes))) != FSGPU_OK) return rc;
    // the query blob, the tables and the letter maps are the same for both passes
    const size_t tblOff = up16(qBytes), taskOff = tblOff + 2048, qdOff = up16(taskOff + (size_t) nt * sizeof(BtTask)), inBytes = qdOff + (size_t) nq * sizeof(BtQuery);
    if ((rc = ensureBt(ctx, ctx->btIn, inBytes)) != FSGPU_OK) return rc;
    if ((rc = ensurePinned(ctx, ctx->hBtIn, inBytes)) != FSGPU_OK) return rc;
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
    memcpy(hb + qdOff, hq.data(), (size_t) nq * sizeof(BtQuery));
    hipStream_t st = ctx->stream;
    BtDrain drain{st};
    // one pass over `todo` (task indices) with blocks of up to maxBlock rows: status / positions / backtraces of the tasks it answers go to res / the shared
    // backtrace buffer, the others stay at status 0
    auto runPass = [&](std::vector<int> &todo, int maxBlock) -> int {
        const int n = (int) todo.size();
        if (n == 0) return FSGPU_OK;
        // kernel order: largest first, so that the alignments of a wave (consecutive tasks) are of similar length and the longest start first
        std::stable_sort(todo.begin(), todo.end(), [&](int x, int y) { return tasks[x].qEnd + tasks[x].dbEnd > tasks[y].qEnd + tasks[y].dbEnd; });
        std::vector<BtTask> ht(n);
        size_t seqBytes = 0, traceWords = 0, blocks = 0;
        for (int t = 0; t < n; t++) {
            const fsgpu_bt_task &k = tasks[todo[t]];
            const size_t qn = (size_t) k.qEnd + 1, tn = (size_t) k.dbEnd + 1;
            const size_t qStride = up16(1 + qn + kBtPad), tStride = up16(1 + tn + kBtPad);
            BtTask &d = ht[t];
            d.query = k.query; d.target = k.target; d.qEnd = k.qEnd; d.dbEnd = k.dbEnd; d.score = k.score; d.pad = 0;
            d.seqOff = seqBytes; d.traceOff = traceWords; d.blockOff = blocks; d.btOff = btOff[todo[t]];
            seqBytes += 2 * qStride + 2 * tStride + 2 * qStride;
            traceWords += 2 * (size_t) (maxBlock / kBtL) * (qn + tn + 2 * (size_t) maxBlock);
            blocks += qn + tn + 16;
        }
        int rc2;
        if ((rc2 = ensureBt(ctx, ctx->btSeq, seqBytes)) != FSGPU_OK) return rc2;
        if ((rc2 = ensureBt(ctx, ctx->btTrace, traceWords * 4)) != FSGPU_OK) return rc2;
        if ((rc2 = ensureBt(ctx, ctx->btBlocks, blocks * sizeof(uint4))) != FSGPU_OK) return rc2;
        memcpy(hb + taskOff, ht.data(), (size_t) n * sizeof(BtTask));
        drain.armed = true;
        HIPCHK(hipMemcpyAsync(ctx->btIn.p, hb, inBytes, hipMemcpyHostToDevice, st));
        BtArgs a;
        const unsigned char *db = (const unsigned char *) ctx->btIn.p;
        a.tasks = (const BtTask *) (db + taskOff); a.nTasks = n;
        a.queries = (const BtQuery *) (db + qdOff); a.qdata = db;
        a.dbAA = ctx->db->alnAA; a.dbSS = ctx->db->aln3di; a.dbOff = ctx->db->dOffsets; a.dbLen = ctx->db->dLengths;
        a.tblAA = (const int8_t *) (db + tblOff); a.tblSS = (const int8_t *) (db + tblOff + 27 * 32);
        a.letAA = db + tblOff + 2 * 27 * 32; a.letSS = a.letAA + 32;
        a.gapOpen = -gapOpen; a.gapExtend = -gapExtend;
        a.seq = (uint8_t *) ctx->btSeq.p; a.trace = (uint32_t *) ctx->btTrace.p; a.blocks = (uint4 *) ctx->btBlocks.p;
        a.bt = (char *) ctx->btOut.p; a.res = (BtRes *) ((char *) ctx->btOut.p + btBytes);
        const size_t wgCU = ctx->btWgPerCU > 0 ? (size_t) ctx->btWgPerCU : 4;
        // fewer alignments than wave slots (3 waves per SIMD): one alignment per wave (latency form); FSGPU_BT_SPREAD = 0 / 1 forces a form of the first pass
        static const int spreadEnv = [] { const char *e = getenv("FSGPU_BT_SPREAD"); return e && *e ? atoi(e) : -1; }();
        const bool spread1 = spreadEnv >= 0 ? spreadEnv != 0 : (size_t) n <= (size_t) ctx->numCU * 12;
        if (maxBlock == kBtMaxBlock && spread1) {
            const unsigned grid = (unsigned) std::min<size_t>(((size_t) n + 3) / 4, (size_t) ctx->numCU * wgCU);
            hipLaunchKernelGGL((k_block_backtrace<kBtMaxBlock, 4, true>), dim3(grid), dim3(4 * 64), 0, st, a);
        } else if (maxBlock == kBtMaxBlock) {
            const unsigned grid = (unsigned) std::min<size_t>(((size_t) n + kBtRows - 1) / kBtRows, (size_t) ctx->numCU * wgCU);
            hipLaunchKernelGGL((k_block_backtrace<kBtMaxBlock, kBtRows>), dim3(grid), dim3(kBtRows * kBtL), 0, st, a);
        } else {
            // the second pass sees a fraction of the call's alignments (those whose block wants more than 128 rows) and restarts each from the smallest block:
            // always one per wave
            const unsigned grid = (unsigned) std::min<size_t>(((size_t) n + kBtRows2 - 1) / kBtRows2, (size_t) ctx->numCU * 2 * wgCU);
            hipLaunchKernelGGL((k_block_backtrace<kBtMaxBlock2, kBtRows2, true>), dim3(grid), dim3(kBtRows2 * 64), 0, st, a);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ctx->hBtOut.p, ctx->btOut.p, btBytes + (size_t) nt * sizeof(BtRes), hipMemcpyDeviceToHost, st));
        if ((rc2 = syncStream(ctx)) != FSGPU_OK) return rc2;
        drain.armed = false;
        const BtRes *hr = (const BtRes *) ((const char *) ctx->hBtOut.p + btBytes);
        std::vector<int> left;
        for (int t = 0; t < n; t++) {
            fsgpu_bt_res &o = res[todo[t]];
            o.blockSizes += hr[t].blockSizes;
            if (hr[t].status == 0) { left.push_back(todo[t]); continue; }
            o.status = hr[t].status; o.qStart = hr[t].qStart; o.dbStart = hr[t].dbStart; o.identicalAA = hr[t].identicalAA; o.btLen = hr[t].btLen;
        }
        todo.swap(left);
        return FSGPU_OK;
    };
    std::vector<int> todo(nt);
    for (int t = 0; t < nt; t++) todo[t] = t;
    if ((rc = runPass(todo, kBtMaxBlock)) != FSGPU_OK) return rc;
    // (the backtraces of the first pass sit in the pinned buffer; the second pass copies the whole device buffer again: its own slices are added, the others unchanged)
    // The second pass is a 3-5 ms kernel whatever it is given (the latency of its longest alignment, restarted from the smallest block): a handful of
    // alignments are back from the host's aligner sooner (36 us each on a core) -- it runs for 64 alignments per core of this GPU's share or more (FSGPU_BT_PASS2 = 1 / 0:
    // always / never); what it does not take stays at status 0 and the caller's host path answers.
    const int pass2Env = [] { const char *e = getenv("FSGPU_BT_PASS2"); return e && *e ? atoi(e) : -1; }();          // per call: the tests switch it
    const int coresPerGpu = [] {
        const char *e = getenv("FSGPU_CORES_PER_GPU");
        if (e && *e && atoi(e) > 0) return atoi(e);
        return std::max(1, fshost_usable_cores() / std::max(1, fsgpu_live_devices()));
    }();
    const bool pass2 = pass2Env >= 0 ? pass2Env != 0 : todo.size() >= (size_t) 64 * (size_t) coresPerGpu;
    if (pass2 && (rc = runPass(todo, kBtMaxBlock2)) != FSGPU_OK) return rc;
    *btBase = (const char *) ctx->hBtOut.p;
    return FSGPU_OK;
}
