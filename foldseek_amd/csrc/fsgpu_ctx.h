// fsgpu_ctx.h -- private: the context / database structures shared by the translation units of libfsgpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fsgpu.h"
#include "fs_kernels.h"
namespace fs {
#include "fs_selmeta.h"
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};
struct PinBuf {          // pinned host staging
    void *p = nullptr;
    size_t cap = 0;
};

struct KmerIndex;        // fsgpu_kmer.hip
struct KmerScratch;      // fsgpu_kmer.hip
void fsgpu_kmer_free_scratch(KmerScratch *s);

using namespace fs;

// Device-resident target database; shared (ref-counted) between a context and its clones.
struct DbStore {
    uint64_t n = 0, residues = 0, bytes = 0;
    bool hasAA = false;
    int maxLen = 0;
    uint4 *scan = nullptr;
    uint64_t *stripeOff = nullptr;
    uint32_t *stripeLen = nullptr;
    uint32_t *stripeTargets = nullptr;   // [nStripes][8] target id of every stripe slot (0xffffffff = empty), stripes follow the length order
    uint32_t nStripes = 0;
    uint64_t scanU4 = 0;          // size of `scan` in uint4 units
    uint8_t *aln3di = nullptr, *alnAA = nullptr;
    uint8_t *raw3di = nullptr;    // codes with the soft-mask flag (+32) kept: input of the k-mer index build
    uint64_t *dOffsets = nullptr;
    int32_t *dLengths = nullptr;
    std::vector<int32_t> hLengths;
    // gapless work lists, one per overlap class (index = overlap in 16-column chunks, 0 = whole stripes): built on first
    // use by gaplessItems() (fsgpu.hip), shared by all contexts of this DB
    std::vector<uint32_t> hStripeLen;
    struct ItemList { uint4 *items = nullptr; uint32_t n = 0; bool split = false, built = false; };
    ItemList itemLists[kGaplessMaxRUntiled + 1];
    std::mutex itemMutex;
    // multi-query scans of the contexts sharing this DB run one after the other ON THE DEVICE: a context enqueues its scan
    // launches behind the event the previous batch's owner recorded after its last scan launch (fsgpu_gapless_scan_multi).
    // The mutex only orders the enqueueing (microseconds), not the execution.
    std::mutex scanMutex;
    hipEvent_t lastScanDone = nullptr;     // owned by the context that recorded it (ctx->scanDoneEv)
    ~DbStore() {
        for (ItemList &l : itemLists) (void) hipFree(l.items);
        (void) hipFree(scan); (void) hipFree(stripeOff); (void) hipFree(stripeLen); (void) hipFree(stripeTargets);
        (void) hipFree(aln3di); (void) hipFree(alnAA); (void) hipFree(raw3di); (void) hipFree(dOffsets); (void) hipFree(dLengths);
    }
};

struct fsgpu_ctx {
    int device = 0;
    int numCU = 256;
    int gaplessBlocksPerCU = 3;                 // gapless workgroups (kGaplessBlock threads) per CU, see launchGapless
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // gapless start/stop, sw start/stop
    bool evValid[2] = {false, false};
    std::string err;

    // database (shared with clones)
    std::shared_ptr<DbStore> db;
    std::shared_ptr<KmerIndex> kidx;   // k-mer prefilter index (shared with clones)
    KmerScratch *kmer = nullptr;       // per-context k-mer prefilter scratch
    double kmerMs[12] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};   // ms of the last k-mer batch: [0] device total, [1..8] stages, [9] host tail, [10] k_kmer_lists alone
    uint32_t kmerSegs[7] = {0, 0, 0, 0, 0, 0, 0};   // last batch: [1] = [4] (query, chunk, key) runs of the duplicate stage, [3] tiles, [5] coarse keys, [6] target ids of the widest key
    double kmerHitsPerQuery = 0;       // index hits per query of the last batch (sizes the next one)
    double kmerKPerPos = 0;            // similar k-mers per query position of the last batch (picks the wave / workgroup form of the next count pass)
    int kmerBatchCap = 0;              // > 0: a batch overflowed 2^32 hits, stay at or below this many queries
    bool kmerDupAttr = false;                      // k_kmer_dup_stream may use more than 64 KB of dynamic LDS on this context's device (keys of more than 24 k target ids)
    int kmerBatchOk = 0;                           // batches that succeeded in a row under the current cap (it is relaxed after four)
    uint64_t kmerCounts[4] = {0, 0, 0, 0};   // last batch: k-mer lists probed, index hits, double-diagonal candidates, elements handed to the host

    // gapless scratch
    DevBuf pssm, scores, chunkHist, baseGt, baseTie, outId, outScore, gBorder0, gBorder1, scoreAcc;
    SelMeta *dMeta = nullptr;
    uint32_t *queue = nullptr;
    PinBuf hPssm, hImg, hTids;           // pinned staging for per-query uploads (no sync needed to reuse host vectors)
    SelMeta *hMeta = nullptr;            // pinned
    PinBuf hOutId, hOutScore;            // pinned
    int pendingMaxRes = 0;
    bool gaplessPending = false;
    // multi-query scan (fsgpu_gapless_scan_multi): per-batch arrays, query-major
    DevBuf mqPssm, mqScores, mqQueues, mqRec, mqHist, mqBaseGt, mqBaseTie, mqMeta, mqOutId, mqOutScore, mqIdent;
    PinBuf hMqPssm, hMqRec, hMqMeta, hMqOutId, hMqOutScore, hMqIdent;
    int mqLaunches = 0, mqQueries = 0;          // scan kernel launches / queries of the last batch
    double mqScanMs = -1.0;                     // >= 0: scan time of the last multi-query call incl. its row-tiled queries
    hipEvent_t swChainEv = nullptr;             // FSGPU_SW_EXCLUSIVE=1: recorded behind a k_sw3 pass that took its turn in the scan chain
    hipEvent_t scanDoneEv = nullptr;            // recorded after the last scan launch of a batch (chained through DbStore::lastScanDone)
    uint64_t mqScoreStride = 0;
    std::vector<int> mqSlot;                    // query index of the last call -> slice of mqScores (-1: went through the single-query path)

    // sw scratch
    hipStream_t swHi = nullptr;                 // highest-priority stream of the batch SW (fsgpu_sw_multi_dir_c); null: ctx->stream
    int swHiPrio = 0;
    std::vector<uint32_t> swCuMask;             // FSGPU_SW_CUS: the CU mask of the batch SW's streams (empty: priorities)
    static constexpr int kSwAux = 12;           // side streams: register-class groups (and, in the one-submission form, directions) of a multi-query launch overlap their tails
    hipStream_t swAux[kSwAux] = {};
    hipEvent_t swAuxEv[kSwAux + 1] = {};        // [kSwAux]: the fork event
    DevBuf img, tids, res0, res1, border0, border1, keys;
    // per-pass accounting of the last fsgpu_sw_multi_dir calls (fsgpu_sw_last_passes): k_sw2 launches only (single-tile queries),
    // events on the context stream around the launches of one direction (the side streams of the register classes join it)
    hipEvent_t swDirEv[4] = {nullptr, nullptr, nullptr, nullptr};   // fwd start / stop, rev start / stop
    bool swDirValid[2] = {false, false};
    double swDirCells[2] = {0, 0}, swDirPairs[2] = {0, 0}, swDirWaveSteps[2] = {0, 0};
    double swDirExtraMs[2] = {0, 0};               // device time of the row-tiled (> 1024 residues) queries' sub-call of a k_sw3 submission, outside swDirEv
    // multi-query row-tiled launches (queries longer than 512 rows inside fsgpu_sw_multi_dir): own stream, own staging
    hipStream_t swLong = nullptr;
    DevBuf lbuf, lres;                             // [ids | border bases | tile blocks per level | images], [fwd results | rev results]
    PinBuf hLbuf, hLres;
    struct LongRev { uint64_t hash = 0; std::vector<int32_t> res; };   // reversed-query results of the forward call's k_sw launches (4 int32 per pair, word 0 = not computed)
    std::vector<LongRev> swLongRev;                // per query of the last dir-0 call
    DevBuf ovAA, ovSS, ovOff, ovLen;               // explicit target sequences of fsgpu_sw_batch_seqs (instead of database entries)
    // fsgpu_sw_multi_dir_c (k_sw3): images built on the device from the compact query data; the images of a forward call are kept for the
    // reversed call over the same queries (s3Sig = hash of what the images depend on)
    DevBuf s3img, s3pass, s3build, s3res;          // images; [target ids | workgroup descriptors] of a pass; [image descriptors | matrices | query data]; results
    PinBuf hS3pass, hS3build, hS3res;
    uint64_t s3Sig = 0;
    std::vector<uint32_t> s3ImgOff;                // per query of the call the images were built for: dword offset of its images
    PinBuf hRes0, hRes1;                           // pinned result staging
    int btWgPerCU = 0;                          // fsgpu_block_backtrace_footprint: workgroups per CU of k_block_backtrace (0: default)
    DevBuf btSeq, btTrace, btBlocks, btOut, btIn;  // fsgpu_block_backtrace (k_btrace.hpp): padded reversed prefixes, trace words, block lists, [backtraces | results], inputs
    PinBuf hBtIn, hBtOut;
    struct {
        bool pending = false;
        int n = 0, L = 0, go = 0, ge = 0;
        bool hasAA = false;
        std::vector<uint32_t> tids;
        const int16_t *pAAf = nullptr, *p3f = nullptr, *pAAr = nullptr, *p3r = nullptr;
        bool explicitTargets = false;              // the batch runs on ovAA/ovSS/ovOff/ovLen, ids = 0..n-1
        std::vector<int32_t> ovLengths;            // host copy of their lengths
    } sw;
};

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
            return FSGPU_E_HIP;                                                                        \
        }                                                                                              \
    } while (0)

// Wait for the context stream by polling: hipStreamSynchronize from a non-main host thread falls back to a blocking
// wait that costs ~0.2 ms per call on this stack, more than the kernels it waits for.
// Wait for the context's stream.  Poll for a short while (a scan finishes in a few hundred microseconds and the caller
// wants the result right away), then back off to short sleeps: a host thread that waits must not burn a core -- several
// feeder threads per GPU times eight GPUs exceeds the CPU quota of a container long before it exceeds the GPUs.
// FSGPU_SPIN_US overrides the polling window (microseconds, default 40; 0 = sleep immediately).
inline int syncStreamOf(fsgpu_ctx *ctx, hipStream_t stream) {
    static const long spinUs = [] { const char *e = getenv("FSGPU_SPIN_US"); return e ? atol(e) : 40L; }();
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned polls = 0;; polls++) {
        hipError_t e = hipStreamQuery(stream);
        if (e == hipSuccess) return FSGPU_OK;
        if (e != hipErrorNotReady) { ctx->err = std::string("hipStreamQuery: ") + hipGetErrorString(e); return FSGPU_E_HIP; }
        if ((polls & 15) == 15 || spinUs == 0) {
            const long us = (long) std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            if (us >= spinUs) std::this_thread::sleep_for(std::chrono::microseconds(us < 2000 ? 20 : 100));
        }
    }
}

inline int syncStream(fsgpu_ctx *ctx) { return syncStreamOf(ctx, ctx->stream); }

inline int ensurePinned(fsgpu_ctx *ctx, PinBuf &b, size_t bytes) {
    if (b.cap >= bytes && b.p) return FSGPU_OK;
    if (b.p) { int rc = syncStream(ctx); if (rc != FSGPU_OK) return rc; (void) hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
    // grow geometrically in whole 64 KiB granules so steady-state queries never reallocate
    size_t want = ((std::max(bytes * 2, (size_t) 65536) + 65535) / 65536) * 65536;
    hipError_t e = hipHostMalloc(&b.p, want);
    if (e != hipSuccess) { ctx->err = std::string("hipHostMalloc: ") + hipGetErrorString(e); return FSGPU_E_HIP; }
    b.cap = want;
    return FSGPU_OK;
}

inline int ensure(fsgpu_ctx *ctx, DevBuf &b, size_t bytes) {
    if (b.cap >= bytes && b.p) return FSGPU_OK;
    if (b.p) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    // grow geometrically: request sizes vary from call to call (hit counts, register classes, batch composition) and
    // every hipFree/hipMalloc pair stalls the stream for milliseconds
    size_t want = std::max(bytes + bytes / 2, (size_t) 4096);
    HIPCHK(hipMalloc(&b.p, want));
    b.cap = want;
    return FSGPU_OK;
}

