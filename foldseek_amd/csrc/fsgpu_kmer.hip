// fsgpu_kmer.hip -- C ABI of the k-mer prefilter (include/fsgpu.h: fsgpu_kmer_index_build / fsgpu_kmer_search).
// Host side: HIP runtime + rocPRIM's device radix sort / scan (AMD's native primitives; the order-defining work is in
// k_kmer.hpp).  No CPU fallback: everything below fails with FSGPU_E_* when there is no device.
#include <cstring>
#include <hip/hip_runtime.h>
#include <cstddef>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

#include "fsgpu_ctx.h"
#include "k_kmer.hpp"

#include <functional>
void fshostParallelFor(int n, const std::function<void(int)> &fn);      // host/search.cpp: the library's host worker pool

struct KmerIndex {
    fsgpu_kmer_index_params p{};
    KmerPattern pat{};
    std::shared_ptr<DbStore> db;          // keeps offsets / lengths alive
    uint64_t n = 0;
    int tbits = 1;
    uint8_t *masked = nullptr;            // SequenceLookup: masked codes in the padded layout of the DB
    uint32_t *offsets = nullptr;          // 20^6 + 1, in device k-mer order (kmerDeviceIndex)
    uint32_t *bitmap = nullptr;           // 20^6 bits: list non-empty
    uint64_t *entries = nullptr;          // seqId << 16 | first position (posBits == 0) ...
    uint32_t *entries32 = nullptr;        // ... or seqId << posBits | first position when id bits + position bits <= 32 (round 5: the gather of a list moves half the bytes)
    int posBits = 0;
    uint64_t nEntries = 0;
    int16_t *s3 = nullptr;                // extended 3-mer matrix, rows sorted descending
    uint16_t *i3 = nullptr;
    // coarse keys of the hit-stream partition (k_kmer.hpp, stage 2): runs of blocks of 1024 target ids.  levels[0] balances the residues over about
    // 128 keys (what a batch of the metric's size uses); the others hold a fixed number of blocks per key (1, 2, 4 ... 64) for batches with few
    // hits per query and for tests that force a granularity
    struct CoarseLevel { uint32_t nKeys = 0, nBlk = 0, maxIds = 0, blocksPerKey = 0; uint16_t *blkKey = nullptr; uint32_t *keyFirst = nullptr; };   // k_kmer.hpp KmerCoarse
    std::vector<CoarseLevel> levels;
    uint64_t residues = 0;
    ~KmerIndex() {
        (void) hipFree(masked); (void) hipFree(offsets); (void) hipFree(bitmap); (void) hipFree(entries); (void) hipFree(entries32); (void) hipFree(s3); (void) hipFree(i3);
        for (CoarseLevel &l : levels) { (void) hipFree(l.blkKey); (void) hipFree(l.keyFirst); }
    }
};

// Coarse keys of one level (KmerCoarse): consecutive blocks of 1024 target ids are joined into keys.  blocksPerKey > 0: that many blocks per key.
// blocksPerKey == 0: about 128 keys of equal residue count (index hits are proportional to residues; a length-sorted database would otherwise
// give the keys of its long end several times the hits of the others), no key longer than twice the average number of blocks (the duplicate
// stage keeps 16 bits of LDS per target id of a key) nor than kCoarseBlocks (the low 16 bits of an id must be unique inside a key).
// Host-only, tested without a GPU through fsgpu_kmer_plan_coarse.
static void planCoarse(const int32_t *lengths, uint64_t n, uint32_t blocksPerKey, std::vector<uint16_t> &blkKey, std::vector<uint32_t> &keyFirst) {
    const uint64_t nBlk = std::max<uint64_t>(1, (n + 1023) / 1024);
    blkKey.assign(nBlk, 0); keyFirst.clear();
    if (blocksPerKey == 0) {
        std::vector<uint64_t> res(nBlk, 0);
        uint64_t R = 0;
        for (uint64_t i = 0; i < n; i++) { res[i >> 10] += (uint64_t) std::max(lengths[i], 0); R += (uint64_t) std::max(lengths[i], 0); }
        const uint64_t target = std::max<uint64_t>(1, (R + 127) / 128);
        const uint64_t cap = std::min<uint64_t>(kCoarseBlocks, std::max<uint64_t>(1, 2 * ((nBlk + 127) / 128)));
        uint64_t acc = 0, cnt = 0;
        for (uint64_t k = 0; k < nBlk; k++) {
            if (cnt > 0 && (cnt >= cap || acc + res[k] > target)) { acc = 0; cnt = 0; }
            if (cnt == 0) keyFirst.push_back((uint32_t) (k * 1024));
            blkKey[k] = (uint16_t) (keyFirst.size() - 1);
            acc += res[k]; cnt++;
        }
        if (keyFirst.size() <= (size_t) kMaxCoarse) { keyFirst.push_back((uint32_t) n); return; }
        blocksPerKey = (uint32_t) ((nBlk + kMaxCoarse - 1) / kMaxCoarse);        // too many keys (cannot happen below 33.5 M targets): equal id ranges instead
        keyFirst.clear();
    }
    for (uint64_t k = 0; k < nBlk; k++) {
        if (k % blocksPerKey == 0) keyFirst.push_back((uint32_t) (k * 1024));
        blkKey[k] = (uint16_t) (k / blocksPerKey);
    }
    keyFirst.push_back((uint32_t) n);
}

extern "C" int fsgpu_kmer_plan_coarse(const int32_t *lengths, uint64_t n, uint32_t blocksPerKey, uint16_t *blkKey /*[max(1, ceil(n / 1024))]*/, uint32_t *keyFirst /*[cap]*/, uint32_t cap) {
    if ((!lengths && n) || blocksPerKey > (uint32_t) kCoarseBlocks) return FSGPU_E_ARG;
    std::vector<uint16_t> bk; std::vector<uint32_t> kf;
    planCoarse(lengths, n, blocksPerKey, bk, kf);
    if (kf.size() > cap) return FSGPU_E_ARG;
    if (blkKey) std::copy(bk.begin(), bk.end(), blkKey);
    if (keyFirst) std::copy(kf.begin(), kf.end(), keyFirst);
    return (int) kf.size() - 1;
}

struct KmerScratch {
    DevBuf qs, posQuery, seqs, thrs, profiles, K, Kbase, listStart, listSize, listPos, listP, chunks,
           rec, recKey, recA, ordA, tileL, cntA, colA, grpSum, segCnt, segStart, tiles, candKey, candVal, candCount, ckeys, cvals, kept, score, scrA, scrB, best,
           ec, rounds, resSize, hist, thr, outCount, out, tmp, nCand, kept0, qSlot, truncHist, trunc;
    PinBuf hQs, hPosQuery, hSeqs, hThrs, hProfiles, hChunks, hEc, hRounds, hResSize, hThr, hOutCount, hOut, hMisc, hTiles;
    hipEvent_t ev[14] = {};
    bool evInit = false;
};

void fsgpu_kmer_free_scratch(KmerScratch *s) {
    if (!s) return;
    DevBuf *d[] = {&s->qs, &s->posQuery, &s->seqs, &s->thrs, &s->profiles, &s->K, &s->Kbase, &s->listStart, &s->listSize, &s->listPos, &s->listP,
                   &s->chunks, &s->rec, &s->recKey, &s->recA, &s->ordA, &s->tileL, &s->cntA, &s->colA, &s->grpSum, &s->segCnt, &s->segStart, &s->tiles,
                   &s->candKey, &s->candVal, &s->candCount, &s->ckeys, &s->cvals, &s->kept, &s->score,
                   &s->scrA, &s->scrB, &s->best, &s->ec, &s->rounds, &s->resSize, &s->hist, &s->thr, &s->outCount, &s->out, &s->tmp, &s->nCand,
                   &s->kept0, &s->qSlot, &s->truncHist, &s->trunc};
    for (DevBuf *b : d) if (b->p) (void) hipFree(b->p);
    PinBuf *h[] = {&s->hQs, &s->hPosQuery, &s->hSeqs, &s->hThrs, &s->hProfiles, &s->hChunks, &s->hEc, &s->hRounds, &s->hResSize, &s->hThr,
                   &s->hOutCount, &s->hOut, &s->hMisc, &s->hTiles};
    for (PinBuf *b : h) if (b->p) (void) hipHostFree(b->p);
    if (s->evInit) for (hipEvent_t e : s->ev) (void) hipEventDestroy(e);
    delete s;
}

#define RPCHK(call)                                                                                    \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
            return FSGPU_E_HIP;                                                                        \
        }                                                                                              \
    } while (0)

static inline unsigned gridFor(uint64_t n, unsigned block) { return (unsigned) std::max<uint64_t>(1, (n + block - 1) / block); }

// scratch grows geometrically: batch sizes differ from call to call and a hipFree/hipMalloc pair costs milliseconds
// scratch of a k-mer batch.  Out of device memory is reported as FSGPU_E_NOMEM (with the over-allocation dropped first): the caller of
// kmerBatch halves the batch instead of failing the search
static int ensureK(fsgpu_ctx *ctx, DevBuf &b, size_t bytes) {
    if (b.cap >= bytes && b.p) return FSGPU_OK;
    if (b.p) { RPCHK(hipStreamSynchronize(ctx->stream)); (void) hipFree(b.p); b.p = nullptr; b.cap = 0; }
    const size_t wants[2] = {bytes + bytes / 2 + (1u << 20), bytes};
    for (size_t want : wants) {
        const hipError_t e = hipMalloc(&b.p, want);
        if (e == hipSuccess) { b.cap = want; return FSGPU_OK; }
        b.p = nullptr;
        (void) hipGetLastError();
        if (e != hipErrorOutOfMemory) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); return FSGPU_E_HIP; }
    }
    ctx->err = "k-mer search: out of device memory for the scratch of one batch (" + std::to_string(bytes >> 20) + " MiB requested)";
    return FSGPU_E_NOMEM;
}

template <class T>
static int scanExclusive(fsgpu_ctx *ctx, DevBuf &tmp, const T *in, T *out, size_t n) {
    size_t bytes = 0;
    RPCHK(rocprim::exclusive_scan(nullptr, bytes, in, out, (T) 0, n, rocprim::plus<T>(), ctx->stream));
    int rc = ensureK(ctx, tmp, bytes);
    if (rc != FSGPU_OK) return rc;
    RPCHK(rocprim::exclusive_scan(tmp.p, bytes, in, out, (T) 0, n, rocprim::plus<T>(), ctx->stream));
    return FSGPU_OK;
}
// u32 sizes -> u64 exclusive prefix
struct U32to64 { __host__ __device__ uint64_t operator()(uint32_t v) const { return v; } };
static int scanExclusive32to64(fsgpu_ctx *ctx, DevBuf &tmp, const uint32_t *in, uint64_t *out, size_t n) {
    auto it = rocprim::make_transform_iterator(in, U32to64());
    size_t bytes = 0;
    RPCHK(rocprim::exclusive_scan(nullptr, bytes, it, out, (uint64_t) 0, n, rocprim::plus<uint64_t>(), ctx->stream));
    int rc = ensureK(ctx, tmp, bytes);
    if (rc != FSGPU_OK) return rc;
    RPCHK(rocprim::exclusive_scan(tmp.p, bytes, it, out, (uint64_t) 0, n, rocprim::plus<uint64_t>(), ctx->stream));
    return FSGPU_OK;
}
template <class K, class V>
static int sortPairs(fsgpu_ctx *ctx, DevBuf &tmp, const K *kin, K *kout, const V *vin, V *vout, size_t n, int bits) {
    size_t bytes = 0;
    RPCHK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, bits, ctx->stream));
    int rc = ensureK(ctx, tmp, bytes);
    if (rc != FSGPU_OK) return rc;
    RPCHK(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, n, 0, bits, ctx->stream));
    return FSGPU_OK;
}

static int bitsFor(uint64_t n) { int b = 1; while (b < 32 && (1ull << b) < n) b++; return b; }

extern "C" {

uint64_t fsgpu_kmer_index_entries(const fsgpu_ctx *ctx) { return ctx && ctx->kidx ? ctx->kidx->nEntries : 0; }

int fsgpu_kmer_index_build(fsgpu_ctx *ctx, const fsgpu_kmer_index_params *p, const int16_t *kmerSub) {
    if (!ctx || !p || !kmerSub) return FSGPU_E_ARG;
    if (!ctx->db || !ctx->db->raw3di) { ctx->err = "no database loaded"; return FSGPU_E_NODB; }
    if (p->kmerSize != 6) { ctx->err = "k-mer prefilter: only k = 6 is implemented on the device"; return FSGPU_E_UNSUPPORTED; }
    // UngappedAlignment switches to computeLongScore (wrapped 16-bit diagonals) from 32768 residues on (UngappedAlignment.cpp:198,295-312)
    if (ctx->db->maxLen >= 32768) { ctx->err = "k-mer prefilter: targets of 32768 residues or more are not supported on the device"; return FSGPU_E_UNSUPPORTED; }
    RPCHK(hipSetDevice(ctx->device));
    std::shared_ptr<KmerIndex> ix = std::make_shared<KmerIndex>();
    ix->p = *p; ix->db = ctx->db; ix->n = ctx->db->n;
    ix->tbits = bitsFor(std::max<uint64_t>(ix->n, 2));
    static const int s6[10] = {1, 1, 0, 1, 0, 1, 0, 0, 1, 1};
    ix->pat.size = p->spaced ? 10 : 6;
    for (int i = 0, k = 0; i < ix->pat.size; i++) if (!p->spaced || s6[i]) ix->pat.pos[k++] = i;
    const DbStore &db = *ctx->db;
    const uint64_t n = db.n, bytes = std::max<uint64_t>(db.bytes, 1);
    const uint64_t tableSize = 64000000ull;

    // extended 3-mer matrix
    int16_t *dSub = nullptr;
    struct SubGuard { int16_t *&p; bool armed = true; ~SubGuard() { if (armed) (void) hipFree(p); } } subGuard{dSub};   // early returns below must not leak it
    RPCHK(hipMalloc((void **) &dSub, 441 * sizeof(int16_t)));
    RPCHK(hipMemcpy(dSub, kmerSub, 441 * sizeof(int16_t), hipMemcpyHostToDevice));
    RPCHK(hipMalloc((void **) &ix->s3, (size_t) kRow3 * kRow3 * sizeof(int16_t)));
    RPCHK(hipMalloc((void **) &ix->i3, (size_t) kRow3 * kRow3 * sizeof(uint16_t)));
    hipLaunchKernelGGL(k_kmer_rows3, dim3(kRow3), dim3(1024), 0, ctx->stream, dSub, ix->s3, ix->i3);
    RPCHK(hipGetLastError());

    // masked lookup
    RPCHK(hipMalloc((void **) &ix->masked, bytes + 16));          // + 16: k_kmer_score8 reads whole 8-byte words around a diagonal
    RPCHK(hipMemsetAsync(ix->masked, 20, bytes, ctx->stream));
    if (n) {
        hipLaunchKernelGGL(k_kmer_mask, dim3(gridFor(n, 128)), dim3(128), 0, ctx->stream, db.raw3di, db.dOffsets, db.dLengths, n,
                           p->maskLowerCase, p->maskNrepeats, ix->masked);
        RPCHK(hipGetLastError());
    }
    // k-mer extraction in (seqId, pos) order
    std::vector<uint64_t> resOff(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) resOff[i + 1] = resOff[i] + (uint64_t) db.hLengths[i];
    const uint64_t R = resOff[n];
    if (R >= 0xFFFFFFF0ull) { ctx->err = "k-mer index: more than 2^32 residues"; return FSGPU_E_UNSUPPORTED; }
    RPCHK(hipMalloc((void **) &ix->offsets, (tableSize + 1) * sizeof(uint32_t)));
    RPCHK(hipMemsetAsync(ix->offsets, 0, (tableSize + 1) * sizeof(uint32_t), ctx->stream));
    uint64_t *dResOff = nullptr, *v0 = nullptr, *v1 = nullptr;
    uint32_t *k0 = nullptr, *k1 = nullptr, *flags = nullptr, *scan = nullptr;
    int8_t *dSelf = nullptr;
    DevBuf tmp;
    int rc = FSGPU_OK;
    subGuard.armed = false;           // from here on cleanup() owns dSub
    auto cleanup = [&]() {
        (void) hipFree(dResOff); (void) hipFree(v0); (void) hipFree(v1); (void) hipFree(k0); (void) hipFree(k1);
        (void) hipFree(flags); (void) hipFree(scan); (void) hipFree(dSelf); (void) hipFree(tmp.p); (void) hipFree(dSub);
    };
#define IXCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); cleanup(); return FSGPU_E_HIP; } } while (0)
    if (R > 0) {
        int8_t self[21];
        for (int a = 0; a < 21; a++) self[a] = (int8_t) kmerSub[a * 21 + a];
        IXCHK(hipMalloc((void **) &dSelf, 32));
        IXCHK(hipMemcpy(dSelf, self, 21, hipMemcpyHostToDevice));
        IXCHK(hipMalloc((void **) &dResOff, (n + 1) * sizeof(uint64_t)));
        IXCHK(hipMemcpy(dResOff, resOff.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        IXCHK(hipMalloc((void **) &k0, R * sizeof(uint32_t)));
        IXCHK(hipMalloc((void **) &k1, R * sizeof(uint32_t)));
        IXCHK(hipMalloc((void **) &v0, R * sizeof(uint64_t)));
        IXCHK(hipMalloc((void **) &v1, R * sizeof(uint64_t)));
        IXCHK(hipMalloc((void **) &flags, (R + 1) * sizeof(uint32_t)));
        IXCHK(hipMalloc((void **) &scan, (R + 1) * sizeof(uint32_t)));
        hipLaunchKernelGGL(k_kmer_extract, dim3((unsigned) std::min<uint64_t>(n, 65535 * 16)), dim3(256), 0, ctx->stream, ix->masked, db.dOffsets, db.dLengths,
                           dResOff, n, ix->pat, p->kmerThr, dSelf, k0, v0);
        IXCHK(hipGetLastError());
        // stable sort by k-mer keeps the (seqId, pos) order inside every k-mer: IndexEntryLocalTmp::comapreByIdAndPos
        rc = sortPairs(ctx, tmp, k0, k1, v0, v1, R, 26);
        if (rc != FSGPU_OK) { cleanup(); return rc; }
        IXCHK(hipMemsetAsync(flags + R, 0, sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(k_kmer_unique_flags, dim3(gridFor(R, 256)), dim3(256), 0, ctx->stream, k1, v1, R, flags, ix->offsets);
        IXCHK(hipGetLastError());
        rc = scanExclusive<uint32_t>(ctx, tmp, flags, scan, R + 1);
        if (rc != FSGPU_OK) { cleanup(); return rc; }
        uint32_t ne = 0;
        IXCHK(hipMemcpyAsync(&ne, scan + R, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        IXCHK(hipStreamSynchronize(ctx->stream));
        ix->nEntries = ne;
        {   // 4-byte entries while a target id and a position fit 32 bits together (FSGPU_KMER_ENTRY64=1 keeps the 8-byte form: A/B runs, tests)
            const int pbits = bitsFor((uint64_t) std::max(db.maxLen, 1) + 1);
            const char *e64 = getenv("FSGPU_KMER_ENTRY64");
            ix->posBits = (ix->tbits + pbits <= 32 && pbits <= 16 && !(e64 && atoi(e64) != 0)) ? pbits : 0;
        }
        if (ix->posBits) {
            IXCHK(hipMalloc((void **) &ix->entries32, std::max<uint64_t>(ne, 1) * sizeof(uint32_t)));
            hipLaunchKernelGGL(k_kmer_compact_entries32, dim3(gridFor(R, 256)), dim3(256), 0, ctx->stream, v1, flags, scan, R, ix->posBits, ix->entries32);
        } else {
            IXCHK(hipMalloc((void **) &ix->entries, std::max<uint64_t>(ne, 1) * sizeof(uint64_t)));
            hipLaunchKernelGGL(k_kmer_compact_entries, dim3(gridFor(R, 256)), dim3(256), 0, ctx->stream, v1, flags, scan, R, ix->entries);
        }
        IXCHK(hipGetLastError());
    } else {
        IXCHK(hipMalloc((void **) &ix->entries, sizeof(uint64_t)));
    }
    // counts -> offsets (in place)
    rc = scanExclusive<uint32_t>(ctx, tmp, ix->offsets, ix->offsets, tableSize + 1);
    if (rc != FSGPU_OK) { cleanup(); return rc; }
    IXCHK(hipMalloc((void **) &ix->bitmap, (tableSize / 32) * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_kmer_bitmap, dim3(gridFor(tableSize / 32, 256)), dim3(256), 0, ctx->stream, ix->offsets, (uint32_t) (tableSize / 32), ix->bitmap);
    IXCHK(hipGetLastError());
    // coarse-key levels of the hit-stream partition
    {
        ix->residues = R;
        if ((n + 1023) / 1024 > (uint64_t) kMaxCoarse * kCoarseBlocks) {
            ctx->err = "k-mer prefilter: more than " + std::to_string((uint64_t) kMaxCoarse * kCoarseBlocks * 1024) + " targets are not supported by the device hit-stream partition";
            cleanup(); return FSGPU_E_UNSUPPORTED;
        }
        std::vector<uint16_t> bk; std::vector<uint32_t> kf;
        uint32_t prevKeys = 0;
        for (uint32_t bpk : {0u, 1u, 2u, 4u, 8u, 16u, 32u, 64u}) {
            planCoarse(db.hLengths.data(), n, bpk, bk, kf);
            const uint32_t nk = (uint32_t) kf.size() - 1;
            if (nk > (uint32_t) kMaxCoarse || (bpk > 1 && nk == prevKeys)) continue;           // (a coarser level with the same keys is the same level)
            prevKeys = bpk ? nk : 0;
            ix->levels.emplace_back();                         // owned by the index from here on (freed by its destructor)
            KmerIndex::CoarseLevel &lv = ix->levels.back();
            lv.nKeys = nk; lv.nBlk = (uint32_t) bk.size(); lv.blocksPerKey = bpk;
            for (uint32_t k = 0; k < nk; k++) lv.maxIds = std::max(lv.maxIds, kf[k + 1] - kf[k]);
            IXCHK(hipMalloc((void **) &lv.blkKey, bk.size() * sizeof(uint16_t)));
            IXCHK(hipMalloc((void **) &lv.keyFirst, kf.size() * sizeof(uint32_t)));
            IXCHK(hipMemcpy(lv.blkKey, bk.data(), bk.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            IXCHK(hipMemcpy(lv.keyFirst, kf.data(), kf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        if (ix->levels.empty()) { ctx->err = "k-mer prefilter: no coarse-key level fits the device partition"; cleanup(); return FSGPU_E_UNSUPPORTED; }
    }
    IXCHK(hipStreamSynchronize(ctx->stream));
    cleanup();
#undef IXCHK
    ctx->kidx = ix;
    return FSGPU_OK;
}

// Test / inspection accessors: copy parts of the resident index to host memory (any pointer may be NULL).
int fsgpu_kmer_index_copy(fsgpu_ctx *ctx, uint32_t *offsets /*64e6+1*/, uint64_t *entries /*nEntries*/, uint8_t *masked /*db bytes*/) {
    if (!ctx) return FSGPU_E_ARG;
    if (!ctx->kidx) { ctx->err = "k-mer index not built"; return FSGPU_E_NODB; }
    const KmerIndex &ix = *ctx->kidx;
    if (offsets) RPCHK(hipMemcpy(offsets, ix.offsets, (64000000ull + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (entries && ix.nEntries) {
        if (ix.posBits) {             // the accessor's format stays seqId << 16 | position
            uint64_t *wide = nullptr;
            RPCHK(hipMalloc((void **) &wide, ix.nEntries * sizeof(uint64_t)));
            hipLaunchKernelGGL(k_kmer_widen_entries, dim3(gridFor(ix.nEntries, 256)), dim3(256), 0, ctx->stream, ix.entries32, ix.nEntries, ix.posBits, wide);
            hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(ctx->stream);
            hipError_t e3 = hipMemcpy(entries, wide, ix.nEntries * sizeof(uint64_t), hipMemcpyDeviceToHost);
            (void) hipFree(wide);
            RPCHK(e1); RPCHK(e2); RPCHK(e3);
        } else RPCHK(hipMemcpy(entries, ix.entries, ix.nEntries * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    if (masked && ix.db->bytes) RPCHK(hipMemcpy(masked, ix.masked, ix.db->bytes, hipMemcpyDeviceToHost));
    return FSGPU_OK;
}
int fsgpu_kmer_batch_hint(const fsgpu_ctx *ctx) {
    if (!ctx || !ctx->kidx || ctx->kmerHitsPerQuery <= 0) return 32;
    const int tb = ctx->kidx->tbits;
    const int maxBatch = std::max(1, std::min(1 << std::min(12, 32 - tb), 1024));
    int b = (int) std::min<double>(maxBatch, 2.4e8 / ctx->kmerHitsPerQuery);
    if (ctx->kmerBatchCap > 0) b = std::min(b, ctx->kmerBatchCap);
    return std::max(32, b / 32 * 32);
}
void fsgpu_kmer_last_segments(const fsgpu_ctx *ctx, uint32_t *out7) {
    for (int i = 0; i < 7; i++) out7[i] = ctx ? ctx->kmerSegs[i] : 0;
}
void fsgpu_kmer_last_counts(const fsgpu_ctx *ctx, uint64_t *out4) {
    for (int i = 0; i < 4; i++) out4[i] = ctx ? ctx->kmerCounts[i] : 0;
}
int fsgpu_kmer_row_copy(fsgpu_ctx *ctx, int row, int16_t *score /*8000*/, uint16_t *index /*8000*/) {
    if (!ctx || row < 0 || row >= kRow3) return FSGPU_E_ARG;
    if (!ctx->kidx) { ctx->err = "k-mer index not built"; return FSGPU_E_NODB; }
    RPCHK(hipMemcpy(score, ctx->kidx->s3 + (size_t) row * kRow3, kRow3 * sizeof(int16_t), hipMemcpyDeviceToHost));
    RPCHK(hipMemcpy(index, ctx->kidx->i3 + (size_t) row * kRow3, kRow3 * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return FSGPU_OK;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------------------------------
namespace {

struct HostOut { uint32_t id, count, diag; int32_t score; uint64_t g; };
static_assert(sizeof(HostOut) == sizeof(KmerOut), "layout");

unsigned pickBins(const fsgpu_kmer_search_params &p, uint64_t n) {
    if (p.bins) return (unsigned) p.bins;
    uint64_t l2 = p.l2CacheSize;
    if (l2 == 0) {
        long v = sysconf(_SC_LEVEL2_CACHE_SIZE);      // Util::getL2CacheSize (M/src/commons/Util.cpp:346-361)
        l2 = v > 0 ? (uint64_t) v : 262144;
    }
    for (unsigned x = 2; x <= 1024; x *= 2) if (n / x < l2) return x;
    return 2048;
}

int scalarDiag(const int8_t *profile, int len, const uint8_t *db) {
    int mx = 0, s = 0;
    for (int pos = 0; pos < len; pos++) { s += profile[pos * 21 + db[pos]]; s = s < 0 ? 0 : s; mx = s > mx ? s : mx; }
    return mx;
}

// The tail of QueryMatcher::matchQuery (:146-239) on the elements at or above the cut: array order of the reference
// = (bin = id & (B-1), then the order the overflow rounds left the elements in), radix by score, getResult, final sort.
// findDuplicates output capacity (CacheFriendlyOperations.cpp:217-219), conservative form: true when some chunk's candidates plus the elements the earlier
// rounds left reach foundDiagonalsSize, i.e. when the reference MAY have cut that chunk short (the exact, per-bin test is replayOutputTruncation's)
bool outputTestFires(const fsgpu_kmer_search_params &sp, uint64_t n, const KmerChunks &ck, const uint32_t *ec, const uint32_t *rounds) {
    const uint64_t foundSize = sp.foundDiagonalsSize ? (uint64_t) sp.foundDiagonalsSize : std::max<uint64_t>(n, 1000000);
    const uint32_t C = ck.nChunks - 1;
    const bool lastEmpty = ck.start[ck.nChunks] == ck.start[ck.nChunks - 1];
    if (ck.aborted != 0 || (C >= 1 && lastEmpty)) return false;          // answered empty / with another status
    for (uint32_t c = 0; c <= C; c++)
        if ((uint64_t) ec[c] + (c == 0 ? 0 : rounds[c]) >= foundSize) return true;
    return false;
}

int finishQuery(const fsgpu_kmer_search_params &sp, uint64_t n, const fsgpu_kmer_query &q, const KmerChunks &ck, const uint32_t *ec, const uint32_t *rounds,
                uint64_t resultSize, uint32_t thr, std::vector<HostOut> &el, fsgpu_kmer_hit *out, int32_t *nout, bool truncationReplayed) {
    const uint64_t big = std::max<uint64_t>(n, 1000000);
    const uint64_t foundSize = sp.foundDiagonalsSize ? (uint64_t) sp.foundDiagonalsSize : big;
    const size_t maxHits = (size_t) std::min<uint64_t>((uint64_t) sp.maxResListLen, n);
    const unsigned B = pickBins(sp, n);
    const uint32_t C = ck.nChunks - 1;
    int status = FSGPU_KMER_OK;
    bool empty = false;
    if (ck.aborted == 2) status = FSGPU_KMER_E_CHUNKS;
    if (ck.aborted == 1) empty = true;
    const bool lastEmpty = ck.start[ck.nChunks] == ck.start[ck.nChunks - 1];
    if (C >= 1 && lastEmpty) empty = true;
    // findDuplicates output capacity (CacheFriendlyOperations.cpp:217-219): conservative replay
    for (uint32_t c = 0; c <= C && status == FSGPU_KMER_OK && !empty && !truncationReplayed; c++) {
        if (c == C && lastEmpty) break;
        const uint64_t before = c == 0 ? 0 : rounds[c];
        if ((uint64_t) ec[c] + before >= foundSize) status = FSGPU_KMER_E_OUTPUT;
    }
    size_t cur = 0;
    // getResult: the query's own entry first, with the top raw score of the mode (UCHAR_MAX for KMER_SCORE, USHRT_MAX otherwise; QueryMatcher.cpp:409-421)
    if (q.identity >= 0 && maxHits > 0) { out[0].id = (uint32_t) q.identity; out[0].score = sp.kmerScoreOnly ? 255 : 65535; out[0].diagonal = 0; out[0].pad = 0; cur = 1; }
    if (status < 0) { *nout = 0; return status; }
    if (!empty && resultSize >= foundSize / 2) status = FSGPU_KMER_UNSTABLE;
    if (!empty && !el.empty()) {
        // order of the chunk runs in foundDiagonals after the overflow rounds (k-mer prefilter)
        std::vector<int> rank(C + 1, 0), dir(C + 1, 1);
        {
            std::vector<std::pair<int, int>> order;       // (chunk, direction)
            order.push_back({0, 1});
            for (uint32_t j = 2; j <= C; j++) {
                std::vector<std::pair<int, int>> nx;
                nx.push_back({(int) j - 1, -1});
                for (size_t z = order.size(); z-- > 0;) nx.push_back({order[z].first, -order[z].second});
                order.swap(nx);
            }
            if (C >= 1) order.push_back({(int) C, 1});
            for (size_t z = 0; z < order.size(); z++) { rank[order[z].first] = (int) z; dir[order[z].first] = order[z].second; }
            // --diag-score 0: mergeScoreDuplicates hands every element on in the order it came in -- the earlier rounds' elements, then the new
            // chunk's -- so a bin holds its elements by round of origin, each round's in arrival order of the first candidate
            if (sp.kmerScoreOnly) for (uint32_t c = 0; c <= C; c++) { rank[c] = (int) c; dir[c] = 1; }
        }
        auto chunkOf = [&](uint64_t g) { uint32_t c = 0; while (c + 1 < ck.nChunks && ck.start[c + 1] <= g) c++; return c; };
        // array order of the reference inside one score bucket: (bin, order the overflow rounds left the elements in)
        auto orderKey = [&](const HostOut &e, uint32_t &bin) {
            const uint32_t c = chunkOf(e.g);
            bin = e.id & (B - 1);
            return ((uint64_t) rank[c] << 40) | (dir[c] > 0 ? e.g : ((1ull << 40) - 1 - e.g));
        };
        struct Key { uint32_t count; uint32_t bin; uint64_t ok; const HostOut *e; };
        auto arrayOrder = [](const Key &a, const Key &b) { return a.bin != b.bin ? a.bin < b.bin : a.ok < b.ok; };
        const bool truncated = thr >= 255 && !sp.kmerScoreOnly;      // rescoreHits belongs to the diagonal-score mode only
        if (status == FSGPU_KMER_UNSTABLE) {
            // resultSize >= foundDiagonalsSize/2 (QueryMatcher.cpp:205-215): the reference compacts the elements at or above
            // the cut in array order and orders them with std::sort(sortScore), which is not stable.  The same call on the
            // same sequence reproduces its permutation wherever both sides use libstdc++'s introsort.
            std::vector<Key> ks(el.size());
            for (size_t i = 0; i < el.size(); i++) { uint32_t bin; const uint64_t ok = orderKey(el[i], bin); ks[i] = {el[i].count, bin, ok, &el[i]}; }
            std::sort(ks.begin(), ks.end(), arrayOrder);
            std::sort(ks.begin(), ks.end(), [](const Key &a, const Key &b) { return a.count > b.count; });
            for (size_t i = 0; i < ks.size() && cur < maxHits; i++) {
                if (q.identity >= 0 && (uint32_t) q.identity == ks[i].e->id) continue;
                fsgpu_kmer_hit &h = out[cur++];
                h.id = ks[i].e->id; h.diagonal = (uint16_t) ks[i].e->diag; h.pad = 0;
                h.score = ks[i].e->count >= 255 ? ks[i].e->score : (int32_t) ks[i].e->count;
            }
        } else if (truncated) {
            // rescoreHits (QueryMatcher.cpp:563-589): only the 255-capped hits survive, re-ranked by their real score
            std::vector<Key> ks(el.size());
            for (size_t i = 0; i < el.size(); i++) { uint32_t bin; const uint64_t ok = orderKey(el[i], bin); ks[i] = {el[i].count, bin, ok, &el[i]}; }
            std::sort(ks.begin(), ks.end(), arrayOrder);
            std::vector<uint8_t> clean(q.L);       // Sequence::numSequence of the query: soft-mask flag dropped
            for (int i = 0; i < q.L; i++) { uint8_t c = q.seq[i]; c = c >= 32 ? c - 32 : c; clean[i] = c > 20 ? 20 : c; }
            int maxSelf = scalarDiag(q.profile, q.L, clean.data()) - 255;
            maxSelf = std::max(1, maxSelf);
            maxSelf = std::min(maxSelf, 65535);
            const float fmax = (float) maxSelf;
            for (Key &k : ks) {
                unsigned ns = (unsigned) k.e->score - 255u;
                float sc = (float) std::min(ns, 65535u);
                k.count = (uint8_t) ((sc / fmax) * (float) 255 + 0.5);
            }
            const unsigned rescale = (unsigned) maxSelf;
            std::stable_sort(ks.begin(), ks.end(), [](const Key &a, const Key &b) { return a.count > b.count; });
            for (size_t i = 0; i < ks.size() && cur < maxHits; i++) {
                if (q.identity >= 0 && (uint32_t) q.identity == ks[i].e->id) continue;
                fsgpu_kmer_hit &h = out[cur++];
                h.id = ks[i].e->id; h.diagonal = (uint16_t) ks[i].e->diag; h.pad = 0;
                h.score = (int32_t) (255u + ks[i].count * rescale / 255u);
            }
        } else {
            // everything above the cut is taken (computeScoreThreshold leaves fewer than maxHits of them); the ties at the
            // cut fill the remaining slots in the reference's array order, so only they need the (bin, order) key
            std::vector<Key> ties;
            for (const HostOut &e : el) {
                if (q.identity >= 0 && (uint32_t) q.identity == e.id) continue;
                if (e.count > thr) {
                    if (cur < maxHits) {
                        fsgpu_kmer_hit &h = out[cur++];
                        h.id = e.id; h.diagonal = (uint16_t) e.diag; h.pad = 0;
                        h.score = e.count >= 255 ? e.score : (int32_t) e.count;
                    }
                } else if (e.count == thr) {
                    uint32_t bin; const uint64_t ok = orderKey(e, bin);
                    ties.push_back({e.count, bin, ok, &e});
                }
            }
            const size_t room = maxHits - cur;
            if (ties.size() > room) { std::nth_element(ties.begin(), ties.begin() + room, ties.end(), arrayOrder); ties.resize(room); }
            for (const Key &k : ties) {
                fsgpu_kmer_hit &h = out[cur++];
                h.id = k.e->id; h.diagonal = (uint16_t) k.e->diag; h.pad = 0;
                h.score = k.e->count >= 255 ? k.e->score : (int32_t) k.e->count;
            }
        }
    }
    if (cur > 1) {
        auto cmp = [](const fsgpu_kmer_hit &a, const fsgpu_kmer_hit &b) {
            const int aa = abs(a.score), bb = abs(b.score);
            return aa != bb ? aa > bb : a.id < b.id;
        };
        std::sort(out + (q.identity >= 0 ? 1 : 0), out + cur, cmp);
    }
    *nout = (int32_t) cur;
    return status;
}

} // namespace

constexpr double kKmerHitBudget = 2.4e8;          // index hits of one device batch (24 B of scratch each)

static int kmerBatch(fsgpu_ctx *ctx, const fsgpu_kmer_search_params &sp, const fsgpu_kmer_query *queries, int nq,
                     fsgpu_kmer_hit *out, int32_t *nout, int32_t *status, double *stats) {
    KmerIndex &ix = *ctx->kidx;
    if (!ctx->kmer) ctx->kmer = new KmerScratch();
    KmerScratch &S = *ctx->kmer;
    if (!S.evInit) { for (hipEvent_t &e : S.ev) RPCHK(hipEventCreate(&e)); S.evInit = true; }
    const DbStore &db = *ix.db;
    const uint64_t n = ix.n;
    const int tbits = ix.tbits;
    const uint64_t big = std::max<uint64_t>(n, 1000000);
    const uint64_t maxDbMatches = sp.maxDbMatches ? (uint64_t) sp.maxDbMatches : big * 2;
    const uint32_t maxHits = (uint32_t) std::min<uint64_t>((uint64_t) sp.maxResListLen, n);
    int rc;
#define CHK(x) do { rc = (x); if (rc == FSGPU_E_NOMEM && nq > 1) { (void) hipStreamSynchronize(ctx->stream); return 1; } if (rc != FSGPU_OK) return rc; } while (0)   // 1: the caller halves the batch; nothing of this attempt may still read the staging buffers
    // FSGPU_KMER_TRACE=1: host wall clock of the phases of a batch on stderr (where a feeder thread's time goes between the device stages)
    static const bool trace = getenv("FSGPU_KMER_TRACE") != nullptr;
    auto tPrev = std::chrono::steady_clock::now();
    std::string traceLine;
    auto mark = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        char b[64];
        snprintf(b, sizeof(b), " %s %.3f", what, std::chrono::duration<double, std::milli>(now - tPrev).count());
        traceLine += b; tPrev = now;
    };

    // ---- stage the batch ------------------------------------------------------------------------------------
    uint64_t nPos = 0, seqBytes = 0, profBytes = 0;
    for (int q = 0; q < nq; q++) {
        const int L = queries[q].L;
        if (L < 0 || L > FSGPU_MAX_SEQ_LEN || (L > 0 && (!queries[q].seq || !queries[q].profile))) { ctx->err = "k-mer search: bad query"; return FSGPU_E_ARG; }
        if (L >= 32768) { ctx->err = "k-mer search: queries of 32768 residues or more are not supported on the device"; return FSGPU_E_UNSUPPORTED; }
        nPos += (uint64_t) std::max(0, L - ix.pat.size + 1);
        seqBytes += (uint64_t) L + 16;
        profBytes += (uint64_t) L * 21 + 16;
    }
    CHK(ensurePinned(ctx, S.hQs, (size_t) nq * sizeof(KmerQ)));
    CHK(ensurePinned(ctx, S.hPosQuery, (nPos + 1) * sizeof(uint16_t)));
    CHK(ensurePinned(ctx, S.hSeqs, seqBytes));
    CHK(ensurePinned(ctx, S.hThrs, (nPos + 1) * sizeof(int16_t)));
    CHK(ensurePinned(ctx, S.hProfiles, profBytes));
    KmerQ *hq = (KmerQ *) S.hQs.p;
    {
        uint64_t pb = 0, so = 0, po = 0;
        for (int q = 0; q < nq; q++) {
            const int L = queries[q].L;
            const uint32_t np = (uint32_t) std::max(0, L - ix.pat.size + 1);
            hq[q].posBase = (uint32_t) pb; hq[q].nPos = np; hq[q].seqOff = (uint32_t) so; hq[q].L = (uint32_t) L; hq[q].profOff = (uint32_t) po;
            hq[q].pad = 0; hq[q].hitBase = 0; hq[q].listBase = 0;
            uint8_t *sd = (uint8_t *) S.hSeqs.p + so;
            for (int i = 0; i < L; i++) { uint8_t c = queries[q].seq[i]; c = c >= 32 ? c - 32 : c; sd[i] = c > 20 ? 20 : c; }
            memset(sd + L, 20, 16);
            if (L) memcpy((int8_t *) S.hProfiles.p + po, queries[q].profile, (size_t) L * 21);
            for (uint32_t i = 0; i < np; i++) { ((uint16_t *) S.hPosQuery.p)[pb + i] = (uint16_t) q; ((int16_t *) S.hThrs.p)[pb + i] = queries[q].kmerThr ? queries[q].kmerThr[i] : (int16_t) ix.p.kmerThr; }
            pb += np; so += (uint64_t) L + 16; po += (uint64_t) L * 21 + 16;
        }
    }
    CHK(ensureK(ctx, S.qs, (size_t) nq * sizeof(KmerQ)));
    CHK(ensureK(ctx, S.posQuery, (nPos + 1) * sizeof(uint16_t)));
    CHK(ensureK(ctx, S.seqs, seqBytes));
    CHK(ensureK(ctx, S.thrs, (nPos + 1) * sizeof(int16_t)));
    CHK(ensureK(ctx, S.profiles, profBytes));
    CHK(ensureK(ctx, S.K, (nPos + 1) * sizeof(uint32_t)));
    CHK(ensureK(ctx, S.Kbase, (nPos + 1) * sizeof(uint64_t)));
    CHK(ensureK(ctx, S.chunks, (size_t) nq * sizeof(KmerChunks)));
    CHK(ensureK(ctx, S.ec, (size_t) nq * kMaxChunks * sizeof(uint32_t)));
    CHK(ensureK(ctx, S.rounds, (size_t) nq * kMaxChunks * sizeof(uint32_t)));
    CHK(ensureK(ctx, S.resSize, (size_t) nq * sizeof(uint64_t)));
    CHK(ensureK(ctx, S.hist, (size_t) nq * 256 * sizeof(uint32_t)));
    CHK(ensureK(ctx, S.thr, (size_t) nq * sizeof(uint32_t)));
    CHK(ensureK(ctx, S.outCount, (size_t) nq * sizeof(uint32_t)));
    CHK(ensureK(ctx, S.nCand, 64));
    CHK(ensurePinned(ctx, S.hMisc, 256));
    CHK(ensurePinned(ctx, S.hChunks, (size_t) nq * sizeof(KmerChunks)));
    CHK(ensurePinned(ctx, S.hEc, (size_t) nq * kMaxChunks * sizeof(uint32_t)));
    CHK(ensurePinned(ctx, S.hRounds, (size_t) nq * kMaxChunks * sizeof(uint32_t)));
    CHK(ensurePinned(ctx, S.hResSize, (size_t) nq * sizeof(uint64_t)));
    CHK(ensurePinned(ctx, S.hThr, (size_t) nq * sizeof(uint32_t)));
    CHK(ensurePinned(ctx, S.hOutCount, (size_t) nq * sizeof(uint32_t)));
    mark("stage");
    hipStream_t st = ctx->stream;
    RPCHK(hipEventRecord(S.ev[0], st));
    RPCHK(hipMemcpyAsync(S.qs.p, S.hQs.p, (size_t) nq * sizeof(KmerQ), hipMemcpyHostToDevice, st));
    RPCHK(hipMemcpyAsync(S.posQuery.p, S.hPosQuery.p, (nPos + 1) * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    RPCHK(hipMemcpyAsync(S.seqs.p, S.hSeqs.p, seqBytes, hipMemcpyHostToDevice, st));
    RPCHK(hipMemcpyAsync(S.thrs.p, S.hThrs.p, (nPos + 1) * sizeof(int16_t), hipMemcpyHostToDevice, st));
    RPCHK(hipMemcpyAsync(S.profiles.p, S.hProfiles.p, profBytes, hipMemcpyHostToDevice, st));
    RPCHK(hipMemsetAsync(S.ec.p, 0, (size_t) nq * kMaxChunks * sizeof(uint32_t), st));
    RPCHK(hipMemsetAsync(S.rounds.p, 0, (size_t) nq * kMaxChunks * sizeof(uint32_t), st));
    RPCHK(hipMemsetAsync(S.resSize.p, 0, (size_t) nq * sizeof(uint64_t), st));
    RPCHK(hipMemsetAsync(S.hist.p, 0, (size_t) nq * 256 * sizeof(uint32_t), st));
    RPCHK(hipMemsetAsync(S.outCount.p, 0, (size_t) nq * sizeof(uint32_t), st));
    RPCHK(hipMemsetAsync(S.nCand.p, 0, 64, st));                 // [0] candidates of the batch, [1] elements handed to the host

    uint64_t *misc = (uint64_t *) S.hMisc.p;
    uint64_t nLists = 0, nHits = 0;
    uint32_t nCand = 0;
    // ---- stage 1: similar k-mers -> lists ---------------------------------------------------------------------
    // one workgroup or one wave per query position (k_kmer.hpp): the wave form when positions have few similar k-mers -- for the count pass
    // judged by the previous batch of this context, for the list pass by this batch's own count.  FSGPU_KMER_WAVE = 0 / 1 forces a form.
    int waveForm = -1;
    if (const char *e = getenv("FSGPU_KMER_WAVE")) waveForm = atoi(e) != 0;
    if (nPos) {
        const bool w = waveForm >= 0 ? waveForm != 0 : (ctx->kmerKPerPos > 0 && ctx->kmerKPerPos < 2048);
        if (w)
            hipLaunchKernelGGL(k_kmer_count_w, dim3((unsigned) ((nPos + 3) / 4)), dim3(256), 0, st, (const KmerQ *) S.qs.p, (const uint16_t *) S.posQuery.p,
                               (const uint8_t *) S.seqs.p, (const int16_t *) S.thrs.p, (uint32_t) nPos, ix.pat, ix.s3, (uint32_t *) S.K.p);
        else
            hipLaunchKernelGGL(k_kmer_count, dim3((unsigned) nPos), dim3(kKmerBlock), 0, st, (const KmerQ *) S.qs.p, (const uint16_t *) S.posQuery.p,
                               (const uint8_t *) S.seqs.p, (const int16_t *) S.thrs.p, (uint32_t) nPos, ix.pat, ix.s3, (uint32_t *) S.K.p);
        RPCHK(hipGetLastError());
    }
    RPCHK(hipMemsetAsync((uint32_t *) S.K.p + nPos, 0, sizeof(uint32_t), st));
    CHK(scanExclusive32to64(ctx, S.tmp, (const uint32_t *) S.K.p, (uint64_t *) S.Kbase.p, nPos + 1));
    RPCHK(hipMemcpyAsync(&misc[0], (uint64_t *) S.Kbase.p + nPos, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    CHK(syncStream(ctx));
    nLists = misc[0];
    mark("count+sync");
    if (nLists >= 0xFFFFFF00ull) {                           // list slots are 32-bit in k_kmer_tile_lists (and 20 bytes of scratch each)
        if (nq > 1) return 1;                                // caller halves the batch
        ctx->err = "k-mer search: a single query produces more than 2^32 similar k-mers"; return FSGPU_E_UNSUPPORTED;
    }
    RPCHK(hipEventRecord(S.ev[1], st));
    CHK(ensureK(ctx, S.listStart, (nLists + 1) * sizeof(uint32_t)));
    CHK(ensureK(ctx, S.listSize, (nLists + 1) * sizeof(uint32_t)));
    CHK(ensureK(ctx, S.listPos, (nLists + 1) * sizeof(uint32_t)));
    CHK(ensureK(ctx, S.listP, (nLists + 1) * sizeof(uint64_t)));
    if (nLists) {
        RPCHK(hipEventRecord(S.ev[10], st));
        ctx->kmerKPerPos = (double) nLists / (double) std::max<uint64_t>(nPos, 1);
        if (waveForm >= 0 ? waveForm != 0 : ctx->kmerKPerPos < 2048)
        {
            // few similar k-mers per position: the small LDS form (more waves per SIMD); FSGPU_KMER_WAVE_SMALL = 0 / 1 forces a form (A/B)
            static const int smallEnv = [] { const char *e = getenv("FSGPU_KMER_WAVE_SMALL"); return e && *e ? atoi(e) : -1; }();
            const bool small = smallEnv >= 0 ? smallEnv != 0 : ctx->kmerKPerPos < 128;
            if (small)
                hipLaunchKernelGGL((k_kmer_lists_w<128, 64>), dim3((unsigned) ((nPos + 3) / 4)), dim3(256), 0, st, (const KmerQ *) S.qs.p, (const uint16_t *) S.posQuery.p,
                                   (const uint8_t *) S.seqs.p, (const int16_t *) S.thrs.p, (uint32_t) nPos, ix.pat, ix.s3, ix.i3, (const uint32_t *) S.K.p,
                                   (const uint64_t *) S.Kbase.p, ix.offsets, ix.bitmap, (uint32_t *) S.listStart.p, (uint32_t *) S.listSize.p, (uint32_t *) S.listPos.p);
            else
                hipLaunchKernelGGL((k_kmer_lists_w<kWaveRowCap, kWaveRuns>), dim3((unsigned) ((nPos + 3) / 4)), dim3(256), 0, st, (const KmerQ *) S.qs.p, (const uint16_t *) S.posQuery.p,
                                   (const uint8_t *) S.seqs.p, (const int16_t *) S.thrs.p, (uint32_t) nPos, ix.pat, ix.s3, ix.i3, (const uint32_t *) S.K.p,
                                   (const uint64_t *) S.Kbase.p, ix.offsets, ix.bitmap, (uint32_t *) S.listStart.p, (uint32_t *) S.listSize.p, (uint32_t *) S.listPos.p);
        }
        else
            hipLaunchKernelGGL(k_kmer_lists, dim3((unsigned) (nPos + nLists / kListSlice + 1)), dim3(kKmerBlock), 0, st, (const KmerQ *) S.qs.p, (const uint16_t *) S.posQuery.p,
                               (const uint8_t *) S.seqs.p, (const int16_t *) S.thrs.p, (uint32_t) nPos, ix.pat, ix.s3, ix.i3, (const uint32_t *) S.K.p,
                               (const uint64_t *) S.Kbase.p, ix.offsets, ix.bitmap, (uint32_t *) S.listStart.p, (uint32_t *) S.listSize.p, (uint32_t *) S.listPos.p);
        RPCHK(hipGetLastError());
        RPCHK(hipEventRecord(S.ev[11], st));
    }
    RPCHK(hipMemsetAsync((uint32_t *) S.listSize.p + nLists, 0, sizeof(uint32_t), st));
    CHK(scanExclusive32to64(ctx, S.tmp, (const uint32_t *) S.listSize.p, (uint64_t *) S.listP.p, nLists + 1));
    hipLaunchKernelGGL(k_kmer_qbases, dim3(gridFor(nq, 64)), dim3(64), 0, st, (KmerQ *) S.qs.p, nq, (const uint64_t *) S.Kbase.p, (const uint64_t *) S.listP.p);
    hipLaunchKernelGGL(k_kmer_chunks, dim3(gridFor(nq, 64)), dim3(64), 0, st, (const KmerQ *) S.qs.p, nq, (const uint64_t *) S.Kbase.p, (const uint64_t *) S.listP.p, (const uint32_t *) S.listPos.p, maxDbMatches, (KmerChunks *) S.chunks.p);
    RPCHK(hipGetLastError());
    RPCHK(hipMemcpyAsync(&misc[1], (uint64_t *) S.listP.p + nLists, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    // the chunk tables: 2 KB per query, of which a query uses nChunks + 1 entries -- a batch of 1024 all-vs-all queries (one or two chunks each) sent 2.1 MB
    // over PCIe; large batches copy the header + the first nine entries of every query and fetch the rest only when a query has more than eight chunks
    constexpr size_t kChunkHead = offsetof(KmerChunks, start) + 9 * sizeof(uint64_t);
    const bool compactChunks = nq > 64;
    if (compactChunks) RPCHK(hipMemcpy2DAsync(S.hChunks.p, sizeof(KmerChunks), S.chunks.p, sizeof(KmerChunks), kChunkHead, (size_t) nq, hipMemcpyDeviceToHost, st));
    else RPCHK(hipMemcpyAsync(S.hChunks.p, S.chunks.p, (size_t) nq * sizeof(KmerChunks), hipMemcpyDeviceToHost, st));
    RPCHK(hipMemcpyAsync(S.hQs.p, S.qs.p, (size_t) nq * sizeof(KmerQ), hipMemcpyDeviceToHost, st));
    CHK(syncStream(ctx));
    if (compactChunks) {
        uint32_t maxC = 1;
        for (int q = 0; q < nq; q++) maxC = std::max<uint32_t>(maxC, ((const KmerChunks *) S.hChunks.p)[q].nChunks);
        if (maxC > 8) {
            const size_t upTo = offsetof(KmerChunks, start) + ((size_t) std::min<uint32_t>(maxC, kMaxChunks) + 1) * sizeof(uint64_t);
            RPCHK(hipMemcpy2DAsync((char *) S.hChunks.p + kChunkHead, sizeof(KmerChunks), (const char *) S.chunks.p + kChunkHead, sizeof(KmerChunks), upTo - kChunkHead, (size_t) nq,
                                   hipMemcpyDeviceToHost, st));
            CHK(syncStream(ctx));
        }
    }
    nHits = misc[1];
    mark("lists+sync");
    RPCHK(hipEventRecord(S.ev[2], st));
    if (nHits >= 0xFFFFFF00ull) {
        if (nq > 1) return 1;                                // caller halves the batch
        ctx->err = "k-mer search: a single query produces more than 2^32 index hits"; return FSGPU_E_UNSUPPORTED;
    }
    // the batch was sized from the PREVIOUS batch's hits per query: one that comes out far beyond the budget (long queries after short
    // ones, a first call with many queries) is split here, where its size is known and none of the 24-bytes-per-hit scratch exists yet
    if (nq > 1 && (double) nHits > 2.0 * kKmerHitBudget) return 1;
    const KmerChunks *hck = (const KmerChunks *) S.hChunks.p;
    // ---- stage 2: hit stream -> (query, key) runs in arrival order -> double-diagonal candidates (k_kmer.hpp) ---------------------
    if (nHits) {
        // tiles: every databaseHits chunk of every query is cut into tiles of kTileA consecutive hits (the chunk starts came back with the lists)
        uint32_t nVq = 0, nT = 0;
        for (int q = 0; q < nq; q++) {
            nVq += hck[q].nChunks;
            for (uint32_t c = 0; c < hck[q].nChunks; c++) nT += (uint32_t) ((hck[q].start[c + 1] - hck[q].start[c] + kTileA - 1) / kTileA);
        }
        // [tileStart nT+1 | qTile0 nq+1 | vqTile0 nVq+1 | qVq0 nq+1] as u32, then [vqQ nVq] as u16
        const size_t tilesWords = (size_t) nT + 1 + nq + 1 + nVq + 1 + nq + 1, tilesBytes = tilesWords * sizeof(uint32_t) + (size_t) nVq * sizeof(uint16_t);
        CHK(ensurePinned(ctx, S.hTiles, tilesBytes));
        CHK(ensureK(ctx, S.tiles, tilesBytes));
        {
            uint32_t *tileStart = (uint32_t *) S.hTiles.p, *qTile0 = tileStart + nT + 1, *vqTile0 = qTile0 + nq + 1, *qVq0 = vqTile0 + nVq + 1;
            uint16_t *vqQ = (uint16_t *) (qVq0 + nq + 1);
            uint32_t T = 0, v = 0;
            for (int q = 0; q < nq; q++) {
                qTile0[q] = T; qVq0[q] = v;
                for (uint32_t c = 0; c < hck[q].nChunks; c++, v++) {
                    vqTile0[v] = T; vqQ[v] = (uint16_t) q;
                    for (uint64_t o = hck[q].start[c]; o < hck[q].start[c + 1]; o += kTileA) tileStart[T++] = (uint32_t) (hq[q].hitBase + o);
                }
            }
            qTile0[nq] = T; qVq0[nq] = v; vqTile0[nVq] = T; tileStart[T] = (uint32_t) nHits;
            if (T != nT || v != nVq) { ctx->err = "k-mer search: inconsistent tile table"; return FSGPU_E_HIP; }
        }
        RPCHK(hipMemcpyAsync(S.tiles.p, S.hTiles.p, tilesBytes, hipMemcpyHostToDevice, st));
        KmerTiles tl{};
        tl.tileStart = (const uint32_t *) S.tiles.p; tl.qTile0 = tl.tileStart + nT + 1; tl.vqTile0 = tl.qTile0 + nq + 1;
        tl.qVq0 = tl.vqTile0 + nVq + 1; tl.vqQ = (const uint16_t *) (tl.qVq0 + nq + 1); tl.nT = nT; tl.nVq = nVq;
        // granularity of the coarse keys: at most 256 keys of 1, 2, 4 ... blocks of 1024 ids (245 keys of 4096 ids at 1M targets: measured best there -- more
        // keys shorten the scatter's runs, fewer widen the duplicate stage's tables and thin out its waves); when the (query, chunk, key) runs would
        // average fewer than 1024 hits -- the many-queries-few-hits batches of an all-vs-all search -- the coarsest level of at most 16 blocks per key
        const KmerIndex::CoarseLevel *lv = nullptr;
        for (const KmerIndex::CoarseLevel &l : ix.levels) if (l.blocksPerKey >= 1 && (lv == nullptr || lv->nKeys > 256)) lv = &l;
        if (lv == nullptr) lv = &ix.levels.front();
        if ((double) nHits / ((double) nVq * lv->nKeys) < 1024.0)
            for (const KmerIndex::CoarseLevel &l : ix.levels) if (l.blocksPerKey >= 1 && l.blocksPerKey <= 16 && l.nKeys <= lv->nKeys) lv = &l;
        if (const char *e = getenv("FSGPU_KMER_BIN_LEVEL")) lv = &ix.levels[std::min<size_t>(ix.levels.size() - 1, (size_t) std::max(0, atoi(e)))];   // tests: force a level
        const uint32_t nKeys = lv->nKeys;
        const bool blkInLds = (size_t) lv->nBlk * sizeof(uint16_t) <= 16 * 1024;
        const KmerCoarse co{lv->blkKey, lv->keyFirst, lv->nBlk, nKeys, (uint32_t) bitsFor(nKeys), lv->maxIds};
        const size_t nCells = (size_t) nT * nKeys, nSegs = (size_t) nq * nKeys;
        const int qtBits = tbits + bitsFor((uint64_t) std::max(nq, 2));
        CHK(ensureK(ctx, S.rec, nHits * sizeof(uint32_t)));
        CHK(ensureK(ctx, S.recKey, nHits * sizeof(uint16_t)));
        CHK(ensureK(ctx, S.recA, nHits * sizeof(uint32_t)));
        CHK(ensureK(ctx, S.ordA, nHits * sizeof(uint16_t)));
        CHK(ensureK(ctx, S.cntA, std::max<size_t>(nCells, 1) * sizeof(uint32_t)));
        CHK(ensureK(ctx, S.colA, std::max<size_t>(nCells, 1) * sizeof(uint32_t)));
        CHK(ensureK(ctx, S.grpSum, nSegs * kColGroups * sizeof(uint32_t)));
        CHK(ensureK(ctx, S.segCnt, (nSegs + 1) * sizeof(uint32_t)));
        CHK(ensureK(ctx, S.segStart, (nSegs + 1) * sizeof(uint32_t)));
        RPCHK(hipMemsetAsync(S.cntA.p, 0, nCells * sizeof(uint32_t), st));
        RPCHK(hipMemsetAsync((uint32_t *) S.segCnt.p + nSegs, 0, sizeof(uint32_t), st));
        {
            const uint32_t nEmit = nT * kSubTiles;
            CHK(ensureK(ctx, S.tileL, (size_t) nEmit * 2 * sizeof(uint32_t)));
            hipLaunchKernelGGL(k_kmer_tile_lists, dim3(gridFor((uint64_t) nEmit * 2, 256)), dim3(256), 0, st, (const uint64_t *) S.listP.p, nLists, tl.tileStart, nT, (uint32_t *) S.tileL.p);
            const size_t ldsEmit = blkInLds ? (size_t) lv->nBlk * sizeof(uint16_t) : 0;
            if (ix.posBits)
                hipLaunchKernelGGL(k_kmer_emit<uint32_t>, dim3(nEmit), dim3(256), ldsEmit, st, nLists, (const uint64_t *) S.listP.p, (const uint32_t *) S.listStart.p,
                                   (const uint32_t *) S.listPos.p, (const uint32_t *) S.tileL.p, (const uint32_t *) ix.entries32, ix.posBits, tl.tileStart, co, blkInLds ? 1 : 0,
                                   (uint32_t *) S.cntA.p, (uint32_t *) S.rec.p, (uint16_t *) S.recKey.p);
            else
                hipLaunchKernelGGL(k_kmer_emit<uint64_t>, dim3(nEmit), dim3(256), ldsEmit, st, nLists, (const uint64_t *) S.listP.p, (const uint32_t *) S.listStart.p,
                                   (const uint32_t *) S.listPos.p, (const uint32_t *) S.tileL.p, (const uint64_t *) ix.entries, 16, tl.tileStart, co, blkInLds ? 1 : 0,
                                   (uint32_t *) S.cntA.p, (uint32_t *) S.rec.p, (uint16_t *) S.recKey.p);
        }
        RPCHK(hipGetLastError());
        RPCHK(hipEventRecord(S.ev[3], st));
        // where every (tile, key) run starts: column sums -> segment starts -> column prefixes (in place of the counts)
        const dim3 colGrid((unsigned) nq, (nKeys + 63) / 64);
        hipLaunchKernelGGL(k_kmer_col_sums, colGrid, dim3(64 * kColGroups), 0, st, (const uint32_t *) S.cntA.p, tl.qTile0, nKeys, (uint32_t *) S.grpSum.p, (uint32_t *) S.segCnt.p);
        CHK(scanExclusive<uint32_t>(ctx, S.tmp, (const uint32_t *) S.segCnt.p, (uint32_t *) S.segStart.p, nSegs + 1));
        hipLaunchKernelGGL(k_kmer_col_offsets, colGrid, dim3(64 * kColGroups), 0, st, (uint32_t *) S.cntA.p, tl.qTile0, nKeys, (const uint32_t *) S.grpSum.p,
                           (const uint32_t *) S.segStart.p, (uint32_t *) S.colA.p);
        {
#define FS_SCATTER(KB) hipLaunchKernelGGL(k_kmer_scatter_stable<KB>, dim3(nT), dim3(kScThreads), 0, st, (const uint32_t *) S.rec.p, (const uint16_t *) S.recKey.p, tl.tileStart, \
                                          (const uint32_t *) S.cntA.p, nKeys, (uint32_t *) S.recA.p, (uint16_t *) S.ordA.p)
            switch (co.keyBits) {          // ballots per record = key bits: unrolled per width
                case 1: FS_SCATTER(1); break; case 2: FS_SCATTER(2); break; case 3: FS_SCATTER(3); break; case 4: FS_SCATTER(4); break; case 5: FS_SCATTER(5); break;
                case 6: FS_SCATTER(6); break; case 7: FS_SCATTER(7); break; case 8: FS_SCATTER(8); break; default: FS_SCATTER(9); break;
            }
#undef FS_SCATTER
        }
        RPCHK(hipGetLastError());
        RPCHK(hipEventRecord(S.ev[4], st));
        // ---- stage 3: the double-diagonal rule, run by run in arrival order ------------------------------------------------
        KmerDupStream da{};
        da.recA = (const uint32_t *) S.recA.p; da.ordA = (const uint16_t *) S.ordA.p; da.note = (uint32_t *) S.rec.p; da.offA = (const uint32_t *) S.cntA.p; da.colA = (const uint32_t *) S.colA.p;
        da.segStart = (const uint32_t *) S.segStart.p; da.tl = tl; da.qs = (const KmerQ *) S.qs.p; da.keyFirst = lv->keyFirst; da.nKeys = nKeys;
        da.tbits = tbits; da.ecCount = (uint32_t *) S.ec.p;
        const uint32_t nRuns = nVq * nKeys;
        CHK(ensureK(ctx, S.candCount, ((size_t) nRuns + 1) * 2 * sizeof(uint32_t)));            // [runCand nRuns + 1 | runBase nRuns + 1]
        da.runCand = (uint32_t *) S.candCount.p; da.runBase = da.runCand + nRuns + 1;
        RPCHK(hipMemsetAsync(S.candCount.p, 0, ((size_t) nRuns + 1) * sizeof(uint32_t), st));
        const size_t ldsDup = ((size_t) lv->maxIds / 4 + 1) * sizeof(uint32_t);           // one byte per target id of the widest key
        if (ldsDup > 60 * 1024 && !ctx->kmerDupAttr) {          // the attribute belongs to the device: once per context
            RPCHK(hipFuncSetAttribute((const void *) k_kmer_dup_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (kCoarseBlocks * 1024 / 4 + 1) * (int) sizeof(uint32_t)));
            ctx->kmerDupAttr = true;
        }
        hipLaunchKernelGGL(k_kmer_dup_stream, dim3(nRuns), dim3(64), ldsDup, st, da);
        RPCHK(hipGetLastError());
        CHK(scanExclusive<uint32_t>(ctx, S.tmp, (const uint32_t *) da.runCand, (uint32_t *) da.runBase, (size_t) nRuns + 1));
        RPCHK(hipMemcpyAsync(&misc[2], da.runBase + nRuns, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        CHK(syncStream(ctx));
        nCand = (uint32_t) misc[2];
        mark("emit..dup+sync");
        ctx->kmerSegs[0] = 0; ctx->kmerSegs[1] = nVq * nKeys; ctx->kmerSegs[2] = 0; ctx->kmerSegs[3] = nT;
        ctx->kmerSegs[4] = nVq * nKeys; ctx->kmerSegs[5] = nKeys; ctx->kmerSegs[6] = lv->maxIds;
        if (nCand) {
            // (query, target, arrival) order: the gathered candidates stand in (query, key) blocks with ascending stream positions, one STABLE radix sort by
            // (query | target) over the flagged hits finishes it
            CHK(ensureK(ctx, S.candKey, (size_t) nCand * sizeof(uint32_t)));
            CHK(ensureK(ctx, S.candVal, (size_t) nCand * sizeof(uint64_t)));
            CHK(ensureK(ctx, S.ckeys, (size_t) nCand * sizeof(uint32_t)));
            CHK(ensureK(ctx, S.cvals, (size_t) nCand * sizeof(uint64_t)));
            da.candKey = (uint32_t *) S.candKey.p; da.candVal = (uint64_t *) S.candVal.p;
            hipLaunchKernelGGL(k_kmer_cand_gather, dim3((nRuns + 3) / 4), dim3(256), 0, st, da, nRuns);
            RPCHK(hipGetLastError());
            CHK((sortPairs<uint32_t, uint64_t>(ctx, S.tmp, (const uint32_t *) S.candKey.p, (uint32_t *) S.ckeys.p, (const uint64_t *) S.candVal.p, (uint64_t *) S.cvals.p,
                                               nCand, std::min(32, qtBits))));
            RPCHK(hipMemcpyAsync(S.nCand.p, da.runBase + nRuns, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        }
    }
    if (!nHits) { RPCHK(hipEventRecord(S.ev[3], st)); RPCHK(hipEventRecord(S.ev[4], st)); }   // keep every stage event recorded
    RPCHK(hipEventRecord(S.ev[5], st));
    if (nCand) {
        CHK(ensureK(ctx, S.kept, (size_t) nCand));
        CHK(ensureK(ctx, S.score, (size_t) nCand * sizeof(int32_t)));
        CHK(ensureK(ctx, S.scrA, (size_t) nCand * sizeof(uint64_t)));
        CHK(ensureK(ctx, S.scrB, (size_t) nCand * sizeof(uint64_t)));
        CHK(ensureK(ctx, S.best, (size_t) nCand * sizeof(KmerBest)));
        CHK(ensureK(ctx, S.out, (size_t) nCand * sizeof(KmerOut)));
        if (sp.kmerScoreOnly) {
            // --diag-score 0: the score of a target is the number of its candidates, no diagonal is scored; a query that refilled databaseHits has
            // the reference's merge of the per-refill counts replayed per (query, id >> shift) group (k_kmer_merge_heads)
            bool refilled = false;
            for (int q = 0; q < nq; q++) refilled = refilled || hck[q].nChunks > 1;
            if (refilled) {
                const unsigned B = pickBins(sp, n);
                int shift = 0;
                while ((1u << shift) < B) shift++;
                hipLaunchKernelGGL(k_kmer_merge_heads, dim3(gridFor(nCand, 128)), dim3(128), 0, st, (const uint32_t *) S.ckeys.p, (const uint64_t *) S.cvals.p, (const uint32_t *) S.nCand.p,
                                   tbits, shift, (const KmerChunks *) S.chunks.p, (const uint32_t *) S.ec.p, (uint64_t *) S.scrA.p, (uint8_t *) S.kept.p, (int32_t *) S.score.p,
                                   (KmerBest *) S.best.p, (uint32_t *) S.rounds.p, (unsigned long long *) S.resSize.p);
            } else {
                hipLaunchKernelGGL(k_kmer_count_heads, dim3(gridFor(nCand, 256)), dim3(256), 0, st, (const uint32_t *) S.ckeys.p, (const uint32_t *) S.nCand.p, tbits,
                                   (uint8_t *) S.kept.p, (int32_t *) S.score.p, (KmerBest *) S.best.p, (unsigned long long *) S.resSize.p);
            }
            RPCHK(hipGetLastError());
            RPCHK(hipEventRecord(S.ev[6], st));
            RPCHK(hipEventRecord(S.ev[7], st));
        } else {
            int maxL = 0;
            for (int q = 0; q < nq; q++) maxL = std::max(maxL, queries[q].L);
            const int ldsBytes = std::min(maxL * 21, 60 * 1024);
            // eight lanes per candidate (k_kmer_score8) unless FSGPU_KMER_SCORE8=0 asks for the one-lane-per-candidate form (A/B runs)
            static const bool score8 = [] { const char *e = getenv("FSGPU_KMER_SCORE8"); return !(e && atoi(e) == 0); }();
            if (score8)
                hipLaunchKernelGGL(k_kmer_score8, dim3(gridFor(nCand, 32)), dim3(256), (size_t) ldsBytes, st, (const uint32_t *) S.ckeys.p, (const uint64_t *) S.cvals.p,
                                   (const uint32_t *) S.nCand.p, tbits, (const KmerQ *) S.qs.p, (const int8_t *) S.profiles.p, ix.masked, db.dOffsets, db.dLengths,
                                   ldsBytes, (uint8_t *) S.kept.p, (int32_t *) S.score.p);
            else
                hipLaunchKernelGGL(k_kmer_score, dim3(gridFor(nCand, 256)), dim3(256), (size_t) ldsBytes, st, (const uint32_t *) S.ckeys.p, (const uint64_t *) S.cvals.p,
                                   (const uint32_t *) S.nCand.p, tbits, (const KmerQ *) S.qs.p, (const int8_t *) S.profiles.p, ix.masked, db.dOffsets, db.dLengths,
                                   ldsBytes, (uint8_t *) S.kept.p, (int32_t *) S.score.p);
            RPCHK(hipGetLastError());
            RPCHK(hipEventRecord(S.ev[6], st));
            // ---- stage 4: per-target replay ----------------------------------------------------------------------
            // FSGPU_KMER_WALK_FF=0: every round of every target is walked (A/B runs; the truncation replay below always uses that form)
            static const bool walkFF = [] { const char *e = getenv("FSGPU_KMER_WALK_FF"); return !(e && atoi(e) == 0); }();
            if (walkFF)
                hipLaunchKernelGGL(k_kmer_walk<true>, dim3(gridFor(nCand, 128)), dim3(128), 0, st, (const uint32_t *) S.ckeys.p, (const uint64_t *) S.cvals.p,
                                   (const uint8_t *) S.kept.p, (const int32_t *) S.score.p, (const uint32_t *) S.nCand.p, tbits, (const KmerChunks *) S.chunks.p,
                                   (uint64_t *) S.scrA.p, (uint64_t *) S.scrB.p, (KmerBest *) S.best.p, (uint32_t *) S.rounds.p, (unsigned long long *) S.resSize.p);
            else
                hipLaunchKernelGGL(k_kmer_walk<false>, dim3(gridFor(nCand, 128)), dim3(128), 0, st, (const uint32_t *) S.ckeys.p, (const uint64_t *) S.cvals.p,
                                   (const uint8_t *) S.kept.p, (const int32_t *) S.score.p, (const uint32_t *) S.nCand.p, tbits, (const KmerChunks *) S.chunks.p,
                                   (uint64_t *) S.scrA.p, (uint64_t *) S.scrB.p, (KmerBest *) S.best.p, (uint32_t *) S.rounds.p, (unsigned long long *) S.resSize.p);
            RPCHK(hipGetLastError());
            RPCHK(hipEventRecord(S.ev[7], st));
        }
        // ---- stage 5: histogram, cut, hand-over --------------------------------------------------------------
        hipLaunchKernelGGL(k_kmer_hist, dim3(gridFor(nCand, 256)), dim3(256), 0, st, (const uint32_t *) S.ckeys.p, (const KmerBest *) S.best.p,
                           (const uint32_t *) S.nCand.p, tbits, (uint32_t *) S.hist.p);
        hipLaunchKernelGGL(k_kmer_cut, dim3(gridFor(nq, 64)), dim3(64), 0, st, (const uint32_t *) S.hist.p, nq, maxHits, (uint32_t) sp.minDiagScoreThr, (uint32_t *) S.thr.p);
        hipLaunchKernelGGL(k_kmer_out, dim3(gridFor(nCand, 1024)), dim3(1024), 0, st, (const uint32_t *) S.ckeys.p, (const uint64_t *) S.cvals.p, (const int32_t *) S.score.p,
                           (const KmerBest *) S.best.p, (const uint32_t *) S.nCand.p, tbits, (const uint32_t *) S.thr.p, nCand, (uint32_t *) S.outCount.p,
                           (uint32_t *) S.nCand.p + 1, (KmerOut *) S.out.p, (const uint64_t *) S.scrA.p, (const uint64_t *) S.scrB.p);
        RPCHK(hipGetLastError());
    } else {
        for (int e = 6; e <= 7; e++) RPCHK(hipEventRecord(S.ev[e], st));
        RPCHK(hipMemsetAsync(S.thr.p, 0, (size_t) nq * sizeof(uint32_t), st));
    }
    {
        // per (query, chunk) counters: the host reads entries [0, nChunks) of a query only -- the columns the batch uses come back, not the 2 x 1 KB per query
        // (a batch of 1024 all-vs-all queries with one or two chunks each sent 2 MB over PCIe for 8 KB)
        uint32_t maxC = 1;
        for (int q = 0; q < nq; q++) maxC = std::max<uint32_t>(maxC, hck[q].nChunks);
        const size_t pitch = (size_t) kMaxChunks * sizeof(uint32_t), width = (size_t) std::min<uint32_t>(maxC, kMaxChunks) * sizeof(uint32_t);
        RPCHK(hipMemcpy2DAsync(S.hEc.p, pitch, S.ec.p, pitch, width, (size_t) nq, hipMemcpyDeviceToHost, st));
        RPCHK(hipMemcpy2DAsync(S.hRounds.p, pitch, S.rounds.p, pitch, width, (size_t) nq, hipMemcpyDeviceToHost, st));
    }
    RPCHK(hipMemcpyAsync(S.hResSize.p, S.resSize.p, (size_t) nq * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    RPCHK(hipMemcpyAsync(S.hThr.p, S.thr.p, (size_t) nq * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    RPCHK(hipMemcpyAsync(S.hOutCount.p, S.outCount.p, (size_t) nq * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    RPCHK(hipEventRecord(S.ev[8], st));
    CHK(syncStream(ctx));
    mark("score..out+sync");
    // ---- findDuplicates cut short by its output capacity: replayed per (query, chunk) in the diagonal-score mode -----------------------------
    // (--diag-score 0 keeps the status: there the reference's next merge reads the bytes the aborted bin left behind)
    std::vector<char> truncReplayed;
    if (nCand && !sp.kmerScoreOnly) {
        std::vector<int> flagged;
        for (int q = 0; q < nq; q++)
            if (outputTestFires(sp, n, hck[q], (const uint32_t *) S.hEc.p + (size_t) q * kMaxChunks, (const uint32_t *) S.hRounds.p + (size_t) q * kMaxChunks)) flagged.push_back(q);
        const uint32_t B = pickBins(sp, n);
        const size_t histWords = flagged.size() * (size_t) kMaxChunks * B * 2;
        if (!flagged.empty() && histWords <= (64u << 20)) {              // <= 256 MB of counters; beyond that the queries keep their status
            const uint64_t foundSize = sp.foundDiagonalsSize ? (uint64_t) sp.foundDiagonalsSize : std::max<uint64_t>(n, 1000000);
            std::vector<int32_t> slot(nq, -1);
            for (size_t k = 0; k < flagged.size(); k++) slot[flagged[k]] = (int32_t) k;
            CHK(ensureK(ctx, S.kept0, (size_t) nCand));
            CHK(ensureK(ctx, S.qSlot, (size_t) nq * sizeof(int32_t)));
            CHK(ensureK(ctx, S.truncHist, histWords * sizeof(uint32_t)));
            CHK(ensureK(ctx, S.trunc, (size_t) nq * kMaxChunks * sizeof(uint32_t)));
            RPCHK(hipMemcpyAsync(S.kept0.p, S.kept.p, (size_t) nCand, hipMemcpyDeviceToDevice, st));
            RPCHK(hipMemcpyAsync(S.qSlot.p, slot.data(), (size_t) nq * sizeof(int32_t), hipMemcpyHostToDevice, st));
            RPCHK(hipMemsetAsync(S.truncHist.p, 0, histWords * sizeof(uint32_t), st));
            hipLaunchKernelGGL(k_kmer_trunc_hist, dim3(gridFor(nCand, 256)), dim3(256), 0, st, (const uint32_t *) S.ckeys.p, (const uint64_t *) S.cvals.p, (const uint8_t *) S.kept.p,
                               (const uint32_t *) S.nCand.p, tbits, B, (const int32_t *) S.qSlot.p, (uint32_t *) S.truncHist.p);
            RPCHK(hipGetLastError());
            std::vector<uint32_t> hh(histWords);
            RPCHK(hipMemcpyAsync(hh.data(), S.truncHist.p, histWords * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            CHK(syncStream(ctx));
            std::vector<uint32_t> trunc((size_t) nq * kMaxChunks, 0xFFFFFFFFu);
            std::vector<uint32_t> next(flagged.size(), 0);
            bool any = false;
            for (;;) {
                bool changed = false;
                for (size_t k = 0; k < flagged.size(); k++) {
                    const int q = flagged[k];
                    const KmerChunks &ck = hck[q];
                    const uint32_t C = ck.nChunks - 1;
                    const uint32_t *rounds = (const uint32_t *) S.hRounds.p + (size_t) q * kMaxChunks;
                    while (next[k] <= C) {
                        const uint32_t c = next[k]++;
                        const uint64_t before = c == 0 ? 0 : rounds[c];
                        const uint64_t outSize = foundSize > before ? foundSize - before : 0;
                        const uint32_t *h = hh.data() + ((k * kMaxChunks + c) * (size_t) B) * 2;
                        uint64_t dbl = 0;
                        uint32_t cut = 0xFFFFFFFFu;
                        for (uint32_t b = 0; b < B; b++) {
                            if (dbl + h[2 * b] >= outSize) { cut = b; break; }
                            dbl += h[2 * b + 1];
                        }
                        if (cut != 0xFFFFFFFFu) { trunc[(size_t) q * kMaxChunks + c] = cut; changed = true; break; }     // the later rounds' element counts are stale now
                    }
                }
                if (!changed) break;
                any = true;
                RPCHK(hipMemcpyAsync(S.trunc.p, trunc.data(), trunc.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                RPCHK(hipMemsetAsync(S.rounds.p, 0, (size_t) nq * kMaxChunks * sizeof(uint32_t), st));
                RPCHK(hipMemsetAsync(S.resSize.p, 0, (size_t) nq * sizeof(uint64_t), st));
                hipLaunchKernelGGL(k_kmer_apply_trunc, dim3(gridFor(nCand, 256)), dim3(256), 0, st, (const uint32_t *) S.ckeys.p, (const uint64_t *) S.cvals.p, (const uint8_t *) S.kept0.p,
                                   (const uint32_t *) S.nCand.p, tbits, B, (const uint32_t *) S.trunc.p, (uint8_t *) S.kept.p);
                hipLaunchKernelGGL(k_kmer_walk<false>, dim3(gridFor(nCand, 128)), dim3(128), 0, st, (const uint32_t *) S.ckeys.p, (const uint64_t *) S.cvals.p,
                                   (const uint8_t *) S.kept.p, (const int32_t *) S.score.p, (const uint32_t *) S.nCand.p, tbits, (const KmerChunks *) S.chunks.p,
                                   (uint64_t *) S.scrA.p, (uint64_t *) S.scrB.p, (KmerBest *) S.best.p, (uint32_t *) S.rounds.p, (unsigned long long *) S.resSize.p);
                RPCHK(hipGetLastError());
                RPCHK(hipMemcpyAsync(S.hRounds.p, S.rounds.p, (size_t) nq * kMaxChunks * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                RPCHK(hipMemcpyAsync(S.hResSize.p, S.resSize.p, (size_t) nq * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
                CHK(syncStream(ctx));
            }
            if (any) {      // histogram, cut and hand-over once more, on the truncated lists
                RPCHK(hipMemsetAsync(S.hist.p, 0, (size_t) nq * 256 * sizeof(uint32_t), st));
                RPCHK(hipMemsetAsync(S.outCount.p, 0, (size_t) nq * sizeof(uint32_t), st));
                RPCHK(hipMemsetAsync((uint32_t *) S.nCand.p + 1, 0, sizeof(uint32_t), st));
                hipLaunchKernelGGL(k_kmer_hist, dim3(gridFor(nCand, 256)), dim3(256), 0, st, (const uint32_t *) S.ckeys.p, (const KmerBest *) S.best.p,
                                   (const uint32_t *) S.nCand.p, tbits, (uint32_t *) S.hist.p);
                hipLaunchKernelGGL(k_kmer_cut, dim3(gridFor(nq, 64)), dim3(64), 0, st, (const uint32_t *) S.hist.p, nq, maxHits, (uint32_t) sp.minDiagScoreThr, (uint32_t *) S.thr.p);
                hipLaunchKernelGGL(k_kmer_out, dim3(gridFor(nCand, 1024)), dim3(1024), 0, st, (const uint32_t *) S.ckeys.p, (const uint64_t *) S.cvals.p, (const int32_t *) S.score.p,
                                   (const KmerBest *) S.best.p, (const uint32_t *) S.nCand.p, tbits, (const uint32_t *) S.thr.p, nCand, (uint32_t *) S.outCount.p,
                                   (uint32_t *) S.nCand.p + 1, (KmerOut *) S.out.p, (const uint64_t *) S.scrA.p, (const uint64_t *) S.scrB.p);
                RPCHK(hipGetLastError());
                RPCHK(hipMemcpyAsync(S.hThr.p, S.thr.p, (size_t) nq * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                RPCHK(hipMemcpyAsync(S.hOutCount.p, S.outCount.p, (size_t) nq * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                CHK(syncStream(ctx));
            }
            truncReplayed.assign(nq, 0);
            for (int q : flagged) truncReplayed[q] = 1;
            mark("truncation replay");
        }
    }
    const uint32_t *hOutCount = (const uint32_t *) S.hOutCount.p;
    size_t totalOut = 0;
    std::vector<size_t> outOff(nq + 1, 0);
    for (int q = 0; q < nq; q++) { outOff[q] = totalOut; totalOut += hOutCount[q]; }
    outOff[nq] = totalOut;
    if (totalOut > nCand) { ctx->err = "k-mer search: output count exceeds the candidate count"; return FSGPU_E_HIP; }
    CHK(ensurePinned(ctx, S.hOut, std::max<size_t>(1, totalOut) * 2 * sizeof(KmerOut)));
    if (totalOut) RPCHK(hipMemcpyAsync((KmerOut *) S.hOut.p + totalOut, S.out.p, totalOut * sizeof(KmerOut), hipMemcpyDeviceToHost, st));
    RPCHK(hipEventRecord(S.ev[9], st));
    CHK(syncStream(ctx));
    {   // the device hands the elements over in no particular order: counting sort by query (query index in count >> 8)
        const HostOut *src = (const HostOut *) S.hOut.p + totalOut;
        HostOut *dst = (HostOut *) S.hOut.p;
        std::vector<size_t> at(outOff.begin(), outOff.end() - 1);
        for (size_t i = 0; i < totalOut; i++) {
            const uint32_t q = src[i].count >> 8;
            if (q >= (uint32_t) nq || at[q] >= outOff[q + 1]) { ctx->err = "k-mer search: inconsistent output element"; return FSGPU_E_HIP; }
            dst[at[q]] = src[i];
            dst[at[q]++].count &= 0xffu;
        }
    }
    {
        float ms = 0;
        static const int a[9] = {0, 0, 1, 2, 3, 4, 5, 6, 7}, b[9] = {9, 1, 2, 3, 4, 5, 6, 7, 8};
        // [0] total, [1] count+scan, [2] lists+chunks, [3] emit, [4] sort, [5] dup flags+scan, [6] compact+score, [7] walk, [8] hist/cut/out
        // summed over the device batches of one fsgpu_kmer_search call (the caller resets them)
        for (int i = 0; i < 9; i++) if (hipEventElapsedTime(&ms, S.ev[a[i]], S.ev[b[i]]) == hipSuccess) ctx->kmerMs[i] = std::max(0.0, ctx->kmerMs[i]) + (double) ms;
        if (nLists && hipEventElapsedTime(&ms, S.ev[10], S.ev[11]) == hipSuccess) ctx->kmerMs[10] = std::max(0.0, ctx->kmerMs[10]) + (double) ms;
        (void) hipGetLastError();   // an elapsed-time query must never leave a sticky error behind
        ctx->kmerCounts[0] += nLists; ctx->kmerCounts[1] += nHits; ctx->kmerCounts[2] += nCand; ctx->kmerCounts[3] += totalOut;
    }
    mark("copy out+sync");
    // ---- host tail ---------------------------------------------------------------------------------------------
    const auto tTail = std::chrono::steady_clock::now();
    const KmerQ *hq2 = (const KmerQ *) S.hQs.p;
    // one query per call of the host pool (search.cpp::HostPool; the caller takes part): the tails of a batch are independent -- 40 us each, 1.3 ms per batch
    // of 32 when a single feeder thread walks them
    fshostParallelFor(nq, [&](int q) {
        std::vector<HostOut> el;
        const size_t c = outOff[q + 1] - outOff[q];
        el.resize(c);
        if (c) memcpy(el.data(), (const HostOut *) S.hOut.p + outOff[q], c * sizeof(HostOut));
        status[q] = finishQuery(sp, n, queries[q], hck[q], (const uint32_t *) S.hEc.p + (size_t) q * kMaxChunks, (const uint32_t *) S.hRounds.p + (size_t) q * kMaxChunks,
                                ((const uint64_t *) S.hResSize.p)[q], ((const uint32_t *) S.hThr.p)[q], el, out + (size_t) q * sp.maxResListLen, &nout[q],
                                !truncReplayed.empty() && truncReplayed[q]);
        if (stats) {
            uint64_t le = nLists;
            for (int r = q + 1; r < nq; r++) { le = hq2[r].listBase; break; }
            stats[q * 4 + 0] = queries[q].L > 0 ? (double) (le - hq2[q].listBase) / (double) queries[q].L : 0.0;
            stats[q * 4 + 1] = (double) hck[q].total;
            if (hck[q].aborted == 1) {      // match() left its loop at that list (QueryMatcher.cpp:330-332): the statistics stop there too
                stats[q * 4 + 0] = (double) hck[q].abortKmers / (double) queries[q].L;
                stats[q * 4 + 1] = (double) hck[q].start[hck[q].nChunks - 1];
            }
            stats[q * 4 + 2] = hck[q].nChunks > 1 ? 1.0 : 0.0;
            stats[q * 4 + 3] = (double) pickBins(sp, n);
        }
    });
    ctx->kmerMs[9] = std::max(0.0, ctx->kmerMs[9]) + std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tTail).count();
    mark("host tail");
    if (trace) fprintf(stderr, "kmer batch nq=%d hits=%llu cand=%u:%s\n", nq, (unsigned long long) nHits, nCand, traceLine.c_str());
#undef CHK
    return FSGPU_OK;
}

extern "C" int fsgpu_kmer_search(fsgpu_ctx *ctx, const fsgpu_kmer_search_params *p, const fsgpu_kmer_query *queries, int nq,
                                 fsgpu_kmer_hit *out, int32_t *nout, int32_t *status, double *stats) {
    if (!ctx || !p || (nq > 0 && (!queries || !out || !nout || !status))) return FSGPU_E_ARG;
    if (!ctx->kidx) { ctx->err = "k-mer index not built"; return FSGPU_E_NODB; }
    if (p->maxResListLen <= 0 || p->minDiagScoreThr < 0 || p->minDiagScoreThr > 255 * (p->kmerScoreOnly ? 1 : 1000)) {
        ctx->err = "k-mer search: maxResListLen >= 1 and minDiagScoreThr >= 0 required"; return FSGPU_E_UNSUPPORTED;
    }
    if (p->bins && (p->bins & (p->bins - 1))) { ctx->err = "k-mer search: bins must be a power of two"; return FSGPU_E_ARG; }
    RPCHK(hipSetDevice(ctx->device));
    // queries per device batch: the candidate keys are (query << tbits | target) in 32 bits; beyond that the batch is sized by its hit
    // stream (24 bytes of scratch per index hit, fewer than 2^32 hits): the hits per query of the previous batch of this context set the
    // size of the next one, a batch that still comes out too large is halved and redone
    const int tb = ctx->kidx->tbits;
    const int maxBatch = std::max(1, std::min(1 << std::min(12, 32 - tb), 1024));
    const double hitBudget = kKmerHitBudget;
    for (int i = 0; i < 12; i++) ctx->kmerMs[i] = -1;          // < 0: nothing recorded (fsgpu_last_kernel_ms)
    for (int i = 0; i < 4; i++) ctx->kmerCounts[i] = 0;
    int q0 = 0;
    while (q0 < nq) {
        int batch = std::min(maxBatch, 32);                   // no history on this context: a batch of 32 shows what a query costs here
        if (ctx->kmerHitsPerQuery > 0) batch = (int) std::max(1.0, std::min((double) maxBatch, hitBudget / ctx->kmerHitsPerQuery));   // queries of more than budget / 8 hits each: fewer than 8 per batch
        batch = std::min(batch, maxBatch);
        if (ctx->kmerBatchCap > 0) batch = std::min(batch, ctx->kmerBatchCap);
        // the rest of the call in device batches of equal size
        const int left = nq - q0, parts = (left + batch - 1) / batch;
        const int m = (left + parts - 1) / parts;
        const uint64_t hitsBefore = ctx->kmerCounts[1];
        int rc = kmerBatch(ctx, *p, queries + q0, m, out + (size_t) q0 * p->maxResListLen, nout + q0, status + q0, stats ? stats + (size_t) q0 * 4 : nullptr);
        if (rc == 1) { ctx->kmerBatchCap = std::max(1, m / 2); ctx->kmerBatchOk = 0; continue; }            // too many hits / out of memory: redo with half the queries
        if (rc != FSGPU_OK) return rc;
        // one heavy batch does not cap the context for good, but the cap is only relaxed after four batches in a row went through under it: doubling it
        // after every success made every second batch of a run of heavy queries fail, be abandoned after its first stage and be redone
        if (ctx->kmerBatchCap > 0 && ++ctx->kmerBatchOk >= 4) { ctx->kmerBatchCap = 2 * ctx->kmerBatchCap >= maxBatch ? 0 : 2 * ctx->kmerBatchCap; ctx->kmerBatchOk = 0; }
        ctx->kmerHitsPerQuery = (double) (ctx->kmerCounts[1] - hitsBefore) / (double) std::max(1, m);
        q0 += m;
    }
    return FSGPU_OK;
}
