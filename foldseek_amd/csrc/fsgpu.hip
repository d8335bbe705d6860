// fsgpu.hip -- C ABI (include/fsgpu.h) over the gfx950 kernels.  Host side: HIP runtime only, no torch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "fsgpu_ctx.h"
#include "k_select.hpp"
#include "k_gapless.hpp"
#include "k_sw.hpp"
#include "k_sw3.hpp"
#include "fsgpu_sw3.h"

static thread_local std::string g_createError;

extern "C" {

// devices with at least one live context in this process (fsgpu_live_devices: the host side divides the process's cores by it)
static std::mutex g_liveM;
static int g_liveCtx[64] = {0};
static void liveAdd(int device, int d) { std::lock_guard<std::mutex> g(g_liveM); if (device >= 0 && device < 64) g_liveCtx[device] += d; }
int fsgpu_live_devices(void) {
    std::lock_guard<std::mutex> g(g_liveM);
    int n = 0;
    for (int i = 0; i < 64; i++) if (g_liveCtx[i] > 0) n++;
    return n;
}

int fsgpu_create(int device, fsgpu_ctx **out) {
    if (!out) return FSGPU_E_ARG;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_createError = std::string("no HIP device available: ") + hipGetErrorString(e);
        return FSGPU_E_HIP;
    }
    if (device < 0 || device >= count) { g_createError = "device index out of range"; return FSGPU_E_ARG; }
    fsgpu_ctx *ctx = new fsgpu_ctx();
    ctx->device = device;
    auto fail = [&](const char *what, hipError_t err) {
        g_createError = std::string(what) + ": " + hipGetErrorString(err);
        delete ctx;
        return FSGPU_E_HIP;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail("hipGetDeviceProperties", e);
    ctx->numCU = prop.multiProcessorCount;
    if (const char *e2 = getenv("FSGPU_GAPLESS_BLOCKS_PER_CU")) ctx->gaplessBlocksPerCU = std::max(1, atoi(e2));
    // FSGPU_SW_CUS=<n> (A/B measurement): a CU-mask split instead of stream priorities -- the batch SW's streams own the last n CUs of the
    // mask (n / 8 per XCD where the mask interleaves them), the context's stream (scans, selection, k-mer batches) the others
    const int swCUs = [] { const char *e = getenv("FSGPU_SW_CUS"); return e ? atoi(e) : 0; }();
    if (swCUs > 0 && swCUs < ctx->numCU) {
        const int words = (ctx->numCU + 31) / 32;
        std::vector<uint32_t> mScan(words, 0), mSw(words, 0);
        for (int c = 0; c < ctx->numCU; c++) (c < ctx->numCU - swCUs ? mScan : mSw)[c / 32] |= 1u << (c % 32);
        if ((e = hipExtStreamCreateWithCUMask(&ctx->stream, (uint32_t) words, mScan.data())) != hipSuccess) return fail("hipExtStreamCreateWithCUMask", e);
        if ((e = hipExtStreamCreateWithCUMask(&ctx->swHi, (uint32_t) words, mSw.data())) != hipSuccess) return fail("hipExtStreamCreateWithCUMask", e);
        ctx->swCuMask = mSw;
    } else {
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    {
        // the batch SW of a context runs on a stream of the highest priority: its launches are short and a host thread waits for them, while the
        // scans and k-mer batches of the other contexts would otherwise keep them queueing (FSGPU_SW_PRIORITY=0: use the context's stream)
        const char *pe = getenv("FSGPU_SW_PRIORITY");
        int lo = 0, hi = 0;
        if (!(pe && atoi(pe) == 0) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo) {
            if (hipStreamCreateWithPriority(&ctx->swHi, hipStreamNonBlocking, hi) != hipSuccess) { ctx->swHi = nullptr; (void) hipGetLastError(); }
            ctx->swHiPrio = hi;
        }
    }
    }
    for (int i = 0; i < 4; i++)
        if ((e = hipEventCreate(&ctx->ev[i])) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipMalloc((void **) &ctx->dMeta, sizeof(SelMeta))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc((void **) &ctx->queue, 256)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipHostMalloc((void **) &ctx->hMeta, sizeof(SelMeta))) != hipSuccess) return fail("hipHostMalloc", e);
    liveAdd(device, +1);
    *out = ctx;
    return FSGPU_OK;
}

static void freeDb(fsgpu_ctx *ctx) { ctx->db.reset(); ctx->kidx.reset(); }

int fsgpu_clone(const fsgpu_ctx *src, fsgpu_ctx **out) {
    if (!src || !out) return FSGPU_E_ARG;
    int rc = fsgpu_create(src->device, out);
    if (rc != FSGPU_OK) return rc;
    (*out)->db = src->db;
    (*out)->kidx = src->kidx;
    return FSGPU_OK;
}

void fsgpu_destroy(fsgpu_ctx *ctx) {
    if (!ctx) return;
    liveAdd(ctx->device, -1);
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->swLong) (void) hipStreamSynchronize(ctx->swLong);
    if (ctx->swHi) (void) hipStreamSynchronize(ctx->swHi);          // the k_sw3 path runs here and on swAux: nothing may be in flight when its buffers go
    for (int i = 0; i < fsgpu_ctx::kSwAux; i++) if (ctx->swAux[i]) (void) hipStreamSynchronize(ctx->swAux[i]);
    if (ctx->swChainEv) {
        if (ctx->db) { std::lock_guard<std::mutex> g(ctx->db->scanMutex); if (ctx->db->lastScanDone == ctx->swChainEv) ctx->db->lastScanDone = nullptr; }
        (void) hipEventDestroy(ctx->swChainEv);
        ctx->swChainEv = nullptr;
    }
    if (ctx->scanDoneEv) {
        if (ctx->db) { std::lock_guard<std::mutex> g(ctx->db->scanMutex); if (ctx->db->lastScanDone == ctx->scanDoneEv) ctx->db->lastScanDone = nullptr; }
        (void) hipEventDestroy(ctx->scanDoneEv);
    }
    freeDb(ctx);
    ctx->kidx.reset();
    if (ctx->kmer) fsgpu_kmer_free_scratch(ctx->kmer);
    DevBuf *bufs[] = {&ctx->gBorder0, &ctx->gBorder1, &ctx->scoreAcc, &ctx->pssm, &ctx->scores, &ctx->chunkHist, &ctx->baseGt, &ctx->baseTie, &ctx->outId, &ctx->outScore,
                      &ctx->img, &ctx->tids, &ctx->res0, &ctx->res1, &ctx->border0, &ctx->border1, &ctx->keys, &ctx->lbuf, &ctx->lres,
                      &ctx->ovAA, &ctx->ovSS, &ctx->ovOff, &ctx->ovLen, &ctx->s3img, &ctx->s3pass, &ctx->s3build, &ctx->s3res,
                      &ctx->btSeq, &ctx->btTrace, &ctx->btBlocks, &ctx->btOut, &ctx->btIn,
                      &ctx->mqPssm, &ctx->mqScores, &ctx->mqQueues, &ctx->mqRec, &ctx->mqHist, &ctx->mqBaseGt, &ctx->mqBaseTie, &ctx->mqMeta,
                      &ctx->mqOutId, &ctx->mqOutScore, &ctx->mqIdent};
    for (DevBuf *b : bufs) if (b->p) hipFree(b->p);
    hipFree(ctx->dMeta); hipFree(ctx->queue);
    hipHostFree(ctx->hMeta); hipHostFree(ctx->hOutId.p); hipHostFree(ctx->hOutScore.p);
    hipHostFree(ctx->hRes0.p); hipHostFree(ctx->hRes1.p); hipHostFree(ctx->hLbuf.p); hipHostFree(ctx->hLres.p);
    hipHostFree(ctx->hS3pass.p); hipHostFree(ctx->hS3build.p); hipHostFree(ctx->hS3res.p);
    hipHostFree(ctx->hBtIn.p); hipHostFree(ctx->hBtOut.p);
    if (ctx->swLong) (void) hipStreamDestroy(ctx->swLong);
    if (ctx->swHi) (void) hipStreamDestroy(ctx->swHi);
    hipHostFree(ctx->hPssm.p); hipHostFree(ctx->hImg.p); hipHostFree(ctx->hTids.p);
    hipHostFree(ctx->hMqPssm.p); hipHostFree(ctx->hMqRec.p); hipHostFree(ctx->hMqMeta.p); hipHostFree(ctx->hMqOutId.p); hipHostFree(ctx->hMqOutScore.p); hipHostFree(ctx->hMqIdent.p);
    for (int i = 0; i < 4; i++) if (ctx->ev[i]) hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < 4; i++) if (ctx->swDirEv[i]) hipEventDestroy(ctx->swDirEv[i]);
    for (int i = 0; i < fsgpu_ctx::kSwAux; i++) if (ctx->swAux[i]) (void) hipStreamDestroy(ctx->swAux[i]);
    for (int i = 0; i <= fsgpu_ctx::kSwAux; i++) if (ctx->swAuxEv[i]) (void) hipEventDestroy(ctx->swAuxEv[i]);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *fsgpu_last_error(const fsgpu_ctx *ctx) { return ctx ? ctx->err.c_str() : g_createError.c_str(); }
int fsgpu_device(const fsgpu_ctx *ctx) { return ctx ? ctx->device : -1; }
int fsgpu_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
void *fsgpu_stream(const fsgpu_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }
uint64_t fsgpu_db_size(const fsgpu_ctx *ctx) { return ctx && ctx->db ? ctx->db->n : 0; }
uint64_t fsgpu_db_residues(const fsgpu_ctx *ctx) { return ctx && ctx->db ? ctx->db->residues : 0; }

} // extern "C"

// ------------------------------------------------------------------------------------------------------------
// database re-tiling kernels
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_db_scan_layout(const uint8_t *raw, const uint64_t *offsets, const int32_t *lengths,
                                                        uint32_t n, const uint64_t *stripeOff, const uint32_t *stripeLen,
                                                        const uint32_t *stripeTargets, uint4 *out) {
    const uint32_t stripe = blockIdx.x;
    const uint32_t len16 = stripeLen[stripe];
    const int j = threadIdx.x & 7;
    const uint32_t t = stripeTargets[stripe * kStripeTargets + j];
    const bool live = t < n;
    const uint64_t off = live ? offsets[t] : 0;
    const int L = live ? lengths[t] : 0;
    uint4 *dst = out + stripeOff[stripe];
    for (uint32_t c = threadIdx.x >> 3; c < len16; c += 32) {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 16; b++) {
            int col = (int) c * 16 + b;
            uint32_t code = kDeadCode;
            if (col < L) {
                code = raw[off + col];
                code = code > 20 ? 20 : code;     // soft-masked (>= 32) and anything unknown -> X
            }
            w[b >> 2] |= code << ((b & 3) * 8);
        }
        dst[(size_t) c * 8 + j] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

__global__ void k_db_unmask(const uint8_t *raw, uint8_t *out, uint64_t bytes) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (; i < bytes; i += stride) {
        uint8_t c = raw[i];
        c = c >= 32 ? c - 32 : c;
        out[i] = c > 20 ? 20 : c;
    }
}

static int buildDb(fsgpu_ctx *ctx, const uint8_t *dRaw3di, const uint8_t *dRawAA, const uint64_t *dOff, const int32_t *dLen,
                   uint64_t n, uint64_t bytes) {
    ctx->db = std::make_shared<DbStore>();
    // host copy of the lengths drives the stripe table
    ctx->db->hLengths.resize(n);
    HIPCHK(hipMemcpy(ctx->db->hLengths.data(), dLen, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    const uint32_t nStripes = (uint32_t) ((n + kStripeTargets - 1) / kStripeTargets);
    std::vector<uint64_t> sOff(nStripes);
    std::vector<uint32_t> sLen(nStripes);
    uint64_t total = 0, residues = 0;
    int maxLen = 0;
    // a stripe = 8 targets of similar length: group along the length-sorted order (identity for a padded DB, which
    // makepaddedseqdb has already sorted; an ASCII DB arrives in arbitrary order)
    std::vector<uint32_t> sTargets((size_t) nStripes * kStripeTargets, 0xffffffffu);
    {
        std::vector<uint32_t> byLen(n);
        std::iota(byLen.begin(), byLen.end(), 0u);
        const std::vector<int32_t> &hl = ctx->db->hLengths;
        if (!std::is_sorted(hl.begin(), hl.end())) std::stable_sort(byLen.begin(), byLen.end(), [&](uint32_t a, uint32_t b) { return hl[a] < hl[b]; });
        std::copy(byLen.begin(), byLen.end(), sTargets.begin());
    }
    for (uint32_t s = 0; s < nStripes; s++) {
        int mx = 0;
        for (uint64_t k = (uint64_t) s * 8; k < std::min<uint64_t>(n, (uint64_t) s * 8 + 8); k++) {
            int L = ctx->db->hLengths[sTargets[k]];
            if (L < 0 || L > FSGPU_MAX_SEQ_LEN) { ctx->err = "target length out of range"; return FSGPU_E_ARG; }
            mx = std::max(mx, L);
            residues += (uint64_t) L;
        }
        maxLen = std::max(maxLen, mx);
        sLen[s] = (uint32_t) ((mx + 15) / 16);
        sOff[s] = total;
        total += (uint64_t) sLen[s] * 8;
    }
    ctx->db->hStripeLen = sLen;

    HIPCHK(hipMalloc((void **) &ctx->db->scan, std::max<uint64_t>(total, 1) * sizeof(uint4)));
    HIPCHK(hipMalloc((void **) &ctx->db->stripeOff, std::max<size_t>(nStripes, 1) * sizeof(uint64_t)));
    HIPCHK(hipMalloc((void **) &ctx->db->stripeLen, std::max<size_t>(nStripes, 1) * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void **) &ctx->db->stripeTargets, std::max<size_t>(sTargets.size(), 1) * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void **) &ctx->db->aln3di, std::max<uint64_t>(bytes, 1)));
    HIPCHK(hipMalloc((void **) &ctx->db->dOffsets, (n + 1) * sizeof(uint64_t)));
    HIPCHK(hipMalloc((void **) &ctx->db->dLengths, std::max<uint64_t>(n, 1) * sizeof(int32_t)));
    if (nStripes) {
        HIPCHK(hipMemcpy(ctx->db->stripeOff, sOff.data(), nStripes * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->db->stripeLen, sLen.data(), nStripes * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->db->stripeTargets, sTargets.data(), sTargets.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    // stream-ordered copies: a device-to-device hipMemcpy runs on the null stream and is NOT synchronous with the host, and the context's
    // stream is non-blocking, so the layout kernels below could otherwise start before their offsets / lengths have arrived (seen as an
    // intermittent memory fault when two processes time-share one device)
    HIPCHK(hipMemcpyAsync(ctx->db->dOffsets, dOff, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->db->dLengths, dLen, n * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
    if (nStripes) {
        hipLaunchKernelGGL(k_db_scan_layout, dim3(nStripes), dim3(256), 0, ctx->stream, dRaw3di, ctx->db->dOffsets, ctx->db->dLengths,
                           (uint32_t) n, ctx->db->stripeOff, ctx->db->stripeLen, ctx->db->stripeTargets, ctx->db->scan);
        HIPCHK(hipGetLastError());
    }
    if (bytes) {
        HIPCHK(hipMalloc((void **) &ctx->db->raw3di, bytes));
        HIPCHK(hipMemcpyAsync(ctx->db->raw3di, dRaw3di, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(k_db_unmask, dim3(2048), dim3(256), 0, ctx->stream, dRaw3di, ctx->db->aln3di, bytes);
        HIPCHK(hipGetLastError());
        if (dRawAA) {
            HIPCHK(hipMalloc((void **) &ctx->db->alnAA, bytes));
            hipLaunchKernelGGL(k_db_unmask, dim3(2048), dim3(256), 0, ctx->stream, dRawAA, ctx->db->alnAA, bytes);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->db->scanU4 = total;
    ctx->db->n = n; ctx->db->bytes = bytes; ctx->db->residues = residues; ctx->db->nStripes = nStripes; ctx->db->maxLen = maxLen;
    ctx->db->hasAA = dRawAA != nullptr;
    return FSGPU_OK;
}

extern "C" {

int fsgpu_db_adopt_device(fsgpu_ctx *ctx, const void *d3, const void *dA, const void *dOff, const void *dLen,
                          uint64_t n, uint64_t bytes) {
    if (!ctx) return FSGPU_E_ARG;
    if (!d3 || !dOff || !dLen || n == 0 || n > 0xfffffff0ull) { ctx->err = "fsgpu_db_adopt_device: bad argument"; return FSGPU_E_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    freeDb(ctx);
    int rc = buildDb(ctx, (const uint8_t *) d3, (const uint8_t *) dA, (const uint64_t *) dOff, (const int32_t *) dLen, n, bytes);
    if (rc != FSGPU_OK) freeDb(ctx);
    return rc;
}

int fsgpu_db_load(fsgpu_ctx *ctx, const uint8_t *data3di, const uint8_t *dataAA, const uint64_t *offsets,
                  const int32_t *lengths, uint64_t n, uint64_t bytes) {
    if (!ctx) return FSGPU_E_ARG;
    if (!data3di || !offsets || !lengths || n == 0) { ctx->err = "fsgpu_db_load: bad argument"; return FSGPU_E_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    uint8_t *r3 = nullptr, *rA = nullptr;
    uint64_t *dO = nullptr;
    int32_t *dL = nullptr;
    HIPCHK(hipMalloc((void **) &r3, std::max<uint64_t>(bytes, 1)));
    HIPCHK(hipMemcpy(r3, data3di, bytes, hipMemcpyHostToDevice));
    if (dataAA) {
        HIPCHK(hipMalloc((void **) &rA, std::max<uint64_t>(bytes, 1)));
        HIPCHK(hipMemcpy(rA, dataAA, bytes, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc((void **) &dO, (n + 1) * sizeof(uint64_t)));
    HIPCHK(hipMemcpy(dO, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **) &dL, n * sizeof(int32_t)));
    HIPCHK(hipMemcpy(dL, lengths, n * sizeof(int32_t), hipMemcpyHostToDevice));
    int rc = fsgpu_db_adopt_device(ctx, r3, rA, dO, dL, n, bytes);
    hipFree(r3); hipFree(rA); hipFree(dO); hipFree(dL);
    return rc;
}

} // extern "C"



// ------------------------------------------------------------------------------------------------------------
// One node, several GPUs, one process: replicate the resident database of `src` into contexts on other devices
// with ONE broadcast per buffer (RCCL over xGMI, single-process communicator set; librccl is loaded on demand so
// that single-GPU use has no dependency on it) or, when RCCL cannot be loaded, with peer copies.
// ------------------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    bool load() {
        if (getenv("FSGPU_NO_RCCL")) return false;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll)) dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy)) dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart)) dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd)) dlsym(lib, "ncclGroupEnd");
        Broadcast = (decltype(Broadcast)) dlsym(lib, "ncclBroadcast");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast;
    }
};
} // namespace

// librccl on this context's device, alone: a one-rank communicator broadcasts a 1 MiB buffer in place.  What a single-GPU box can show of
// the multi-GPU replication path: the library loads, a communicator comes up on the device, a grouped ncclBroadcast runs on the
// context's stream (FSGPU_REQUIRE_RCCL=1 makes the modules call it even when one GPU is used).
extern "C" int fsgpu_rccl_selfcheck(fsgpu_ctx *ctx) {
    if (!ctx) return FSGPU_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    Rccl rccl;
    if (!rccl.load()) { ctx->err = "fsgpu_rccl_selfcheck: librccl could not be loaded (or FSGPU_NO_RCCL is set)"; return FSGPU_E_UNSUPPORTED; }
    const size_t bytes = 1 << 20;
    unsigned char *buf = nullptr;
    HIPCHK(hipMalloc((void **) &buf, bytes));
    std::vector<unsigned char> host(bytes);
    for (size_t i = 0; i < bytes; i++) host[i] = (unsigned char) (i * 131 + 7);
    int rc = FSGPU_OK;
    void *comm = nullptr;
    const int dev = ctx->device;
    if (hipMemcpy(buf, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { ctx->err = "fsgpu_rccl_selfcheck: copy failed"; rc = FSGPU_E_HIP; }
    else if (rccl.CommInitAll(&comm, 1, &dev) != 0) { ctx->err = "fsgpu_rccl_selfcheck: ncclCommInitAll failed"; rc = FSGPU_E_HIP; }
    else {
        const bool ok = rccl.GroupStart() == 0 && rccl.Broadcast(buf, buf, bytes, 1 /*ncclUint8*/, 0, comm, ctx->stream) == 0 && rccl.GroupEnd() == 0 &&
                        hipStreamSynchronize(ctx->stream) == hipSuccess;
        std::vector<unsigned char> back(bytes);
        if (!ok || hipMemcpy(back.data(), buf, bytes, hipMemcpyDeviceToHost) != hipSuccess || back != host) { ctx->err = "fsgpu_rccl_selfcheck: the one-rank ncclBroadcast failed"; rc = FSGPU_E_HIP; }
        rccl.CommDestroy(comm);
    }
    (void) hipFree(buf);
    return rc;
}

extern "C" int fsgpu_db_broadcast(fsgpu_ctx *src, fsgpu_ctx **dst, int n, int *usedRccl) {
    fsgpu_ctx *ctx = src;       // HIPCHK reports into the source context
    if (usedRccl) *usedRccl = 0;
    if (!src || (n > 0 && !dst) || n < 0) return FSGPU_E_ARG;
    if (!src->db || src->db->n == 0) { src->err = "fsgpu_db_broadcast: no database loaded"; return FSGPU_E_NODB; }
    if (n == 0) return FSGPU_OK;
    bool distinct = true;
    for (int i = 0; i < n; i++) {
        if (!dst[i] || dst[i] == src) { src->err = "fsgpu_db_broadcast: bad destination context"; return FSGPU_E_ARG; }
        distinct = distinct && dst[i]->device != src->device;      // a second copy on the same device is legal (tests), RCCL is not used for it
    }
    const DbStore &db = *src->db;
    const uint64_t nT = db.n, bytes = db.bytes;
    // the four inputs of buildDb as they live on the source device (the unmasked AA copy is a fixed point of k_db_unmask)
    struct Buf { const void *srcp; size_t size; std::vector<void *> dstp; };
    Buf bufs[4] = {{db.raw3di, (size_t) bytes, {}}, {db.hasAA ? db.alnAA : nullptr, db.hasAA ? (size_t) bytes : 0, {}},
                   {db.dOffsets, (size_t) (nT + 1) * sizeof(uint64_t), {}}, {db.dLengths, (size_t) nT * sizeof(int32_t), {}}};
    auto freeAll = [&]() {
        for (Buf &b : bufs) for (size_t i = 0; i < b.dstp.size(); i++) if (b.dstp[i]) { (void) hipSetDevice(dst[i]->device); (void) hipFree(b.dstp[i]); }
        (void) hipSetDevice(src->device);
    };
    for (Buf &b : bufs) {
        b.dstp.assign(n, nullptr);
        if (!b.size) continue;
        for (int i = 0; i < n; i++) {
            if (hipSetDevice(dst[i]->device) != hipSuccess || hipMalloc(&b.dstp[i], b.size) != hipSuccess) { freeAll(); src->err = "fsgpu_db_broadcast: out of device memory"; return FSGPU_E_NOMEM; }
        }
    }
    HIPCHK(hipSetDevice(src->device));
    HIPCHK(hipStreamSynchronize(src->stream));
    Rccl rccl;
    bool done = false;
    if (distinct && rccl.load()) {
        std::vector<int> devs(n + 1);
        devs[0] = src->device;
        for (int i = 0; i < n; i++) devs[i + 1] = dst[i]->device;
        std::vector<void *> comms(n + 1, nullptr);
        if (rccl.CommInitAll(comms.data(), n + 1, devs.data()) == 0) {
            bool ok = true;
            for (Buf &b : bufs) {
                if (!b.size) continue;
                ok = ok && rccl.GroupStart() == 0;
                for (int r = 0; r <= n && ok; r++) {
                    fsgpu_ctx *c = r == 0 ? src : dst[r - 1];
                    ok = hipSetDevice(c->device) == hipSuccess &&
                         rccl.Broadcast(b.srcp, r == 0 ? const_cast<void *>(b.srcp) : b.dstp[r - 1], b.size, 1 /*ncclUint8*/, 0, comms[r], c->stream) == 0;
                }
                ok = (rccl.GroupEnd() == 0) && ok;
            }
            for (int r = 0; r <= n; r++) {
                fsgpu_ctx *c = r == 0 ? src : dst[r - 1];
                ok = hipSetDevice(c->device) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess && ok;
            }
            for (void *c : comms) if (c) rccl.CommDestroy(c);
            done = ok;
            if (usedRccl) *usedRccl = ok ? 1 : 0;
        }
    }
    if (!done) {
        for (Buf &b : bufs) {
            if (!b.size) continue;
            for (int i = 0; i < n; i++)
                if (hipMemcpyPeer(b.dstp[i], dst[i]->device, b.srcp, src->device, b.size) != hipSuccess) { freeAll(); src->err = "fsgpu_db_broadcast: peer copy failed"; return FSGPU_E_HIP; }
        }
        // peer copies are device-side work on the null streams: not synchronous with the host, not ordered against the contexts' non-blocking streams
        for (int r = 0; r <= n; r++) {
            fsgpu_ctx *c = r == 0 ? src : dst[r - 1];
            if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { freeAll(); src->err = "fsgpu_db_broadcast: peer copy failed"; return FSGPU_E_HIP; }
        }
    }
    int rc = FSGPU_OK;
    for (int i = 0; i < n && rc == FSGPU_OK; i++) {
        rc = fsgpu_db_adopt_device(dst[i], bufs[0].dstp[i], bufs[1].dstp[i], bufs[2].dstp[i], bufs[3].dstp[i], nT, bytes);
        if (rc != FSGPU_OK) src->err = std::string("fsgpu_db_broadcast: ") + dst[i]->err;
    }
    freeAll();
    return rc;
}

// ------------------------------------------------------------------------------------------------------------
// gapless work list.  ov = warm-up chunks a column segment needs (= register count R of the query, 16 R >= Lq);
// ov = 0: whole stripes only (row-tiled long queries).  A stripe longer than `cap` chunks is cut into K segments of
// equal new length that each start ov chunks early; cap minimises max(cap, (work + warm-up work) / waves), the
// completion time of a longest-first queue over equally fast waves.
// ------------------------------------------------------------------------------------------------------------
// pure planning step (host only, no device calls; exported as fsgpu_gapless_plan_items for the CPU tests)
static void planGaplessItems(const std::vector<uint32_t> &len, int ov, double waves, bool allowSplit, std::vector<uint64_t> &v, bool &split, uint32_t &capOut) {
    const uint32_t nStripes = (uint32_t) len.size();
    uint64_t total = 0;
    uint32_t maxLen = 0;
    for (uint32_t x : len) { total += x; maxLen = std::max(maxLen, x); }
    uint32_t cap = maxLen;
    if (ov > 0 && maxLen > 2u * ov && allowSplit) {
        // histogram of stripe lengths -> cost of every candidate cap
        std::vector<uint32_t> hist(maxLen + 1, 0);
        for (uint32_t x : len) hist[x]++;
        double best = std::max((double) maxLen, (double) total / waves);
        for (uint32_t c = 2u * ov; c < maxLen; c++) {
            uint64_t extra = 0;
            const uint32_t fresh = c - ov;
            for (uint32_t x = c + 1; x <= maxLen; x++)
                if (hist[x]) extra += (uint64_t) hist[x] * ((x + fresh - 1) / fresh - 1) * ov;
            const double t = std::max((double) c, (double) (total + extra) / waves);
            if (t < best) { best = t; cap = c; }
        }
    }
    capOut = cap;
    v.clear();
    v.reserve(nStripes + 64);
    split = false;
    for (uint32_t s = 0; s < nStripes; s++) {
        const uint32_t L = len[s];
        if (L == 0) continue;
        if (L <= cap || ov == 0) { v.push_back(((uint64_t) s << 32) | L); continue; }
        const uint32_t K = (L + (cap - ov) - 1) / (cap - ov), fresh = (L + K - 1) / K;
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t b = k * fresh, e = std::min(L, (k + 1) * fresh);
            if (b >= e) break;
            const uint32_t b0 = b > (uint32_t) ov ? b - ov : 0;
            v.push_back(((uint64_t) s << 32) | (1ull << 31) | ((uint64_t) b0 << 16) | e);
            split = true;
        }
    }
    std::stable_sort(v.begin(), v.end(), [](uint64_t a, uint64_t b) {
        const uint32_t la = (uint32_t) (a & 0xffff) - (uint32_t) ((a >> 16) & 0x7fff), lb = (uint32_t) (b & 0xffff) - (uint32_t) ((b >> 16) & 0x7fff);
        return la > lb;
    });
}

extern "C" int64_t fsgpu_gapless_plan_items(const uint32_t *stripeLen, uint32_t nStripes, int overlap, double waves, uint64_t *items, uint64_t capacity, uint32_t *cap) {
    if ((!stripeLen && nStripes) || overlap < 0 || waves <= 0) return -1;
    std::vector<uint32_t> len(stripeLen, stripeLen + nStripes);
    std::vector<uint64_t> v;
    bool split = false;
    uint32_t c = 0;
    planGaplessItems(len, overlap, waves, true, v, split, c);
    if (cap) *cap = c;
    if (items) for (size_t i = 0; i < v.size() && i < capacity; i++) items[i] = v[i];
    return (int64_t) v.size();
}

static int gaplessItems(fsgpu_ctx *ctx, int ov, const uint4 **items, uint32_t *nItems, bool *anySplit) {
    DbStore &db = *ctx->db;
    std::lock_guard<std::mutex> lock(db.itemMutex);
    DbStore::ItemList &l = db.itemLists[ov];
    if (!l.built) {
        const std::vector<uint32_t> &len = db.hStripeLen;
        const uint32_t nStripes = (uint32_t) len.size();
        std::vector<uint64_t> v;
        bool split = false;
        uint32_t cap = 0;
        planGaplessItems(len, ov, (double) ctx->numCU * 3 * (kGaplessBlock / 64), !getenv("FSGPU_GAPLESS_NOSPLIT") /* A/B measurements */, v, split, cap);
        // device record: {stripe, range word, stripe offset in the scan layout (uint4 units) lo, hi}
        std::vector<uint4> rec(v.size());
        {
            std::vector<uint64_t> sOff(nStripes);
            uint64_t acc = 0;
            for (uint32_t s = 0; s < nStripes; s++) { sOff[s] = acc; acc += (uint64_t) len[s] * 8; }
            for (size_t i = 0; i < v.size(); i++) {
                const uint32_t st = (uint32_t) (v[i] >> 32);
                rec[i] = make_uint4(st, (uint32_t) v[i], (uint32_t) sOff[st], (uint32_t) (sOff[st] >> 32));
            }
        }
        HIPCHK(hipMalloc((void **) &l.items, std::max<size_t>(rec.size(), 1) * sizeof(uint4)));
        if (!rec.empty()) {
            const hipError_t ce = hipMemcpy(l.items, rec.data(), rec.size() * sizeof(uint4), hipMemcpyHostToDevice);
            if (ce != hipSuccess) { (void) hipFree(l.items); l.items = nullptr; ctx->err = std::string("hipMemcpy(work items): ") + hipGetErrorString(ce); return FSGPU_E_HIP; }
        }
        l.n = (uint32_t) v.size(); l.split = split; l.built = true;
    }
    *items = l.items; *nItems = l.n; *anySplit = l.split;
    return FSGPU_OK;
}

// ------------------------------------------------------------------------------------------------------------
// gapless scan
// ------------------------------------------------------------------------------------------------------------
template <int R, bool TILED, bool PAIRED = false>
static int launchGapless(fsgpu_ctx *ctx, const GaplessArgs &gaIn) {
    GaplessArgs ga = gaIn;
    const int lds = gaplessLdsBytes(R);
    static thread_local uint64_t attrDevs = 0;       // devices on which this thread has set the attribute (it is per device)
    static thread_local int perCUcached = 0;
    const uint64_t devBit = 1ull << (ctx->device & 63);
    if (!(attrDevs & devBit)) {
        HIPCHK(hipFuncSetAttribute((const void *) k_gapless<R, TILED, PAIRED>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCUcached, k_gapless<R, TILED, PAIRED>, gaplessBlockThreads(R), lds));
        attrDevs |= devBit;
    }
    int perCU = perCUcached;
    // Workgroups of 4 waves, each with its own LDS image; 3 per CU (12 waves, <= 135 KB LDS): more does not issue faster
    // (profiles/r01_q_gapless_ablation_ubench.txt, tools/bench_ab2.sh) and this leaves wave slots for the latency-bound SW
    // wavefront kernels of other in-flight queries to co-reside.  FSGPU_GAPLESS_BLOCKS_PER_CU overrides.
    constexpr int wavesPerBlock = gaplessBlockThreads(R) / 64;
    perCU = std::max(1, std::min(perCU, ctx->gaplessBlocksPerCU));
    // multi-query launch: 2 workgroups per CU and query (the third resident slot goes to the next query's workgroups, which
    // start while this query's tail drains): 2.75 vs 2.79 ms per query at 1M targets, tools: FSGPU_GAPLESS_BLOCKS_PER_CU sweep
    if (gaIn.queries) {
        static const int multiPerCU = [] { const char *e = getenv("FSGPU_GAPLESS_MULTI_BLOCKS_PER_CU"); return e ? std::max(1, atoi(e)) : 2; }();
        perCU = std::min(perCU, multiPerCU);
    }
    // one wave needs one stripe at a time: do not launch more waves than stripes
    uint32_t blocks = (uint32_t) std::min<uint64_t>((uint64_t) ctx->numCU * perCU, ((uint64_t) ga.nItems + wavesPerBlock - 1) / wavesPerBlock);
    blocks = std::max(blocks, 1u);
    // multi-query launch: ga.blocksPerQuery carries the number of queries on entry; every query gets `blocks` workgroups
    const uint32_t nQueries = ga.queries ? std::max(1u, ga.blocksPerQuery) : 1u;
    ga.blocksPerQuery = blocks;
    hipLaunchKernelGGL((k_gapless<R, TILED, PAIRED>), dim3(blocks * nQueries), dim3(gaplessBlockThreads(R)), lds, ctx->stream, ga);
    HIPCHK(hipGetLastError());
    return FSGPU_OK;
}

extern "C" {

int fsgpu_gapless_launch(fsgpu_ctx *ctx, const int8_t *pssm, int L, int scoreCap, int minScore, int64_t identityId, int maxRes) {
    if (!ctx) return FSGPU_E_ARG;
    ctx->mqScanMs = -1.0;                                  // fsgpu_last_kernel_ms(ctx, 0) reads this call's own events again
    if (!pssm || L <= 0 || L > FSGPU_MAX_SEQ_LEN || maxRes <= 0) { ctx->err = "fsgpu_gapless_launch: bad argument"; return FSGPU_E_ARG; }
    if (!ctx->db || ctx->db->n == 0) { ctx->err = "no database loaded"; return FSGPU_E_NODB; }
    if (ctx->gaplessPending) { ctx->err = "previous gapless scan not finished"; return FSGPU_E_ARG; }
    const int rows = (L + 15) / 16;                       // rows per strip per lane
    // up to 16 * kGaplessMaxRUntiled rows in one piece; beyond that row tiles of at most 512 rows and equal height: L = 1025 runs as
    // 3 x 352 rows (R = 22), not as 512 + 512 + 1 rows at the full R = 32 cost each
    const int nTiles = L <= 16 * kGaplessMaxRUntiled ? 1 : (L + 16 * kGaplessMaxR - 1) / (16 * kGaplessMaxR);
    const int R = nTiles > 1 ? ((L + nTiles - 1) / nTiles + 15) / 16 : std::max(1, rows);
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t n = (uint32_t) ctx->db->n;
    const uint32_t nChunks = (n + kSelChunk - 1) / kSelChunk;
    const uint32_t K = (uint32_t) std::min<uint64_t>((uint64_t) maxRes, ctx->db->n);
    int rc;
    if ((rc = ensure(ctx, ctx->pssm, (size_t) kAlphabet * L)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->scores, n)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->chunkHist, (size_t) nChunks * 256 * 4)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->baseGt, (size_t) nChunks * 4)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->baseTie, (size_t) nChunks * 4)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->outId, (size_t) K * 4)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->outScore, (size_t) K * 4)) != FSGPU_OK) return rc;
    if (nTiles > 1) {
        // border rows between query row tiles: 2 bytes per padded target column, ping-pong
        const size_t bbytes = (size_t) ctx->db->scanU4 * 32;
        if ((rc = ensure(ctx, ctx->gBorder0, bbytes)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->gBorder1, bbytes)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->scoreAcc, (size_t) n * 2)) != FSGPU_OK) return rc;
    }
    if ((rc = ensurePinned(ctx, ctx->hOutId, (size_t) K * 4)) != FSGPU_OK) return rc;
    if ((rc = ensurePinned(ctx, ctx->hOutScore, (size_t) K * 4)) != FSGPU_OK) return rc;
    if ((rc = ensurePinned(ctx, ctx->hPssm, (size_t) kAlphabet * L)) != FSGPU_OK) return rc;
    memcpy(ctx->hPssm.p, pssm, (size_t) kAlphabet * L);
    HIPCHK(hipMemcpyAsync(ctx->pssm.p, ctx->hPssm.p, (size_t) kAlphabet * L, hipMemcpyHostToDevice, ctx->stream));
    GaplessArgs ga;
    ga.queries = nullptr; ga.blocksPerQuery = 0;
    ga.scan = ctx->db->scan; ga.stripeOff = ctx->db->stripeOff; ga.stripeLen = ctx->db->stripeLen; ga.stripeTargets = ctx->db->stripeTargets;
    bool anySplit = false;
    if ((rc = gaplessItems(ctx, nTiles > 1 ? 0 : R, &ga.items, &ga.nItems, &anySplit)) != FSGPU_OK) return rc;
    ga.nTargets = n; ga.pssm = (const int8_t *) ctx->pssm.p; ga.L = L;
    ga.cap = std::max(0, std::min(scoreCap, 255));
    ga.scores = (uint8_t *) ctx->scores.p; ga.queue = ctx->queue;
    ga.tileBase = 0; ga.firstTile = 1; ga.lastTile = 1; ga.borderIn = nullptr; ga.borderOut = nullptr; ga.scoreAcc = (int16_t *) ctx->scoreAcc.p;
    if (anySplit) HIPCHK(hipMemsetAsync(ctx->scores.p, 0, n, ctx->stream));     // column segments combine by atomic max
    HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
    if (nTiles == 1) {
        HIPCHK(hipMemsetAsync(ctx->queue, 0, 4, ctx->stream));
        // one instantiation per register count: a query of L residues runs with R = ceil(L / 16) (16-row granularity)
        using LaunchFn = int (*)(fsgpu_ctx *, const GaplessArgs &);
        static const LaunchFn table[kGaplessMaxRUntiled + 1] = {nullptr,
            launchGapless<1, false>, launchGapless<2, false>, launchGapless<3, false>, launchGapless<4, false>,
            launchGapless<5, false>, launchGapless<6, false>, launchGapless<7, false>, launchGapless<8, false>,
            launchGapless<9, false>, launchGapless<10, false>, launchGapless<11, false>, launchGapless<12, false>,
            launchGapless<13, false>, launchGapless<14, false>, launchGapless<15, false>, launchGapless<16, false>,
            launchGapless<17, false>, launchGapless<18, false>, launchGapless<19, false>, launchGapless<20, false>,
            launchGapless<21, false>, launchGapless<22, false>, launchGapless<23, false>, launchGapless<24, false>,
            launchGapless<25, false>, launchGapless<26, false>, launchGapless<27, false>, launchGapless<28, false>,
            launchGapless<29, false>, launchGapless<30, false>, launchGapless<31, false>, launchGapless<32, false>,
            launchGapless<33, false>, launchGapless<34, false>, launchGapless<35, false>, launchGapless<36, false>,
            launchGapless<37, false>, launchGapless<38, false>, launchGapless<39, false>, launchGapless<40, false>,
            launchGapless<41, false>, launchGapless<42, false>, launchGapless<43, false>, launchGapless<44, false>,
            launchGapless<45, false>, launchGapless<46, false>, launchGapless<47, false>, launchGapless<48, false>,
            launchGapless<49, false>, launchGapless<50, false>, launchGapless<51, false>, launchGapless<52, false>,
            launchGapless<53, false>, launchGapless<54, false>, launchGapless<55, false>, launchGapless<56, false>};
        if (R < 1 || R > kGaplessMaxRUntiled) { ctx->err = "internal: bad R"; return FSGPU_E_ARG; }
        rc = table[R](ctx, ga);
        if (rc != FSGPU_OK) return rc;
    } else {
        // query row tiles of 16 R <= 512 rows: tile t+1 continues every diagonal of tile t through the border arrays in HBM
        using LaunchFn = int (*)(fsgpu_ctx *, const GaplessArgs &);
        static const LaunchFn tiled[16] = {            // more than one tile means L > 896, so a tile has more than 256 rows: R = 17..32
            launchGapless<17, true>, launchGapless<18, true>, launchGapless<19, true>, launchGapless<20, true>,
            launchGapless<21, true>, launchGapless<22, true>, launchGapless<23, true>, launchGapless<24, true>,
            launchGapless<25, true>, launchGapless<26, true>, launchGapless<27, true>, launchGapless<28, true>,
            launchGapless<29, true>, launchGapless<30, true>, launchGapless<31, true>, launchGapless<32, true>};
        if (R < 17 || R > kGaplessMaxR) { ctx->err = "internal: bad tiled R"; return FSGPU_E_ARG; }
        for (int t = 0; t < nTiles; t++) {
            HIPCHK(hipMemsetAsync(ctx->queue, 0, 4, ctx->stream));
            ga.tileBase = t * 16 * R;
            ga.firstTile = t == 0; ga.lastTile = t == nTiles - 1;
            ga.borderIn = (const uint16_t *) ((t & 1) ? ctx->gBorder1.p : ctx->gBorder0.p);
            ga.borderOut = (uint16_t *) ((t & 1) ? ctx->gBorder0.p : ctx->gBorder1.p);
            if ((rc = tiled[R - 17](ctx, ga)) != FSGPU_OK) return rc;
        }
    }
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
    hipLaunchKernelGGL(k_sel_hist, dim3(nChunks), dim3(kSelThreads), 0, ctx->stream, (const uint8_t *) ctx->scores.p, n, minScore,
                       identityId, (uint32_t *) ctx->chunkHist.p, (const int64_t *) nullptr, (uint64_t) 0);
    hipLaunchKernelGGL(k_sel_threshold, dim3(1), dim3(256), 0, ctx->stream, (const uint32_t *) ctx->chunkHist.p, nChunks, K, ctx->dMeta,
                       (uint32_t *) ctx->baseGt.p, (uint32_t *) ctx->baseTie.p);
    hipLaunchKernelGGL(k_sel_emit, dim3(nChunks), dim3(kSelThreads), 0, ctx->stream, (const uint8_t *) ctx->scores.p, n, minScore,
                       identityId, (const SelMeta *) ctx->dMeta, (const uint32_t *) ctx->baseGt.p, (const uint32_t *) ctx->baseTie.p,
                       (uint32_t *) ctx->outId.p, (int32_t *) ctx->outScore.p, (const int64_t *) nullptr, (uint64_t) 0, K);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ctx->hMeta, ctx->dMeta, sizeof(SelMeta), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->hOutId.p, ctx->outId.p, (size_t) K * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->hOutScore.p, ctx->outScore.p, (size_t) K * 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->pendingMaxRes = (int) K;
    ctx->gaplessPending = true;
    ctx->evValid[0] = true;
    return FSGPU_OK;
}

int fsgpu_gapless_finish(fsgpu_ctx *ctx, fsgpu_hit *out, int *nout) {
    if (!ctx || !out || !nout) return FSGPU_E_ARG;
    if (!ctx->gaplessPending) { ctx->err = "no gapless scan in flight"; return FSGPU_E_ARG; }
    ctx->gaplessPending = false;
    HIPCHK(hipSetDevice(ctx->device));
    { int rc = syncStream(ctx); if (rc != FSGPU_OK) return rc; }
    const uint32_t m = std::min<uint32_t>(ctx->hMeta->nOut, (uint32_t) ctx->pendingMaxRes);
    for (uint32_t i = 0; i < m; i++) { out[i].id = ((const uint32_t *) ctx->hOutId.p)[i]; out[i].score = ((const int32_t *) ctx->hOutScore.p)[i]; }
    // hit_t::compareHitsByScoreAndId (scores are non-negative here)
    std::sort(out, out + m, [](const fsgpu_hit &a, const fsgpu_hit &b) {
        if (a.score != b.score) return a.score > b.score;
        return a.id < b.id;
    });
    *nout = (int) m;
    return FSGPU_OK;
}

int fsgpu_gapless_scan(fsgpu_ctx *ctx, const int8_t *pssm, int L, int scoreCap, int minScore, int64_t identityId, int maxRes,
                       fsgpu_hit *out, int *nout) {
    int rc = fsgpu_gapless_launch(ctx, pssm, L, scoreCap, minScore, identityId, maxRes);
    if (rc != FSGPU_OK) return rc;
    return fsgpu_gapless_finish(ctx, out, nout);
}

// Several queries, one resident-DB pass each, in as few launches as their lengths allow: queries of one register class
// (R = ceil(L / 16)) share ONE launch of k_gapless (GaplessQuery records; workgroups of query q + 1 move in as those of
// query q drain), the three selection passes run once for the whole batch (blockIdx.y = query).  Results are those of nq
// fsgpu_gapless_scan calls.  Queries longer than 512 residues (row tiles) go through the single-query path.
int fsgpu_gapless_scan_multi(fsgpu_ctx *ctx, const fsgpu_gapless_query *q, int nq, int minScore, int maxRes, fsgpu_hit *out, int *nout) {
    if (!ctx) return FSGPU_E_ARG;
    if (nq < 0 || maxRes <= 0 || (nq > 0 && (!q || !out || !nout))) { ctx->err = "fsgpu_gapless_scan_multi: bad argument"; return FSGPU_E_ARG; }
    if (!ctx->db || ctx->db->n == 0) { ctx->err = "no database loaded"; return FSGPU_E_NODB; }
    if (ctx->gaplessPending) { ctx->err = "previous gapless scan not finished"; return FSGPU_E_ARG; }
    for (int i = 0; i < nq; i++)
        if (!q[i].pssm || q[i].L <= 0 || q[i].L > FSGPU_MAX_SEQ_LEN) { ctx->err = "fsgpu_gapless_scan_multi: bad query"; return FSGPU_E_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    ctx->mqLaunches = 0; ctx->mqQueries = 0; ctx->mqScanMs = -1.0;
    ctx->mqSlot.assign(nq, -1);
    // ---- device batch: the single-tile queries, grouped by register class ----
    std::vector<int> batch, longQ;
    for (int i = 0; i < nq; i++) (q[i].L <= 16 * kGaplessMaxRUntiled ? batch : longQ).push_back(i);
    std::stable_sort(batch.begin(), batch.end(), [&](int x, int y) { return (q[x].L + 15) / 16 > (q[y].L + 15) / 16; });   // long queries first
    const int nb = (int) batch.size();
    const uint32_t n = (uint32_t) ctx->db->n;
    const uint32_t nChunks = (n + kSelChunk - 1) / kSelChunk;
    const uint32_t K = (uint32_t) std::min<uint64_t>((uint64_t) maxRes, ctx->db->n);
    const uint64_t scoreStride = ((uint64_t) n + 255) / 256 * 256;
    if (nb > 0) {
        int rc;
        std::vector<size_t> pOff(nb + 1, 0);
        for (int k = 0; k < nb; k++) pOff[k + 1] = pOff[k] + ((size_t) kAlphabet * q[batch[k]].L + 63) / 64 * 64;
        if ((rc = ensure(ctx, ctx->mqPssm, pOff[nb])) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->mqScores, scoreStride * nb)) != FSGPU_OK) return rc;
        const size_t nRecMax = (size_t) nb + (size_t) nb / 2 + 1;          // one record per query + one per pair of short queries
        if ((rc = ensure(ctx, ctx->mqQueues, nRecMax * 4)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->mqRec, nRecMax * sizeof(GaplessQuery))) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->mqHist, (size_t) nb * nChunks * 256 * 4)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->mqBaseGt, (size_t) nb * nChunks * 4)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->mqBaseTie, (size_t) nb * nChunks * 4)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->mqMeta, (size_t) nb * sizeof(SelMeta))) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->mqOutId, (size_t) nb * K * 4)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->mqOutScore, (size_t) nb * K * 4)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->mqIdent, (size_t) nb * 8)) != FSGPU_OK) return rc;
        if ((rc = ensurePinned(ctx, ctx->hMqPssm, pOff[nb])) != FSGPU_OK) return rc;
        if ((rc = ensurePinned(ctx, ctx->hMqRec, nRecMax * sizeof(GaplessQuery))) != FSGPU_OK) return rc;
        if ((rc = ensurePinned(ctx, ctx->hMqMeta, (size_t) nb * sizeof(SelMeta))) != FSGPU_OK) return rc;
        if ((rc = ensurePinned(ctx, ctx->hMqOutId, (size_t) nb * K * 4)) != FSGPU_OK) return rc;
        if ((rc = ensurePinned(ctx, ctx->hMqOutScore, (size_t) nb * K * 4)) != FSGPU_OK) return rc;
        if ((rc = ensurePinned(ctx, ctx->hMqIdent, (size_t) nb * 8)) != FSGPU_OK) return rc;
        ctx->mqScoreStride = scoreStride;
        GaplessQuery *rec = (GaplessQuery *) ctx->hMqRec.p;
        int64_t *ident = (int64_t *) ctx->hMqIdent.p;
        for (int k = 0; k < nb; k++) {
            const fsgpu_gapless_query &qq = q[batch[k]];
            memcpy((char *) ctx->hMqPssm.p + pOff[k], qq.pssm, (size_t) kAlphabet * qq.L);
            rec[k].pssm = (const int8_t *) ctx->mqPssm.p + pOff[k];
            rec[k].scores = (uint8_t *) ctx->mqScores.p + scoreStride * k;
            rec[k].queue = (uint32_t *) ctx->mqQueues.p + k;
            rec[k].L = qq.L;
            rec[k].cap = std::max(0, std::min(qq.scoreCap, 255));
            rec[k].pssmB = nullptr; rec[k].scoresB = nullptr; rec[k].LB = 0; rec[k].capB = 0;
            ident[k] = qq.identityId;
            ctx->mqSlot[batch[k]] = k;
        }
        // Short queries (<= 256 residues) of one 16-row class run two to a kernel (k_gapless<2R, false, PAIRED>: the per-column
        // instructions that do not scale with the rows are shared); an odd one out runs alone.  FSGPU_GAPLESS_PAIR=0: A/B switch.
        struct PairLaunch { int Rq; size_t rec0; int count; };
        std::vector<PairLaunch> pairLaunches;
        std::vector<char> isPaired(nb, 0);
        size_t nRec = (size_t) nb;
        {
            static const bool pairing = [] { const char *e = getenv("FSGPU_GAPLESS_PAIR"); return !e || atoi(e) != 0; }();
            // classes up to 16 registers (256 residues): beyond that the pair would need the 6-wave workgroups of R > 36, which was measured
            // and loses (pairs up to class 20 / 24 / 28 at 1M targets: 2.74 / 2.77 / 2.84 ms per query against 2.76 without)
            static const int pairMaxR = [] { const char *e = getenv("FSGPU_GAPLESS_PAIR_MAXR"); return e ? std::max(1, std::min(atoi(e), kGaplessMaxR / 2)) : kGaplessMaxR / 2; }();
            for (int k0 = 0; pairing && k0 < nb;) {
                const int R = std::max(1, (q[batch[k0]].L + 15) / 16);
                int k1 = k0;
                while (k1 < nb && std::max(1, (q[batch[k1]].L + 15) / 16) == R) k1++;
                if (R <= pairMaxR && k1 - k0 >= 2) {
                    PairLaunch pl{R, nRec, (k1 - k0) / 2};
                    for (int p2 = 0; p2 < pl.count; p2++) {
                        const int ka = k0 + 2 * p2, kb = ka + 1;
                        rec[nRec] = rec[ka];
                        rec[nRec].queue = (uint32_t *) ctx->mqQueues.p + nRec;
                        rec[nRec].pssmB = rec[kb].pssm; rec[nRec].scoresB = rec[kb].scores; rec[nRec].LB = rec[kb].L; rec[nRec].capB = rec[kb].cap;
                        isPaired[ka] = isPaired[kb] = 1;
                        nRec++;
                    }
                    pairLaunches.push_back(pl);
                }
                k0 = k1;
            }
        }
        // One scan batch at a time per database, ordered ON THE DEVICE: every launch already fills the chip, two batches in
        // flight would only stretch each other.  The inputs go up first (they may overlap the previous owner's scans), then
        // -- under the mutex, which covers enqueueing only -- this stream is made to wait for the event the previous batch's
        // owner recorded behind its last scan launch, the scans are enqueued and this batch's event takes its place.  The
        // selection passes, copies and the host wait happen outside the mutex; SW / selection kernels of other contexts
        // co-run in the slots a scan leaves.  FSGPU_SCAN_EXCLUSIVE=0 drops the ordering (A/B measurements).
        static const bool exclusive = [] { const char *e = getenv("FSGPU_SCAN_EXCLUSIVE"); return !e || atoi(e) != 0; }();
        if (!ctx->scanDoneEv) HIPCHK(hipEventCreateWithFlags(&ctx->scanDoneEv, hipEventDisableTiming));
        HIPCHK(hipMemcpyAsync(ctx->mqPssm.p, ctx->hMqPssm.p, pOff[nb], hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->mqRec.p, ctx->hMqRec.p, nRec * sizeof(GaplessQuery), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->mqIdent.p, ctx->hMqIdent.p, (size_t) nb * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->mqQueues.p, 0, nRec * 4, ctx->stream));
        std::unique_lock<std::mutex> scanLock(ctx->db->scanMutex, std::defer_lock);
        using LaunchFn = int (*)(fsgpu_ctx *, const GaplessArgs &);
        static const LaunchFn table[kGaplessMaxRUntiled + 1] = {nullptr,
            launchGapless<1, false>, launchGapless<2, false>, launchGapless<3, false>, launchGapless<4, false>,
            launchGapless<5, false>, launchGapless<6, false>, launchGapless<7, false>, launchGapless<8, false>,
            launchGapless<9, false>, launchGapless<10, false>, launchGapless<11, false>, launchGapless<12, false>,
            launchGapless<13, false>, launchGapless<14, false>, launchGapless<15, false>, launchGapless<16, false>,
            launchGapless<17, false>, launchGapless<18, false>, launchGapless<19, false>, launchGapless<20, false>,
            launchGapless<21, false>, launchGapless<22, false>, launchGapless<23, false>, launchGapless<24, false>,
            launchGapless<25, false>, launchGapless<26, false>, launchGapless<27, false>, launchGapless<28, false>,
            launchGapless<29, false>, launchGapless<30, false>, launchGapless<31, false>, launchGapless<32, false>,
            launchGapless<33, false>, launchGapless<34, false>, launchGapless<35, false>, launchGapless<36, false>,
            launchGapless<37, false>, launchGapless<38, false>, launchGapless<39, false>, launchGapless<40, false>,
            launchGapless<41, false>, launchGapless<42, false>, launchGapless<43, false>, launchGapless<44, false>,
            launchGapless<45, false>, launchGapless<46, false>, launchGapless<47, false>, launchGapless<48, false>,
            launchGapless<49, false>, launchGapless<50, false>, launchGapless<51, false>, launchGapless<52, false>,
            launchGapless<53, false>, launchGapless<54, false>, launchGapless<55, false>, launchGapless<56, false>};
        struct Group { int R, k0, k1; const uint4 *items; uint32_t nItems; };
        std::vector<Group> groups;
        bool anySplitAtAll = false;
        for (int k0 = 0; k0 < nb;) {
            const int R = std::max(1, (q[batch[k0]].L + 15) / 16);
            int k1 = k0;
            while (k1 < nb && std::max(1, (q[batch[k1]].L + 15) / 16) == R) k1++;
            int kFree = k0;                                   // the paired queries of a class are its first ones
            while (kFree < k1 && isPaired[kFree]) kFree++;
            Group g{R, kFree, k1, nullptr, 0};
            bool split = false;
            if ((rc = gaplessItems(ctx, R, &g.items, &g.nItems, &split)) != FSGPU_OK) return rc;
            anySplitAtAll = anySplitAtAll || split;
            if (kFree < k1) groups.push_back(g);
            k0 = k1;
        }
        // column segments combine by atomic max into zeroed score bytes: clear all slices BEFORE the first launch (a memset
        // between launches would wipe what earlier groups stored)
        if (anySplitAtAll) HIPCHK(hipMemsetAsync(ctx->mqScores.p, 0, scoreStride * nb, ctx->stream));
        if (exclusive) {
            scanLock.lock();
            if (ctx->db->lastScanDone && ctx->db->lastScanDone != ctx->scanDoneEv) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->db->lastScanDone, 0));
        }
        HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
        for (const Group &g : groups) {
            GaplessArgs ga;
            ga.scan = ctx->db->scan; ga.stripeOff = ctx->db->stripeOff; ga.stripeLen = ctx->db->stripeLen; ga.stripeTargets = ctx->db->stripeTargets;
            ga.items = g.items; ga.nItems = g.nItems;
            ga.queries = (const GaplessQuery *) ctx->mqRec.p + g.k0;
            ga.blocksPerQuery = (uint32_t) (g.k1 - g.k0);         // number of queries on entry, see launchGapless
            ga.nTargets = n; ga.pssm = nullptr; ga.L = 0; ga.cap = 0; ga.scores = nullptr; ga.queue = nullptr;
            ga.tileBase = 0; ga.firstTile = 1; ga.lastTile = 1; ga.borderIn = nullptr; ga.borderOut = nullptr; ga.scoreAcc = nullptr;
            if ((rc = table[g.R](ctx, ga)) != FSGPU_OK) return rc;
            ctx->mqLaunches++;
        }
        for (const PairLaunch &pl : pairLaunches) {
            static const LaunchFn paired[kGaplessMaxR / 2 + 1] = {nullptr,
                launchGapless<2, false, true>, launchGapless<4, false, true>, launchGapless<6, false, true>, launchGapless<8, false, true>,
                launchGapless<10, false, true>, launchGapless<12, false, true>, launchGapless<14, false, true>, launchGapless<16, false, true>,
                launchGapless<18, false, true>, launchGapless<20, false, true>, launchGapless<22, false, true>, launchGapless<24, false, true>,
                launchGapless<26, false, true>, launchGapless<28, false, true>, launchGapless<30, false, true>, launchGapless<32, false, true>};
            GaplessArgs ga;
            ga.scan = ctx->db->scan; ga.stripeOff = ctx->db->stripeOff; ga.stripeLen = ctx->db->stripeLen; ga.stripeTargets = ctx->db->stripeTargets;
            bool split = false;
            if ((rc = gaplessItems(ctx, pl.Rq, &ga.items, &ga.nItems, &split)) != FSGPU_OK) return rc;     // warm-up of a column segment = the QUERY's rows
            ga.queries = (const GaplessQuery *) ctx->mqRec.p + pl.rec0;
            ga.blocksPerQuery = (uint32_t) pl.count;
            ga.nTargets = n; ga.pssm = nullptr; ga.L = 0; ga.cap = 0; ga.scores = nullptr; ga.queue = nullptr;
            ga.tileBase = 0; ga.firstTile = 1; ga.lastTile = 1; ga.borderIn = nullptr; ga.borderOut = nullptr; ga.scoreAcc = nullptr;
            if ((rc = paired[pl.Rq](ctx, ga)) != FSGPU_OK) return rc;
            ctx->mqLaunches++;
        }
        HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
        if (exclusive) {
            HIPCHK(hipEventRecord(ctx->scanDoneEv, ctx->stream));
            ctx->db->lastScanDone = ctx->scanDoneEv;
            scanLock.unlock();
        }
        ctx->evValid[0] = true;
        ctx->mqQueries = nb;
        hipLaunchKernelGGL(k_sel_hist, dim3(nChunks, nb), dim3(kSelThreads), 0, ctx->stream, (const uint8_t *) ctx->mqScores.p, n, minScore,
                           (int64_t) -1, (uint32_t *) ctx->mqHist.p, (const int64_t *) ctx->mqIdent.p, scoreStride);
        hipLaunchKernelGGL(k_sel_threshold, dim3(nb), dim3(256), 0, ctx->stream, (const uint32_t *) ctx->mqHist.p, nChunks, K, (SelMeta *) ctx->mqMeta.p,
                           (uint32_t *) ctx->mqBaseGt.p, (uint32_t *) ctx->mqBaseTie.p);
        hipLaunchKernelGGL(k_sel_emit, dim3(nChunks, nb), dim3(kSelThreads), 0, ctx->stream, (const uint8_t *) ctx->mqScores.p, n, minScore,
                           (int64_t) -1, (const SelMeta *) ctx->mqMeta.p, (const uint32_t *) ctx->mqBaseGt.p, (const uint32_t *) ctx->mqBaseTie.p,
                           (uint32_t *) ctx->mqOutId.p, (int32_t *) ctx->mqOutScore.p, (const int64_t *) ctx->mqIdent.p, scoreStride, K);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ctx->hMqMeta.p, ctx->mqMeta.p, (size_t) nb * sizeof(SelMeta), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->hMqOutId.p, ctx->mqOutId.p, (size_t) nb * K * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->hMqOutScore.p, ctx->mqOutScore.p, (size_t) nb * K * 4, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = syncStream(ctx)) != FSGPU_OK) return rc;
        const SelMeta *meta = (const SelMeta *) ctx->hMqMeta.p;
        for (int k = 0; k < nb; k++) {
            const int qi = batch[k];
            fsgpu_hit *o = out + (size_t) qi * maxRes;
            const uint32_t m = std::min<uint32_t>(meta[k].nOut, K);
            const uint32_t *ids = (const uint32_t *) ctx->hMqOutId.p + (size_t) k * K;
            const int32_t *sc = (const int32_t *) ctx->hMqOutScore.p + (size_t) k * K;
            for (uint32_t i = 0; i < m; i++) { o[i].id = ids[i]; o[i].score = sc[i]; }
            std::sort(o, o + m, [](const fsgpu_hit &a, const fsgpu_hit &b) {      // hit_t::compareHitsByScoreAndId
                if (a.score != b.score) return a.score > b.score;
                return a.id < b.id;
            });
            nout[qi] = (int) m;
        }
    }
    // scan time of the whole call (fsgpu_last_kernel_ms(ctx, 0)): the batch's launches plus the row-tiled scans of the long queries, which
    // run one at a time and reuse the same pair of events
    double scanMs = nb > 0 ? fsgpu_last_kernel_ms(ctx, 0) : 0.0;
    for (int qi : longQ) {
        const int rc = fsgpu_gapless_scan(ctx, q[qi].pssm, q[qi].L, q[qi].scoreCap, minScore, q[qi].identityId, maxRes, out + (size_t) qi * maxRes, &nout[qi]);
        if (rc != FSGPU_OK) return rc;
        const double ms = fsgpu_last_kernel_ms(ctx, 0);
        if (ms >= 0 && scanMs >= 0) scanMs += ms; else scanMs = -1.0;
        ctx->mqLaunches++; ctx->mqQueries++;
    }
    ctx->mqScanMs = scanMs;
    return FSGPU_OK;
}

int fsgpu_gapless_scores_multi(fsgpu_ctx *ctx, int queryIndex, uint8_t *scores_out) {
    if (!ctx || !scores_out) return FSGPU_E_ARG;
    if (!ctx->db || queryIndex < 0 || queryIndex >= (int) ctx->mqSlot.size() || ctx->mqSlot[queryIndex] < 0) { ctx->err = "no batched scan results for this query"; return FSGPU_E_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(scores_out, (const uint8_t *) ctx->mqScores.p + ctx->mqScoreStride * (uint64_t) ctx->mqSlot[queryIndex], ctx->db->n, hipMemcpyDeviceToHost));
    return FSGPU_OK;
}

int fsgpu_gapless_last_batch(const fsgpu_ctx *ctx, int *launches, int *queries) {
    if (!ctx) return FSGPU_E_ARG;
    if (launches) *launches = ctx->mqLaunches;
    if (queries) *queries = ctx->mqQueries;
    return FSGPU_OK;
}

int fsgpu_gapless_scores(fsgpu_ctx *ctx, uint8_t *scores_out) {
    if (!ctx || !scores_out) return FSGPU_E_ARG;
    if (!ctx->db || ctx->db->n == 0 || !ctx->scores.p) { ctx->err = "no scan results"; return FSGPU_E_NODB; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(scores_out, ctx->scores.p, ctx->db->n, hipMemcpyDeviceToHost));
    return FSGPU_OK;
}

/* out[2][4]: per direction (0 forward, 1 reversed query) of the last fsgpu_sw_multi_dir calls: device ms of the pass's k_sw2 launches (HIP events
 * on the context stream, -1 when the pass did not run), DP cells, pairs, VALU wave-instructions of the DP rows (the roofline denominator) */
void fsgpu_sw_last_passes(const fsgpu_ctx *ctx, double *out) {
    for (int d = 0; d < 2; d++) {
        float ms = -1;
        if (!ctx || !ctx->swDirValid[d] || hipEventElapsedTime(&ms, ctx->swDirEv[2 * d], ctx->swDirEv[2 * d + 1]) != hipSuccess) { ms = -1; (void) hipGetLastError(); }
        out[d * 4 + 0] = ms >= 0 ? ms + (float) ctx->swDirExtraMs[d] : ms;
        out[d * 4 + 1] = ctx ? ctx->swDirCells[d] : 0; out[d * 4 + 2] = ctx ? ctx->swDirPairs[d] : 0; out[d * 4 + 3] = ctx ? ctx->swDirWaveSteps[d] : 0;
    }
}

double fsgpu_last_kernel_ms(const fsgpu_ctx *ctx, int which) {
    if (ctx && which >= 2 && which < 14) return ctx->kmerMs[which - 2];
    if (!ctx || which < 0 || which > 1 || !ctx->evValid[which]) return -1.0;
    if (which == 0 && ctx->mqScanMs >= 0) return ctx->mqScanMs;      // a multi-query call with row-tiled queries: all of its scans
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[2 * which], ctx->ev[2 * which + 1]) != hipSuccess) return -1.0;
    return (double) ms;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------------------
// Smith-Waterman batch
// ------------------------------------------------------------------------------------------------------------
static int swPickR(int rows) {
    const int opts[] = {1, 2, 3, 4, 6, 8};   // 5 and 7 were measured: more launch groups + odd LDS chunking cost more than the padding they save
    for (int r : opts) if (64 * r >= rows) return r;
    return 8;
}

template <int R, bool HAS_AA, typename A>
static int launchSwT(fsgpu_ctx *ctx, const SwArgs &sa, int nPairs) {
    const int lds = (HAS_AA ? 2 : 1) * kAlphabet * swRowDwords(R) * 4;
    static thread_local uint64_t attrDevs = 0;       // devices on which this thread has set the attribute (it is per device)
    const uint64_t devBit = 1ull << (ctx->device & 63);
    if (!(attrDevs & devBit)) {
        HIPCHK(hipFuncSetAttribute((const void *) k_sw<R, HAS_AA, A>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attrDevs |= devBit;
    }
    // small batches (one query) -> 4 waves per workgroup so that all CUs get work; big batches -> 8
    const int waves = nPairs <= 4096 ? 4 : 8;
    const int blocks = (nPairs + waves - 1) / waves;
    hipLaunchKernelGGL((k_sw<R, HAS_AA, A>), dim3(blocks), dim3(waves * 64), lds, ctx->stream, sa);
    HIPCHK(hipGetLastError());
    return FSGPU_OK;
}

template <typename A>
static int launchSw(fsgpu_ctx *ctx, int R, bool hasAA, const SwArgs &sa, int nPairs) {
#define FS_SW_CASE(RR)                                                                   \
    case RR: return hasAA ? launchSwT<RR, true, A>(ctx, sa, nPairs) : launchSwT<RR, false, A>(ctx, sa, nPairs);
    switch (R) {
        FS_SW_CASE(1) FS_SW_CASE(2) FS_SW_CASE(3) FS_SW_CASE(4) FS_SW_CASE(6) FS_SW_CASE(8)
        default: ctx->err = "internal: bad SW R"; return FSGPU_E_ARG;
    }
#undef FS_SW_CASE
}

// k_sw2: two targets per wave, one direction (image with the extra "past the end" row)
template <int R, bool HAS_AA>
static int launchSwBlocks2T(fsgpu_ctx *ctx, const SwArgs &sa, int nBlocks, int pairsPerBlock, hipStream_t stream) {
    const int lds = (HAS_AA ? 2 : 1) * kSw2Rows * swRowDwords(R) * 4;
    static thread_local uint64_t attrDevs = 0;       // devices on which this thread has set the attribute (it is per device)
    const uint64_t devBit = 1ull << (ctx->device & 63);
    if (!(attrDevs & devBit)) {
        HIPCHK(hipFuncSetAttribute((const void *) k_sw2<R, HAS_AA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attrDevs |= devBit;
    }
    hipLaunchKernelGGL((k_sw2<R, HAS_AA>), dim3(nBlocks), dim3(32 * pairsPerBlock), lds, stream, sa);
    HIPCHK(hipGetLastError());
    return FSGPU_OK;
}
// Pairs of one query that share a workgroup (and one copy of the query's LDS image): two per wave.  8 pairs = 4 waves per image gives
// 4 waves per SIMD (3Di, 33 KB image at R = 6) / 2 per SIMD (3Di + AA).  More waves per image (16 / 32 pairs: 8 waves per SIMD) is SLOWER:
// the kernel is VALU-issue bound already at 4 waves per SIMD and a workgroup lasts as long as its longest pair (32 queries x 1000 targets,
// forward pass, alone on the device: 1.69 / 1.87 / 2.08 ms at 8 / 16 / 32 pairs for 3Di, 2.44 / 2.60 / 2.71 ms for 3Di + AA;
// tools/sw2_probe.py).  FSGPU_SW2_PAIRS = 8 | 16 | 32 overrides it for such measurements.
static int sw2PairsPerBlock() {
    static const int env = [] { const char *e = getenv("FSGPU_SW2_PAIRS"); const int v = e ? atoi(e) : 0; return (v == 8 || v == 16 || v == 32) ? v : 0; }();
    return env ? env : 8;
}
static int launchSwBlocks2(fsgpu_ctx *ctx, int R, bool hasAA, const SwArgs &sa, int nBlocks, int pairsPerBlock, hipStream_t stream) {
#define FS_SW_CASE(RR) case RR: return hasAA ? launchSwBlocks2T<RR, true>(ctx, sa, nBlocks, pairsPerBlock, stream) : launchSwBlocks2T<RR, false>(ctx, sa, nBlocks, pairsPerBlock, stream);
    switch (R) {
        FS_SW_CASE(1) FS_SW_CASE(2) FS_SW_CASE(3) FS_SW_CASE(4) FS_SW_CASE(6) FS_SW_CASE(8)
        default: ctx->err = "internal: bad SW R"; return FSGPU_E_ARG;
    }
#undef FS_SW_CASE
}


// Builds the per-tile LDS images (host) and runs all row tiles of one pass.
//   packed:  value = (fwd int16) | (rev int16) << 16;   int32: value = the selected direction's score
static int runSwPass(fsgpu_ctx *ctx, bool packed, const int16_t *pAA0, const int16_t *p3_0, const int16_t *pAA1, const int16_t *p3_1,
                     int L, const uint32_t *dTids, int nPairs, int maxLt, int go, int ge, int32_t *dRes0, int32_t *dRes1) {
    const bool hasAA = pAA0 != nullptr;
    const int nTiles = L <= 64 * kSwMaxR ? 1 : (L + 64 * kSwMaxR - 1) / (64 * kSwMaxR);
    const int R = nTiles == 1 ? swPickR(L) : kSwMaxR;
    const int rowDw = swRowDwords(R);
    const size_t tblDw = (size_t) kAlphabet * rowDw;
    const size_t imgDw = tblDw * (hasAA ? 2 : 1);
    int rc;
    if ((rc = ensurePinned(ctx, ctx->hImg, imgDw * nTiles * 4)) != FSGPU_OK) return rc;
    uint32_t *img = (uint32_t *) ctx->hImg.p;
    memset(img, 0, imgDw * nTiles * 4);
    for (int t = 0; t < nTiles; t++) {
        const int base = t * 64 * R;
        for (int tbl = 0; tbl < (hasAA ? 2 : 1); tbl++) {
            const int16_t *f = tbl == 0 ? p3_0 : pAA0;
            const int16_t *r = tbl == 0 ? p3_1 : pAA1;
            uint32_t *dst = img + imgDw * t + tblDw * tbl;
            for (int a = 0; a < kAlphabet; a++)
                for (int lane = 0; lane < 64; lane++)
                    for (int rr = 0; rr < R; rr++) {
                        const int q = base + lane * R + rr;
                        uint32_t v = 0;
                        if (q < L) {
                            if (packed) v = (uint32_t) (uint16_t) f[(size_t) a * L + q] | ((uint32_t) (uint16_t) r[(size_t) a * L + q] << 16);
                            else v = (uint32_t) (int32_t) f[(size_t) a * L + q];
                        }
                        dst[(size_t) a * rowDw + swDwordIndex(R, lane, rr)] = v;
                    }
        }
    }
    if ((rc = ensure(ctx, ctx->img, imgDw * nTiles * 4)) != FSGPU_OK) return rc;
    HIPCHK(hipMemcpyAsync(ctx->img.p, img, imgDw * nTiles * 4, hipMemcpyHostToDevice, ctx->stream));
    const uint32_t stride = (uint32_t) ((maxLt + 63) / 64 * 64);
    if (nTiles > 1) {
        const size_t bbytes = (size_t) nPairs * stride * 3 * 4;
        if ((rc = ensure(ctx, ctx->border0, bbytes)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->border1, bbytes)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->keys, (size_t) nPairs * 2 * 8)) != FSGPU_OK) return rc;
    }
    for (int t = 0; t < nTiles; t++) {
        SwArgs sa;
        if (ctx->sw.explicitTargets) {
            sa.aa = (const uint8_t *) ctx->ovAA.p; sa.ss = (const uint8_t *) ctx->ovSS.p; sa.offsets = (const uint64_t *) ctx->ovOff.p; sa.lengths = (const int32_t *) ctx->ovLen.p;
        } else {
            sa.aa = ctx->db->alnAA; sa.ss = ctx->db->aln3di; sa.offsets = ctx->db->dOffsets; sa.lengths = ctx->db->dLengths;
        }
        sa.targetIds = dTids; sa.nPairs = nPairs;
        sa.profSS = (const uint32_t *) ctx->img.p + imgDw * t;
        sa.profAA = hasAA ? sa.profSS + tblDw : nullptr;
        sa.tileBase = t * 64 * R;
        sa.rowsInTile = std::min(64 * R, L - sa.tileBase);
        sa.segLen = packed ? (L + 15) / 16 : (L + 7) / 8;
        sa.go = packed ? ((uint32_t) go | ((uint32_t) go << 16)) : (uint32_t) go;
        sa.ge = packed ? ((uint32_t) ge | ((uint32_t) ge << 16)) : (uint32_t) ge;
        sa.tileIn = t > 0; sa.tileOut = t + 1 < nTiles;
        sa.borderIn = (const uint32_t *) ((t & 1) ? ctx->border1.p : ctx->border0.p);
        sa.borderOut = (uint32_t *) ((t & 1) ? ctx->border0.p : ctx->border1.p);
        sa.borderStride = stride;
        sa.keys = (uint64_t *) ctx->keys.p;
        sa.res0 = dRes0; sa.res1 = dRes1;
        sa.blocks = nullptr;
        rc = packed ? launchSw<Pk16>(ctx, R, hasAA, sa, nPairs) : launchSw<I32>(ctx, R, hasAA, sa, nPairs);
        if (rc != FSGPU_OK) return rc;
    }
    return FSGPU_OK;
}

// ---- multi-query row-tiled SW: the queries of a fsgpu_sw_multi_dir call that are longer than 64 * kSwMaxR rows ----
// One k_sw launch per tile LEVEL serves the tiles of that level of ALL such queries (workgroup = up to 8 pairs of one query,
// SwTileBlock); levels follow each other on a side stream (ctx->swLong) next to the single-tile launches of the same call.
// k_sw carries the forward and the reversed query in the int16 halves, so the forward call already has the reversed-query
// results: they are kept per query (with a hash of what they depend on) and handed out by the following reversed call.
struct SwLongPlan {
    std::vector<uint32_t> slotQ, slotJ;         // launch slot -> query, index into its targetIds
    size_t n = 0;
};

static uint64_t hashWords(uint64_t h, const void *p, size_t bytes) {
    const unsigned char *c = (const unsigned char *) p;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) { uint64_t w; memcpy(&w, c + i, 8); h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
    for (; i < bytes; i++) h = (h ^ c[i]) * 0x100000001B3ull;
    return h;
}
static uint64_t swLongHash(const fsgpu_sw_query &q) {
    uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t) q.L ^ ((uint64_t) q.n << 32);
    h = hashWords(h, q.targetIds, (size_t) q.n * 4);
    h = hashWords(h, q.p3Di_rev, (size_t) q.L * kAlphabet * 2);
    if (q.pAA_rev) h = hashWords(h, q.pAA_rev, (size_t) q.L * kAlphabet * 2);
    return h ? h : 1;
}

template <bool HAS_AA>
static int launchSwTilesT(fsgpu_ctx *ctx, const SwArgs &sa, int nBlocks, hipStream_t stream) {
    const int lds = (HAS_AA ? 2 : 1) * kAlphabet * swRowDwords(kSwMaxR) * 4;
    static thread_local uint64_t attrDevs = 0;
    const uint64_t devBit = 1ull << (ctx->device & 63);
    if (!(attrDevs & devBit)) {
        HIPCHK(hipFuncSetAttribute((const void *) k_sw<kSwMaxR, HAS_AA, Pk16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attrDevs |= devBit;
    }
    hipLaunchKernelGGL((k_sw<kSwMaxR, HAS_AA, Pk16>), dim3(nBlocks), dim3(512), lds, stream, sa);
    HIPCHK(hipGetLastError());
    return FSGPU_OK;
}

// isLong[i]: query i is row-tiled and has selected pairs.  Pairs answered from the forward call's results go straight to out[].
static int swLongEnqueue(fsgpu_ctx *ctx, const fsgpu_sw_query *q, int nq, const std::vector<char> &isLong, const int32_t *const *sel, const int32_t *nsel,
                         const std::vector<size_t> &base, bool hasAA, int gapOpen, int gapExtend, int dir, fsgpu_swres *out, SwLongPlan &plan) {
    plan.n = 0; plan.slotQ.clear(); plan.slotJ.clear();
    if (dir == 0) ctx->swLongRev.assign((size_t) nq, fsgpu_ctx::LongRev());
    const std::vector<int32_t> &len = ctx->db->hLengths;
    std::vector<size_t> qFirst(nq + 1, 0);                 // slots of query i: [qFirst[i], qFirst[i + 1])
    std::vector<uint64_t> lkey;
    for (int i = 0; i < nq; i++) {
        qFirst[i] = plan.n;
        if (!isLong[i]) continue;
        const int ns = sel ? nsel[i] : q[i].n;
        const fsgpu_ctx::LongRev *have = nullptr;
        if (dir == 1 && (size_t) i < ctx->swLongRev.size() && ctx->swLongRev[i].hash != 0 && ctx->swLongRev[i].res.size() == (size_t) q[i].n * 4 &&
            ctx->swLongRev[i].hash == swLongHash(q[i]))
            have = &ctx->swLongRev[i];
        lkey.clear();
        for (int k = 0; k < ns; k++) {
            const int j = sel ? sel[i][k] : k;
            if (have && have->res[(size_t) j * 4 + 3] != 0) { memcpy(&out[base[i] + j], &have->res[(size_t) j * 4], 16); continue; }
            lkey.push_back(((uint64_t) (0xFFFFFF - len[q[i].targetIds[j]]) << 32) | (uint32_t) j);
        }
        std::sort(lkey.begin(), lkey.end());               // longest target first: the waves of a workgroup are of similar length
        for (uint64_t k : lkey) { plan.slotQ.push_back((uint32_t) i); plan.slotJ.push_back((uint32_t) k); }
        plan.n += lkey.size();
        if (dir == 0) { ctx->swLongRev[i].hash = swLongHash(q[i]); ctx->swLongRev[i].res.assign((size_t) q[i].n * 4, 0); }
    }
    qFirst[nq] = plan.n;
    if (plan.n == 0) return FSGPU_OK;
    const size_t n = plan.n;
    const int R = kSwMaxR, rowsPerTile = 64 * R, rowDw = swRowDwords(R);
    const size_t tblDw = (size_t) kAlphabet * rowDw, imgDw = tblDw * (hasAA ? 2 : 1);
    // geometry: tiles per query, image offsets, blocks per level
    int maxTiles = 0;
    std::vector<int> nTiles(nq, 0);
    std::vector<size_t> imgFirst(nq, 0);                   // dword offset of tile 0 of query i inside the image section
    size_t imgTotalDw = 0;
    for (int i = 0; i < nq; i++) {
        if (qFirst[i + 1] == qFirst[i]) continue;
        nTiles[i] = (q[i].L + rowsPerTile - 1) / rowsPerTile;
        maxTiles = std::max(maxTiles, nTiles[i]);
        imgFirst[i] = imgTotalDw;
        imgTotalDw += imgDw * (size_t) nTiles[i];
    }
    if (imgTotalDw >= (1ull << 32)) { ctx->err = "fsgpu_sw_multi_dir: tile images of one call exceed 16 GiB"; return FSGPU_E_NOMEM; }
    std::vector<size_t> levelFirst(maxTiles + 1, 0);
    for (int t = 0; t < maxTiles; t++) {
        size_t nb = 0;
        for (int i = 0; i < nq; i++) if (nTiles[i] > t) nb += (qFirst[i + 1] - qFirst[i] + 7) / 8;
        levelFirst[t + 1] = levelFirst[t] + nb;
    }
    auto align16 = [](size_t x) { return (x + 15) / 16 * 16; };
    const size_t offTids = 0, offBase = align16(n * 4), offBlocks = align16(offBase + n * 4), offImg = align16(offBlocks + levelFirst[maxTiles] * sizeof(SwTileBlock));
    const size_t bytes = offImg + imgTotalDw * 4;
    int rc;
    if ((rc = ensurePinned(ctx, ctx->hLbuf, bytes)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->lbuf, bytes)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->lres, n * 32)) != FSGPU_OK) return rc;
    if ((rc = ensurePinned(ctx, ctx->hLres, n * 32)) != FSGPU_OK) return rc;
    unsigned char *h = (unsigned char *) ctx->hLbuf.p;
    uint32_t *hT = (uint32_t *) (h + offTids), *hB = (uint32_t *) (h + offBase);
    SwTileBlock *hBlk = (SwTileBlock *) (h + offBlocks);
    uint32_t *hImg = (uint32_t *) (h + offImg);
    uint64_t cols = 0;
    for (size_t s2 = 0; s2 < n; s2++) {
        const uint32_t tid = q[plan.slotQ[s2]].targetIds[plan.slotJ[s2]];
        hT[s2] = tid;
        if (cols >= (1ull << 32)) { ctx->err = "fsgpu_sw_multi_dir: tile borders of one call exceed 2^32 columns"; return FSGPU_E_NOMEM; }
        hB[s2] = (uint32_t) cols;
        cols += (uint64_t) ((len[tid] + 63) / 64 * 64);
    }
    for (int t = 0; t < maxTiles; t++) {
        SwTileBlock *b = hBlk + levelFirst[t];
        size_t nb = 0;
        for (int i = 0; i < nq; i++) {
            if (nTiles[i] <= t) continue;
            const int L = q[i].L;
            for (size_t p0 = qFirst[i]; p0 < qFirst[i + 1]; p0 += 8) {
                SwTileBlock &d = b[nb++];
                d.imgOff = (uint32_t) (imgFirst[i] + imgDw * (size_t) t); d.firstPair = (uint32_t) p0; d.nPairs = (uint16_t) std::min<size_t>(8, qFirst[i + 1] - p0);
                d.rowsInTile = (uint16_t) std::min(rowsPerTile, L - t * rowsPerTile); d.segLen = (uint32_t) ((L + 15) / 16);
                d.tileBase = (uint32_t) (t * rowsPerTile); d.flags = (t > 0 ? 1u : 0u) | (t + 1 < nTiles[i] ? 2u : 0u);
            }
        }
        std::stable_sort(b, b + nb, [&](const SwTileBlock &x, const SwTileBlock &y) { return len[hT[x.firstPair]] > len[hT[y.firstPair]]; });
    }
    for (int i = 0; i < nq; i++) {
        if (nTiles[i] == 0) continue;
        const int L = q[i].L;
        for (int t = 0; t < nTiles[i]; t++) {
            const int rowBase = t * rowsPerTile;
            for (int tbl = 0; tbl < (hasAA ? 2 : 1); tbl++) {
                const int16_t *f = tbl == 0 ? q[i].p3Di_fwd : q[i].pAA_fwd;
                const int16_t *r = tbl == 0 ? q[i].p3Di_rev : q[i].pAA_rev;
                uint32_t *dst = hImg + imgFirst[i] + imgDw * (size_t) t + tblDw * tbl;
                for (int a = 0; a < kAlphabet; a++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int rr = 0; rr < R; rr++) {
                            const int row = rowBase + lane * R + rr;
                            uint32_t v = 0;
                            if (row < L) v = (uint32_t) (uint16_t) f[(size_t) a * L + row] | ((uint32_t) (uint16_t) r[(size_t) a * L + row] << 16);
                            dst[(size_t) a * rowDw + swDwordIndex(R, lane, rr)] = v;
                        }
            }
        }
    }
    if (maxTiles > 1) {
        if ((rc = ensure(ctx, ctx->border0, cols * 12)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->border1, cols * 12)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->keys, n * 2 * 8)) != FSGPU_OK) return rc;
    }
    if (!ctx->swLong) HIPCHK(hipStreamCreateWithFlags(&ctx->swLong, hipStreamNonBlocking));
    HIPCHK(hipMemcpyAsync(ctx->lbuf.p, h, bytes, hipMemcpyHostToDevice, ctx->swLong));
    const unsigned char *d = (const unsigned char *) ctx->lbuf.p;
    for (int t = 0; t < maxTiles; t++) {
        SwArgs sa;
        sa.aa = ctx->db->alnAA; sa.ss = ctx->db->aln3di; sa.offsets = ctx->db->dOffsets; sa.lengths = ctx->db->dLengths;
        sa.targetIds = (const uint32_t *) (d + offTids); sa.nPairs = (int) n;
        sa.profSS = (const uint32_t *) (d + offImg); sa.profAA = nullptr;
        sa.tileBase = 0; sa.rowsInTile = 0; sa.segLen = 1;
        sa.go = (uint32_t) gapOpen | ((uint32_t) gapOpen << 16);
        sa.ge = (uint32_t) gapExtend | ((uint32_t) gapExtend << 16);
        sa.tileIn = 0; sa.tileOut = 0;
        sa.borderIn = (const uint32_t *) ((t & 1) ? ctx->border1.p : ctx->border0.p);
        sa.borderOut = (uint32_t *) ((t & 1) ? ctx->border0.p : ctx->border1.p);
        sa.borderStride = 0;
        sa.keys = (uint64_t *) ctx->keys.p;
        sa.res0 = (int32_t *) ctx->lres.p; sa.res1 = (int32_t *) ctx->lres.p + n * 4;
        sa.blocks = nullptr; sa.dir = 0;
        sa.tblocks = (const SwTileBlock *) (d + offBlocks) + levelFirst[t];
        sa.borderBase = (const uint32_t *) (d + offBase);
        const int nb = (int) (levelFirst[t + 1] - levelFirst[t]);
        rc = hasAA ? launchSwTilesT<true>(ctx, sa, nb, ctx->swLong) : launchSwTilesT<false>(ctx, sa, nb, ctx->swLong);
        if (rc != FSGPU_OK) return rc;
    }
    HIPCHK(hipMemcpyAsync(ctx->hLres.p, ctx->lres.p, n * 32, hipMemcpyDeviceToHost, ctx->swLong));
    return FSGPU_OK;
}

static int swLongCollect(fsgpu_ctx *ctx, const SwLongPlan &plan, const std::vector<size_t> &base, int dir, fsgpu_swres *out) {
    if (plan.n == 0) return FSGPU_OK;
    int rc = syncStreamOf(ctx, ctx->swLong);
    if (rc != FSGPU_OK) return rc;
    const int32_t *fwd = (const int32_t *) ctx->hLres.p, *rev = fwd + plan.n * 4;
    for (size_t s2 = 0; s2 < plan.n; s2++) {
        const uint32_t i = plan.slotQ[s2], j = plan.slotJ[s2];
        memcpy(&out[base[i] + j], (dir == 0 ? fwd : rev) + s2 * 4, 16);
        if (dir == 0) memcpy(&ctx->swLongRev[i].res[(size_t) j * 4], rev + s2 * 4, 16);
    }
    return FSGPU_OK;
}

extern "C" {

static int swLaunchImpl(fsgpu_ctx *ctx, const int16_t *pAA_fwd, const int16_t *p3Di_fwd, const int16_t *pAA_rev,
                        const int16_t *p3Di_rev, int L, const uint32_t *targetIds, int n, int gapOpen, int gapExtend, bool explicitTargets);

int fsgpu_sw_launch(fsgpu_ctx *ctx, const int16_t *pAA_fwd, const int16_t *p3Di_fwd, const int16_t *pAA_rev,
                    const int16_t *p3Di_rev, int L, const uint32_t *targetIds, int n, int gapOpen, int gapExtend) {
    return swLaunchImpl(ctx, pAA_fwd, p3Di_fwd, pAA_rev, p3Di_rev, L, targetIds, n, gapOpen, gapExtend, false);
}

// Explicit target sequences (structurealign --alt-ali re-aligns a target whose previous alignment range was overwritten with X,
// F/src/strucclustutils/structurealign.cpp:115-138): the sequences are staged in per-context device buffers and the same
// kernels run on them with ids 0..n-1.
int fsgpu_sw_batch_seqs(fsgpu_ctx *ctx, const int16_t *pAA_fwd, const int16_t *p3Di_fwd, const int16_t *pAA_rev, const int16_t *p3Di_rev,
                        int L, const uint8_t *tAA, const uint8_t *t3Di, const uint64_t *offsets, const int32_t *lengths, int n,
                        int gapOpen, int gapExtend, fsgpu_swres *fwd, fsgpu_swres *rev) {
    if (!ctx || !t3Di || !offsets || !lengths || n < 0 || !fwd || !rev || (pAA_fwd && !tAA)) { if (ctx) ctx->err = "fsgpu_sw_batch_seqs: bad argument"; return FSGPU_E_ARG; }
    if (ctx->sw.pending) { ctx->err = "previous SW batch not finished"; return FSGPU_E_ARG; }
    if (n == 0) return FSGPU_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t bytes = offsets[n];
    int rc;
    if ((rc = ensure(ctx, ctx->ovSS, bytes + 16)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ovAA, bytes + 16)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ovOff, (size_t) (n + 1) * 8)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ovLen, (size_t) n * 4)) != FSGPU_OK) return rc;
    ctx->sw.ovLengths.assign(lengths, lengths + n);
    for (int i = 0; i < n; i++)
        if (lengths[i] <= 0 || lengths[i] > FSGPU_MAX_SEQ_LEN || offsets[i] + (uint64_t) lengths[i] > bytes) { ctx->err = "fsgpu_sw_batch_seqs: bad target layout"; return FSGPU_E_ARG; }
    // pageable sources: synchronous copies (this path serves a handful of pairs per query)
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(ctx->ovSS.p, t3Di, bytes, hipMemcpyHostToDevice));
    if (tAA) HIPCHK(hipMemcpy(ctx->ovAA.p, tAA, bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->ovOff.p, offsets, (size_t) (n + 1) * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->ovLen.p, lengths, (size_t) n * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> ids(n);
    for (int i = 0; i < n; i++) ids[i] = (uint32_t) i;
    rc = swLaunchImpl(ctx, pAA_fwd, p3Di_fwd, pAA_rev, p3Di_rev, L, ids.data(), n, gapOpen, gapExtend, true);
    if (rc != FSGPU_OK) { ctx->sw.explicitTargets = false; return rc; }
    rc = fsgpu_sw_finish(ctx, fwd, rev);
    ctx->sw.explicitTargets = false;
    return rc;
}

static int swLaunchImpl(fsgpu_ctx *ctx, const int16_t *pAA_fwd, const int16_t *p3Di_fwd, const int16_t *pAA_rev,
                        const int16_t *p3Di_rev, int L, const uint32_t *targetIds, int n, int gapOpen, int gapExtend, bool explicitTargets) {
    if (!ctx) return FSGPU_E_ARG;
    if (!p3Di_fwd || !p3Di_rev || L <= 0 || L > FSGPU_MAX_SEQ_LEN || n < 0 || (n > 0 && !targetIds) || ((pAA_fwd == nullptr) != (pAA_rev == nullptr))) {
        ctx->err = "fsgpu_sw_launch: bad argument"; return FSGPU_E_ARG;
    }
    if (!ctx->db || ctx->db->n == 0) { ctx->err = "no database loaded"; return FSGPU_E_NODB; }
    if (pAA_fwd && !explicitTargets && !ctx->db->hasAA) { ctx->err = "AA profiles given but the database was loaded without AA sequences"; return FSGPU_E_NODB; }
    if (!(gapOpen > gapExtend && gapExtend >= 0 && gapOpen < 32768)) {
        ctx->err = "device SW requires gapOpen > gapExtend >= 0 (the striped reference kernel's lazy-F shortcut is only reproduced for that case)";
        return FSGPU_E_UNSUPPORTED;
    }
    if (ctx->sw.pending) { ctx->err = "previous SW batch not finished"; return FSGPU_E_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    ctx->sw.explicitTargets = explicitTargets;
    const std::vector<int32_t> &hLen = explicitTargets ? ctx->sw.ovLengths : ctx->db->hLengths;
    const uint64_t nTargets = explicitTargets ? ctx->sw.ovLengths.size() : ctx->db->n;
    ctx->sw.n = n; ctx->sw.L = L; ctx->sw.go = gapOpen; ctx->sw.ge = gapExtend; ctx->sw.hasAA = pAA_fwd != nullptr;
    ctx->sw.pAAf = pAA_fwd; ctx->sw.p3f = p3Di_fwd; ctx->sw.pAAr = pAA_rev; ctx->sw.p3r = p3Di_rev;
    ctx->sw.tids.assign(targetIds, targetIds + n);
    if (n == 0) { ctx->sw.pending = true; return FSGPU_OK; }
    int maxLt = 1;
    for (int i = 0; i < n; i++) {
        if (targetIds[i] >= nTargets) { ctx->err = "target id out of range"; return FSGPU_E_ARG; }
        maxLt = std::max(maxLt, hLen[targetIds[i]]);
    }
    int rc;
    if ((rc = ensure(ctx, ctx->tids, (size_t) n * 4)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->res0, (size_t) n * 16)) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->res1, (size_t) n * 16)) != FSGPU_OK) return rc;
    if ((rc = ensurePinned(ctx, ctx->hRes0, (size_t) n * 16)) != FSGPU_OK) return rc;
    if ((rc = ensurePinned(ctx, ctx->hRes1, (size_t) n * 16)) != FSGPU_OK) return rc;
    if ((rc = ensurePinned(ctx, ctx->hTids, (size_t) n * 4)) != FSGPU_OK) return rc;
    memcpy(ctx->hTids.p, ctx->sw.tids.data(), (size_t) n * 4);
    HIPCHK(hipMemcpyAsync(ctx->tids.p, ctx->hTids.p, (size_t) n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
    rc = runSwPass(ctx, true, pAA_fwd, p3Di_fwd, pAA_rev, p3Di_rev, L, (const uint32_t *) ctx->tids.p, n, maxLt, gapOpen, gapExtend,
                   (int32_t *) ctx->res0.p, (int32_t *) ctx->res1.p);
    if (rc != FSGPU_OK) return rc;
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->evValid[1] = true;
    HIPCHK(hipMemcpyAsync(ctx->hRes0.p, ctx->res0.p, (size_t) n * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->hRes1.p, ctx->res1.p, (size_t) n * 16, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sw.pending = true;          // only now: an error above leaves the context free for the next launch
    return FSGPU_OK;
}

int fsgpu_sw_finish(fsgpu_ctx *ctx, fsgpu_swres *fwd, fsgpu_swres *rev) {
    if (!ctx || !fwd || !rev) return FSGPU_E_ARG;
    if (!ctx->sw.pending) { ctx->err = "no SW batch in flight"; return FSGPU_E_ARG; }
    ctx->sw.pending = false;
    HIPCHK(hipSetDevice(ctx->device));        // the int32 re-run below launches kernels: the calling thread may have another device current
    const int n = ctx->sw.n;
    if (n == 0) return FSGPU_OK;
    { int rc = syncStream(ctx); if (rc != FSGPU_OK) return rc; }
    memcpy(fwd, ctx->hRes0.p, (size_t) n * 16);
    memcpy(rev, ctx->hRes1.p, (size_t) n * 16);
    // int16 saturation -> int32 re-run with the int32 kernel's segment length (alignScoreEndPos, :313-336)
    for (int dir = 0; dir < 2; dir++) {
        fsgpu_swres *res = dir == 0 ? fwd : rev;
        std::vector<uint32_t> ids;
        std::vector<int> where;
        int maxLt = 1;
        for (int i = 0; i < n; i++)
            if (res[i].score == 32767) {
                ids.push_back(ctx->sw.tids[i]); where.push_back(i);
                maxLt = std::max(maxLt, (ctx->sw.explicitTargets ? ctx->sw.ovLengths : ctx->db->hLengths)[ctx->sw.tids[i]]);
            }
        if (ids.empty()) continue;
        const int m = (int) ids.size();
        memcpy(ctx->hTids.p, ids.data(), (size_t) m * 4);         // pinned staging (sized for n >= m at launch), ordered on the context stream
        HIPCHK(hipMemcpyAsync(ctx->tids.p, ctx->hTids.p, (size_t) m * 4, hipMemcpyHostToDevice, ctx->stream));
        const int16_t *pA = dir == 0 ? ctx->sw.pAAf : ctx->sw.pAAr;
        const int16_t *p3 = dir == 0 ? ctx->sw.p3f : ctx->sw.p3r;
        int rc = runSwPass(ctx, false, pA, p3, nullptr, nullptr, ctx->sw.L, (const uint32_t *) ctx->tids.p, m, maxLt, ctx->sw.go, ctx->sw.ge,
                           (int32_t *) ctx->res0.p, nullptr);
        if (rc != FSGPU_OK) return rc;
        HIPCHK(hipMemcpyAsync(ctx->hRes0.p, ctx->res0.p, (size_t) m * 16, hipMemcpyDeviceToHost, ctx->stream));
        { int rc2 = syncStream(ctx); if (rc2 != FSGPU_OK) return rc2; }
        for (int k = 0; k < m; k++) memcpy(&res[where[k]], (const int32_t *) ctx->hRes0.p + (size_t) k * 4, 16);
    }
    return FSGPU_OK;
}

int fsgpu_sw_batch(fsgpu_ctx *ctx, const int16_t *pAA_fwd, const int16_t *p3Di_fwd, const int16_t *pAA_rev, const int16_t *p3Di_rev,
                   int L, const uint32_t *targetIds, int n, int gapOpen, int gapExtend, fsgpu_swres *fwd, fsgpu_swres *rev) {
    int rc = fsgpu_sw_launch(ctx, pAA_fwd, p3Di_fwd, pAA_rev, p3Di_rev, L, targetIds, n, gapOpen, gapExtend);
    if (rc != FSGPU_OK) return rc;
    return fsgpu_sw_finish(ctx, fwd, rev);
}

// Several queries in one go: all single-tile queries (L <= 512) of one register class R share ONE launch -- workgroups
// of 4 waves, each workgroup serving pairs of a single query and loading that query's LDS image -- so the device sees
// tens of thousands of independent waves instead of ~1000 per launch and the long-target tail of one query overlaps
// the bulk of the others.  One call runs ONE direction (dir 0: forward query, 1: reversed query) over the selected
// pairs with k_sw2 (two targets per wave): structurealign looks at the reversed-query score only for pairs that pass
// the forward gates, so the caller runs dir 0 over everything, gates, and runs dir 1 over the survivors.
// Longer queries run as multi-query row-tiled k_sw launches (swLongEnqueue); int16-saturated pairs go through the single-query path.
int fsgpu_sw_multi_dir(fsgpu_ctx *ctx, const fsgpu_sw_query *q, int nq, int gapOpen, int gapExtend, int dir,
                       const int32_t *const *sel, const int32_t *nsel, fsgpu_swres *out) {
    if (!ctx || nq < 0 || (nq > 0 && (!q || !out)) || (dir != 0 && dir != 1) || ((sel == nullptr) != (nsel == nullptr))) return FSGPU_E_ARG;
    if (!ctx->db || ctx->db->n == 0) { ctx->err = "no database loaded"; return FSGPU_E_NODB; }
    if (!(gapOpen > gapExtend && gapExtend >= 0 && gapOpen < 32768)) {
        ctx->err = "device SW requires gapOpen > gapExtend >= 0 (the striped reference kernel's lazy-F shortcut is only reproduced for that case)";
        return FSGPU_E_UNSUPPORTED;
    }
    if (ctx->sw.pending) { ctx->err = "previous SW batch not finished"; return FSGPU_E_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<size_t> base(nq + 1, 0), sbase(nq + 1, 0);     // offsets into out[] (all pairs) / into the launch (selected pairs)
    bool hasAA = false, anyAA = false, allAA = true;
    for (int i = 0; i < nq; i++) {
        if (!q[i].p3Di_fwd || !q[i].p3Di_rev || q[i].L <= 0 || q[i].L > FSGPU_MAX_SEQ_LEN || q[i].n < 0 || (q[i].n > 0 && !q[i].targetIds) ||
            ((q[i].pAA_fwd == nullptr) != (q[i].pAA_rev == nullptr))) { ctx->err = "fsgpu_sw_multi: bad query"; return FSGPU_E_ARG; }
        anyAA = anyAA || q[i].pAA_fwd != nullptr; allAA = allAA && q[i].pAA_fwd != nullptr;
        base[i + 1] = base[i] + (size_t) q[i].n;
        const int ns = sel ? nsel[i] : q[i].n;
        if (ns < 0 || ns > q[i].n || (sel && ns > 0 && !sel[i])) { ctx->err = "fsgpu_sw_multi_dir: bad selection"; return FSGPU_E_ARG; }
        sbase[i + 1] = sbase[i] + (size_t) ns;
        for (int k = 0; k < q[i].n; k++) if (q[i].targetIds[k] >= ctx->db->n) { ctx->err = "target id out of range"; return FSGPU_E_ARG; }
        if (sel) for (int k = 0; k < ns; k++) if (sel[i][k] < 0 || sel[i][k] >= q[i].n) { ctx->err = "fsgpu_sw_multi_dir: selection index out of range"; return FSGPU_E_ARG; }
    }
    if (anyAA != allAA) { ctx->err = "fsgpu_sw_multi: either all or none of the queries carry AA profiles"; return FSGPU_E_ARG; }
    hasAA = anyAA;
    if (hasAA && !ctx->db->hasAA) { ctx->err = "AA profiles given but the database was loaded without AA sequences"; return FSGPU_E_NODB; }
    const size_t total = sbase[nq];
    auto nSel = [&](int i) { return (int) (sbase[i + 1] - sbase[i]); };
    auto selIdx = [&](int i, int k) { return sel ? sel[i][k] : k; };
    int rc;
    // row-tiled queries: own launches on a side stream, next to the single-tile launches below
    SwLongPlan longPlan;
    {
        std::vector<char> isLong(nq, 0);
        bool any = false;
        for (int i = 0; i < nq; i++) if (q[i].L > 64 * kSwMaxR && nSel(i) > 0) { isLong[i] = 1; any = true; }
        if (any && (rc = swLongEnqueue(ctx, q, nq, isLong, sel, nsel, base, hasAA, gapOpen, gapExtend, dir, out, longPlan)) != FSGPU_OK) {
            if (ctx->swLong) (void) hipStreamSynchronize(ctx->swLong);
            return rc;
        }
    }
    // whatever goes wrong below: the side stream must be idle before its buffers are reused
    struct LongGuard { fsgpu_ctx *c; bool armed; ~LongGuard() { if (armed && c->swLong) (void) hipStreamSynchronize(c->swLong); } } longGuard{ctx, longPlan.n > 0};
    std::vector<uint32_t> perm;       // launch slot -> index into q[i].targetIds
    std::vector<uint64_t> lkey;
    // ---- launch groups by register class ----
    const int classes[6] = {1, 2, 3, 4, 6, 8};
    std::vector<int> cls(nq, -1);
    for (int i = 0; i < nq; i++) if (q[i].L <= 64 * kSwMaxR && nSel(i) > 0) cls[i] = swPickR(q[i].L);
    size_t imgDwTotal = 0, nBlocks = 0;
    const int ppb = sw2PairsPerBlock();
    for (int i = 0; i < nq; i++) if (cls[i] > 0) { imgDwTotal += (size_t) kSw2Rows * swRowDwords(cls[i]) * (hasAA ? 2 : 1); nBlocks += ((size_t) nSel(i) + ppb - 1) / ppb; }
    if (total) {
        if ((rc = ensure(ctx, ctx->tids, total * 4)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->res0, total * 16)) != FSGPU_OK) return rc;
        if ((rc = ensurePinned(ctx, ctx->hRes0, total * 16)) != FSGPU_OK) return rc;
        if ((rc = ensurePinned(ctx, ctx->hTids, total * 4)) != FSGPU_OK) return rc;
        // inside every query the pairs are issued longest target first (perm) -- neighbours share a wave, so they should
        // be of similar length --, and the workgroups of a launch are ordered by their longest target (LPT)
        perm.resize(total);
        for (int i = 0; i < nq; i++) {
            const int ns = nSel(i);
            uint32_t *p = perm.data() + sbase[i];
            const uint32_t *ids = q[i].targetIds;
            const std::vector<int32_t> &len = ctx->db->hLengths;
            lkey.resize(ns);
            for (int k = 0; k < ns; k++) { const int j = selIdx(i, k); lkey[k] = ((uint64_t) (0xFFFFFF - len[ids[j]]) << 32) | (uint32_t) j; }
            std::sort(lkey.begin(), lkey.end());
            uint32_t *dst = (uint32_t *) ctx->hTids.p + sbase[i];
            for (int k = 0; k < ns; k++) { p[k] = (uint32_t) lkey[k]; dst[k] = ids[p[k]]; }
        }
        HIPCHK(hipMemcpyAsync(ctx->tids.p, ctx->hTids.p, total * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    if (dir == 0 || !ctx->evValid[1]) HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));     // a forward + reversed pass pair is timed as one
    if (!ctx->swDirEv[3]) for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&ctx->swDirEv[i]));
    ctx->swDirValid[dir] = false;
    if (dir == 0) ctx->swDirValid[1] = false;
    ctx->swDirExtraMs[dir] = 0;
    {
        // work of this pass in the units of the kernel's roofline: DP cells (query rows x target columns of every single-tile pair) and
        // VALU wave-instructions (a wave carries two targets of one query and runs max(LtA, LtB) + lanes - 1 steps; a step is 14 packed
        // instructions per register row + 14 around them -- lane shifts, LDS addresses, column and maximum bookkeeping; counted in the ISA
        // of k_sw2<6, false>: 98 VALU instructions per step outside the new-maximum block -- and 2 per row + 9 more with the AA table: 119)
        double cells = 0, pairs = 0, wsteps = 0;
        const std::vector<int32_t> &len = ctx->db->hLengths;
        for (int i = 0; i < nq; i++) {
            if (cls[i] <= 0) continue;
            const int ns = nSel(i), R = cls[i], lanes = (q[i].L + R - 1) / R;
            const uint32_t *p = perm.data() + sbase[i];
            for (int k = 0; k < ns; k++) {
                const int lt = len[q[i].targetIds[p[k]]];
                cells += (double) q[i].L * lt;
                if ((k & 1) == 0 && lt > 0) wsteps += (double) (lt + lanes - 1) * (14.0 * R + 14.0 + (hasAA ? 2.0 * R + 9.0 : 0.0));   // pairs are longest first: the even one sets the wave's length
            }
            pairs += ns;
        }
        ctx->swDirCells[dir] = cells; ctx->swDirPairs[dir] = pairs; ctx->swDirWaveSteps[dir] = wsteps;
    }
    HIPCHK(hipEventRecord(ctx->swDirEv[2 * dir], ctx->stream));
    
    if (nBlocks) {
        if ((rc = ensurePinned(ctx, ctx->hImg, imgDwTotal * 4 + nBlocks * sizeof(SwBlockDesc) + 64)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->img, imgDwTotal * 4 + nBlocks * sizeof(SwBlockDesc) + 64)) != FSGPU_OK) return rc;
        uint32_t *img = (uint32_t *) ctx->hImg.p;
        SwBlockDesc *hb = (SwBlockDesc *) ((unsigned char *) ctx->hImg.p + ((imgDwTotal * 4 + 15) / 16) * 16);
        const size_t descOff = ((imgDwTotal * 4 + 15) / 16) * 16;
        size_t imgPos = 0, blkPos = 0;
        struct Group { int R; size_t blk0, nblk; };
        std::vector<Group> groups;
        for (int R : classes) {
            Group g{R, blkPos, 0};
            const int rowDw = swRowDwords(R);
            const size_t tblDw = (size_t) kSw2Rows * rowDw;
            for (int i = 0; i < nq; i++) {
                if (cls[i] != R) continue;
                const int L = q[i].L;
                uint32_t *dst0 = img + imgPos;
                for (int tbl = 0; tbl < (hasAA ? 2 : 1); tbl++) {
                    const int16_t *f = tbl == 0 ? q[i].p3Di_fwd : q[i].pAA_fwd;
                    const int16_t *r = tbl == 0 ? q[i].p3Di_rev : q[i].pAA_rev;
                    uint32_t *dst = dst0 + tblDw * tbl;
                    for (int a = 0; a < kAlphabet; a++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int rr = 0; rr < R; rr++) {
                                const int row = lane * R + rr;
                                uint32_t v = 0;
                                if (row < L) v = (uint32_t) (uint16_t) f[(size_t) a * L + row] | ((uint32_t) (uint16_t) r[(size_t) a * L + row] << 16);
                                dst[(size_t) a * rowDw + swDwordIndex(R, lane, rr)] = v;
                            }
                    // row 21: "past the end of this target" -- INT16_MIN in the 3Di table, 0 in the AA table (their sum must not wrap)
                    const uint32_t dead = tbl == 0 ? 0x80008000u : 0u;
                    for (int x = 0; x < rowDw; x++) dst[(size_t) kAlphabet * rowDw + x] = dead;
                }
                const int ns = nSel(i);
                for (int p0 = 0; p0 < ns; p0 += ppb) {
                    SwBlockDesc &d = hb[blkPos++];
                    d.imgOff = (uint32_t) imgPos; d.firstPair = (uint32_t) (sbase[i] + p0); d.nPairs = (uint16_t) std::min(ppb, ns - p0);
                    d.rowsInTile = (uint16_t) L; d.segLen = (uint32_t) ((L + 15) / 16);
                    g.nblk++;
                }
                imgPos += tblDw * (hasAA ? 2 : 1);
            }
            if (g.nblk) {
                // first pair of a workgroup is its longest (pairs are length-sorted inside the query)
                const uint32_t *ht = (const uint32_t *) ctx->hTids.p;
                const std::vector<int32_t> &len = ctx->db->hLengths;
                std::stable_sort(hb + g.blk0, hb + g.blk0 + g.nblk, [&](const SwBlockDesc &x, const SwBlockDesc &y) { return len[ht[x.firstPair]] > len[ht[y.firstPair]]; });
                groups.push_back(g);
            }
        }
        HIPCHK(hipMemcpyAsync(ctx->img.p, ctx->hImg.p, descOff + nBlocks * sizeof(SwBlockDesc), hipMemcpyHostToDevice, ctx->stream));
        // every register-class group gets its own stream: their long-target tails overlap instead of queueing up
        if (groups.size() > 1) {
            if (!ctx->swAuxEv[fsgpu_ctx::kSwAux]) for (int i = 0; i <= fsgpu_ctx::kSwAux; i++) HIPCHK(hipEventCreateWithFlags(&ctx->swAuxEv[i], hipEventDisableTiming));
            for (size_t gi = 1; gi < groups.size(); gi++)
                if (!ctx->swAux[gi]) HIPCHK(hipStreamCreateWithFlags(&ctx->swAux[gi], hipStreamNonBlocking));
            HIPCHK(hipEventRecord(ctx->swAuxEv[fsgpu_ctx::kSwAux], ctx->stream));          // inputs (ids, images, descriptors) are on their way
        }
        size_t gi = 0;
        for (const Group &g : groups) {
            hipStream_t gs = gi == 0 ? ctx->stream : ctx->swAux[gi];
            if (gi > 0) HIPCHK(hipStreamWaitEvent(gs, ctx->swAuxEv[fsgpu_ctx::kSwAux], 0));
            SwArgs sa;
            sa.aa = ctx->db->alnAA; sa.ss = ctx->db->aln3di; sa.offsets = ctx->db->dOffsets; sa.lengths = ctx->db->dLengths;
            sa.targetIds = (const uint32_t *) ctx->tids.p; sa.nPairs = (int) total;
            sa.profSS = (const uint32_t *) ctx->img.p; sa.profAA = nullptr;
            sa.tileBase = 0; sa.rowsInTile = 0; sa.segLen = 1;
            sa.go = (uint32_t) gapOpen | ((uint32_t) gapOpen << 16);
            sa.ge = (uint32_t) gapExtend | ((uint32_t) gapExtend << 16);
            sa.tileIn = 0; sa.tileOut = 0; sa.borderIn = nullptr; sa.borderOut = nullptr; sa.borderStride = 0; sa.keys = nullptr;
            sa.res0 = (int32_t *) ctx->res0.p; sa.res1 = nullptr;
            sa.blocks = (const SwBlockDesc *) ((const unsigned char *) ctx->img.p + descOff) + g.blk0;
            sa.dir = dir;
            rc = launchSwBlocks2(ctx, g.R, hasAA, sa, (int) g.nblk, ppb, gs);
            if (rc != FSGPU_OK) return rc;
            if (gi > 0) { HIPCHK(hipEventRecord(ctx->swAuxEv[gi], gs)); HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->swAuxEv[gi], 0)); }
            gi++;
        }
    }
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
    HIPCHK(hipEventRecord(ctx->swDirEv[2 * dir + 1], ctx->stream));
    ctx->swDirValid[dir] = true;
    ctx->evValid[1] = true;
    if (total) {
        HIPCHK(hipMemcpyAsync(ctx->hRes0.p, ctx->res0.p, total * 16, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = syncStream(ctx)) != FSGPU_OK) return rc;
        const fsgpu_swres *r0 = (const fsgpu_swres *) ctx->hRes0.p;
        for (int i = 0; i < nq; i++)
            if (cls[i] > 0)
                for (int k = 0; k < nSel(i); k++) out[base[i] + perm[sbase[i] + k]] = r0[sbase[i] + k];
    }
    rc = swLongCollect(ctx, longPlan, base, dir, out);
    longGuard.armed = false;
    if (rc != FSGPU_OK) return rc;
    if (dir == 1) ctx->swLongRev.clear();
    // int16-saturated pairs: the single-query path re-runs them with the int32 kernel (computes both directions, keeps `dir`)
    std::vector<fsgpu_swres> f2, r2;
    for (int i = 0; i < nq; i++) {
        const int ns = nSel(i);
        if (ns == 0) continue;
        std::vector<uint32_t> ids;
        std::vector<int> where;
        for (int k = 0; k < ns; k++) {
            const int j = selIdx(i, k);
            if (out[base[i] + j].score == 32767) { ids.push_back(q[i].targetIds[j]); where.push_back(j); }
        }
        if (ids.empty()) continue;
        f2.resize(ids.size()); r2.resize(ids.size());
        rc = fsgpu_sw_batch(ctx, q[i].pAA_fwd, q[i].p3Di_fwd, q[i].pAA_rev, q[i].p3Di_rev, q[i].L, ids.data(), (int) ids.size(), gapOpen, gapExtend,
                            f2.data(), r2.data());
        if (rc != FSGPU_OK) return rc;
        for (size_t k = 0; k < ids.size(); k++) out[base[i] + where[k]] = dir == 0 ? f2[k] : r2[k];
    }
    return FSGPU_OK;
}

// ---- compact-query form of fsgpu_sw_multi_dir: k_sw3 over device-built images ---------------------------------------------------
// A structurealign profile is matrix column + position bias (StructureSmithWaterman.cpp:1566-1640), so a query is given by its codes and
// biases; the LDS images are built by k_sw3_image.  Queries of up to 32 * 16 rows run with 32 lanes per target pair (four targets per
// wave), up to 64 * 16 rows with 64 lanes; longer ones and the int32 re-run of int16-saturated pairs go through the profile-based
// entry points with profiles materialised here.
static bool sw3Class(int L, int &R, int &HL) {
    if (L <= 32 * kSw3MaxR) { HL = 32; R = (L + 31) / 32; return true; }
    if (L <= 64 * kSw3MaxR) { HL = 64; R = (L + 63) / 64; return true; }
    return false;
}
static int sw3Waves(int R, int HL, bool hasAA) {
    static const int env = [] { const char *e = getenv("FSGPU_SW3_WAVES"); const int v = e ? atoi(e) : 0; return (v == 2 || v == 4 || v == 8) ? v : 0; }();
    if (env) return env;
    return (160 * 1024) / sw3LdsBytes(R, HL, hasAA, 4) >= 3 ? 4 : 8;
}
static void sw3Materialize(const int8_t *mat, const uint8_t *codes, const int8_t *cb, int L, bool reversed, std::vector<int16_t> &out) {
    out.resize((size_t) kAlphabet * L);
    for (int a = 0; a < kAlphabet; a++)
        for (int i = 0; i < L; i++)
            out[(size_t) a * L + i] = (int16_t) ((int) mat[a * kAlphabet + codes[reversed ? L - 1 - i : i]] + (cb ? (int) cb[i] : 0));
}

// dir 0 / 1: one direction into out; dir 2: both directions in ONE submission (forward into out, reversed into out2)
static int sw3MultiImpl(fsgpu_ctx *ctx, const int8_t *mat3Di, const int8_t *matAA, const fsgpu_sw_cquery *q, int nq, int gapOpen, int gapExtend, int dir,
                        const int32_t *const *sel, const int32_t *nsel, fsgpu_swres *out, fsgpu_swres *out2) {
    if (!ctx || !mat3Di || nq < 0 || (nq > 0 && (!q || !out)) || dir < 0 || dir > 2 || (dir == 2 && nq > 0 && !out2) || ((sel == nullptr) != (nsel == nullptr))) return FSGPU_E_ARG;
    const int slot = dir == 1 ? 1 : 0;                // accounting slot of fsgpu_sw_last_passes
    const int nDirs = dir == 2 ? 2 : 1, dir0 = dir == 2 ? 0 : dir;
    if (!ctx->db || ctx->db->n == 0) { ctx->err = "no database loaded"; return FSGPU_E_NODB; }
    if (!(gapOpen > gapExtend && gapExtend >= 0 && gapOpen < 32768)) {
        ctx->err = "device SW requires gapOpen > gapExtend >= 0 (the striped reference kernel's lazy-F shortcut is only reproduced for that case)";
        return FSGPU_E_UNSUPPORTED;
    }
    if (ctx->sw.pending) { ctx->err = "previous SW batch not finished"; return FSGPU_E_ARG; }
    if (nq > 65535) { ctx->err = "fsgpu_sw_multi_dir_c: more than 65535 queries in one call"; return FSGPU_E_ARG; }
    const bool hasAA = matAA != nullptr;
    if (hasAA && !ctx->db->hasAA) { ctx->err = "AA matrix given but the database was loaded without AA sequences"; return FSGPU_E_NODB; }
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t S = ctx->swHi ? ctx->swHi : ctx->stream;          // everything of the k_sw3 path: uploads, image build, launches, download
    // an error return after work was enqueued must not leave copies out of the pinned staging buffers or kernels in flight: the next call (or
    // fsgpu_destroy) would refill / free memory that is still being read
    struct Drain {
        fsgpu_ctx *c; hipStream_t s; bool ok = false;
        ~Drain() { if (ok) return; (void) hipStreamSynchronize(s); for (int i = 0; i < fsgpu_ctx::kSwAux; i++) if (c->swAux[i]) (void) hipStreamSynchronize(c->swAux[i]); (void) hipGetLastError(); }
    } drain{ctx, S};
    std::vector<size_t> base(nq + 1, 0), sbase(nq + 1, 0);
    for (int i = 0; i < nq; i++) {
        if (!q[i].q3Di || (hasAA && !q[i].qAA) || q[i].L <= 0 || q[i].L > FSGPU_MAX_SEQ_LEN || q[i].n < 0 || (q[i].n > 0 && !q[i].targetIds)) { ctx->err = "fsgpu_sw_multi_dir_c: bad query"; return FSGPU_E_ARG; }
        for (int k = 0; k < q[i].L; k++) if (q[i].q3Di[k] >= kAlphabet || (hasAA && q[i].qAA[k] >= kAlphabet)) { ctx->err = "fsgpu_sw_multi_dir_c: residue code out of range"; return FSGPU_E_ARG; }
        base[i + 1] = base[i] + (size_t) q[i].n;
        const int ns = sel ? nsel[i] : q[i].n;
        if (ns < 0 || ns > q[i].n || (sel && ns > 0 && !sel[i])) { ctx->err = "fsgpu_sw_multi_dir_c: bad selection"; return FSGPU_E_ARG; }
        for (int k = 0; k < q[i].n; k++) if (q[i].targetIds[k] >= ctx->db->n) { ctx->err = "target id out of range"; return FSGPU_E_ARG; }
        if (sel) for (int k = 0; k < ns; k++) if (sel[i][k] < 0 || sel[i][k] >= q[i].n) { ctx->err = "fsgpu_sw_multi_dir_c: selection index out of range"; return FSGPU_E_ARG; }
    }
    auto nSelAll = [&](int i) { return sel ? (int) nsel[i] : q[i].n; };
    auto selIdx = [&](int i, int k) { return sel ? sel[i][k] : k; };
    int rc;
    // ---- queries outside k_sw3's classes: the profile-based path, all of them in one sub-call (the same set in both directions) ----
    std::vector<int> cR(nq, 0);                       // > 0: the query runs through k_sw3
    std::vector<int> classic;
    for (int i = 0; i < nq; i++) { int hl; if (!sw3Class(q[i].L, cR[i], hl)) { cR[i] = 0; classic.push_back(i); } }
    struct Prof { std::vector<int16_t> aF, sF, aR, sR; };
    auto profilesOf = [&](int i, Prof &pr) {
        sw3Materialize(mat3Di, q[i].q3Di, q[i].cb3Di_fwd, q[i].L, false, pr.sF);
        sw3Materialize(mat3Di, q[i].q3Di, q[i].cb3Di_rev, q[i].L, true, pr.sR);
        if (hasAA) { sw3Materialize(matAA, q[i].qAA, q[i].cbAA_fwd, q[i].L, false, pr.aF); sw3Materialize(matAA, q[i].qAA, q[i].cbAA_rev, q[i].L, true, pr.aR); }
    };
    double clMs = 0, clCells = 0, clPairs = 0, clSteps = 0;          // what the sub-call below ran
    if (!classic.empty()) {
        std::vector<Prof> prof(classic.size());
        std::vector<fsgpu_sw_query> cq(classic.size());
        std::vector<const int32_t *> csel(classic.size());
        std::vector<int32_t> cnsel(classic.size());
        size_t ctotal = 0;
        for (size_t c = 0; c < classic.size(); c++) {
            const int i = classic[c];
            profilesOf(i, prof[c]);
            cq[c].pAA_fwd = hasAA ? prof[c].aF.data() : nullptr; cq[c].pAA_rev = hasAA ? prof[c].aR.data() : nullptr;
            cq[c].p3Di_fwd = prof[c].sF.data(); cq[c].p3Di_rev = prof[c].sR.data();
            cq[c].L = q[i].L; cq[c].n = q[i].n; cq[c].targetIds = q[i].targetIds;
            csel[c] = sel ? sel[i] : nullptr; cnsel[c] = sel ? nsel[i] : 0;
            ctotal += (size_t) q[i].n;
        }
        std::vector<fsgpu_swres> cout(std::max<size_t>(ctotal, 1));
        for (int d = 0; d < nDirs; d++) {
            rc = fsgpu_sw_multi_dir(ctx, cq.data(), (int) cq.size(), gapOpen, gapExtend, dir0 + d, sel ? csel.data() : nullptr, sel ? cnsel.data() : nullptr, cout.data());
            if (rc != FSGPU_OK) return rc;
            {   // the sub-call's pass belongs to this submission's accounting (fsgpu_sw_last_passes): it is reset and re-recorded for the k_sw3 launches below
                double pp[8];
                fsgpu_sw_last_passes(ctx, pp);
                const int cs = dir0 + d;
                if (pp[cs * 4] >= 0) { clMs += pp[cs * 4]; clCells += pp[cs * 4 + 1]; clPairs += pp[cs * 4 + 2]; clSteps += pp[cs * 4 + 3]; }
            }
            fsgpu_swres *dst = d == 0 ? out : out2;
            size_t cb = 0;
            for (size_t c = 0; c < classic.size(); c++) {
                const int i = classic[c];
                for (int k = 0; k < nSelAll(i); k++) { const int j = selIdx(i, k); dst[base[i] + j] = cout[cb + j]; }
                cb += (size_t) q[i].n;
            }
        }
    }
    // ---- k_sw3 ----
    // A query of up to 512 rows has two shapes: 32 lanes per target pair (R32 = ceil(L / 32) rows per lane, four targets per wave: fewest
    // instructions per cell) and 64 lanes (R64 = ceil(L / 64), two targets per wave: half the instructions per target COLUMN).  A wave's
    // run time is (columns + lanes - 1) steps of ~(14 R + 16) dependent-ish instructions, so the longest targets of a launch set its
    // critical path: pairs whose target is longer than the threshold take the 64-lane shape, the others the 32-lane one.
    // (32 queries x 1000 random targets, forward pass alone on the device: thresholds 384 / 640 / 896 / none = 1.21 / 1.20 / 1.20 / 1.21 ms for 3Di,
    // 1.37 / 1.30 / 1.28 / 1.31 ms for 3Di + AA -- the split matters little once all classes share a launch; FSGPU_SW3_LONG overrides it)
    static const int longT = [] { const char *e = getenv("FSGPU_SW3_LONG"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 896; }();
    // Round 6: a third shape, 16 lanes per target pair (eight targets per wave, up to 24 rows per lane), for queries of up to 384 rows: least
    // fill / drain, bookkeeping and row padding per cell, but twice the run time per target column of the 32-lane shape -- it takes the pairs
    // whose target has at most FSGPU_SW3_MID columns (0 switches the shape off).
    // Measured (tools/sw2_probe.py N, forward pass alone, fraction of the issue bound without / with the 16-lane shape): N = 32 queries x 1000 targets
    // 0.57 / 0.49, 64: 0.67 / 0.59, 128: 0.69 / 0.72, 256: 0.72 / 0.75 (3Di; 3Di + AA the same picture) -- its waves are half as many and twice as
    // long, which a launch of one or two rounds of waves pays for in its tail.  All-vs-all's lists of ~8 pairs per query want the opposite: the LDS
    // image of a query (34-45 KB with AA) admits three workgroups per CU whatever the shape, so the shape with the MOST waves per pair keeps the SIMDs
    // busiest (a batch of 1024 queries solo: 16 lanes 1.42 ms, 32 lanes 1.03 ms, 64 lanes 0.88 ms).  Hence the automatic rule: a call whose lists
    // hold at most 16 pairs on average runs with 64 lanes per pair throughout; the 16-lane shape is taken when the call holds at least 100 000 pairs.
    // FSGPU_SW3_MID=<columns> forces the 16-lane shape for every query (0: never), FSGPU_SW3_SHORT=<pairs> moves the short-list limit (0: off).
    const int midEnv = [] { const char *e = getenv("FSGPU_SW3_MID"); return e && *e ? atoi(e) : -1; }();      // read per call: the tests switch shapes inside one process
    const int midT = midEnv >= 0 ? midEnv : 512;
    const int shortList = [] { const char *e = getenv("FSGPU_SW3_SHORT"); return e && *e ? atoi(e) : 16; }();
    static const int maxR16 = [] { const char *e = getenv("FSGPU_SW3_MAXR16"); const int v = e ? atoi(e) : 0; return v > 0 && v <= kSw3MaxR16 ? v : kSw3MaxR16; }();
    auto nSel = [&](int i) { return cR[i] > 0 ? nSelAll(i) : 0; };
    for (int i = 0; i < nq; i++) sbase[i + 1] = sbase[i] + (size_t) nSel(i);
    const size_t total = sbase[nq];
    if (!ctx->swDirEv[3]) for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&ctx->swDirEv[i]));
    ctx->swDirValid[slot] = false;
    if (slot == 0) ctx->swDirValid[1] = false;
    ctx->swDirCells[slot] = clCells; ctx->swDirPairs[slot] = clPairs; ctx->swDirWaveSteps[slot] = clSteps;
    ctx->swDirExtraMs[slot] = clMs;
    if (total == 0) {
        if (!classic.empty()) {          // every query of the call was row-tiled: an empty k_sw3 interval carries the sub-call's figures
            HIPCHK(hipEventRecord(ctx->swDirEv[2 * slot], S)); HIPCHK(hipEventRecord(ctx->swDirEv[2 * slot + 1], S));
            if ((rc = syncStreamOf(ctx, S)) != FSGPU_OK) return rc;
            ctx->swDirValid[slot] = true;
        }
        drain.ok = true;
        return FSGPU_OK;
    }
    const std::vector<int32_t> &len = ctx->db->hLengths;
    // target ids of the pass, longest first inside a query (neighbours share a wave), and the split into the three shapes:
    // pairs [0, nLong) of a query's sorted list run with 64 lanes, [nLong, nLong + nMid) with 32, the rest with 16
    if ((rc = ensurePinned(ctx, ctx->hS3pass, total * 4 + 64)) != FSGPU_OK) return rc;      // grown below once the descriptors are counted
    std::vector<uint32_t> perm(total);
    std::vector<int> nLong(nq, 0), nMid(nq, 0);
    // the short-list rule is per CALL (mean pairs per query of the call): a per-query rule left all-vs-all's batches with 64-lane groups for the short
    // lists AND 32-lane groups for the others -- 1.29 ms per batch against 0.88 ms with one shape for the whole call
    size_t nActive = 0;
    for (int i = 0; i < nq; i++) if (nSel(i) > 0) nActive++;
    const bool callShort = shortList > 0 && nActive > 0 && total <= (size_t) shortList * nActive;
    {
        std::vector<uint64_t> lkey;
        for (int i = 0; i < nq; i++) {
            const int ns = nSel(i);
            if (ns == 0) continue;
            uint32_t *p = perm.data() + sbase[i];
            const uint32_t *ids = q[i].targetIds;
            lkey.resize(ns);
            for (int k = 0; k < ns; k++) { const int j = selIdx(i, k); lkey[k] = ((uint64_t) (0xFFFFFF - len[ids[j]]) << 32) | (uint32_t) j; }
            std::sort(lkey.begin(), lkey.end());
            int nl = 0, nm = 0;
            for (int k = 0; k < ns; k++) { p[k] = (uint32_t) lkey[k]; const int lt = len[ids[p[k]]]; if (lt > longT) nl++; else if (lt > midT) nm++; }
            nLong[i] = (q[i].L > 32 * kSw3MaxR || callShort) ? ns : nl;
            const bool shape16 = q[i].L <= 16 * maxR16 && midT > 0 && (midEnv >= 0 || total >= 100000);
            nMid[i] = nLong[i] == ns ? 0 : shape16 ? nm : ns - nLong[i];
        }
    }
    constexpr int kShapeHL[3] = {16, 32, 64};
    auto nShape = [&](int i, int shape) { return shape == 2 ? nLong[i] : shape == 1 ? nMid[i] : nSel(i) - nLong[i] - nMid[i]; };
    auto firstOfShape = [&](int i, int shape) { return shape == 2 ? 0 : shape == 1 ? nLong[i] : nLong[i] + nMid[i]; };
    // images: built once per set of queries (the reversed call of a forward call finds them in place)
    uint64_t sig = 0xcbf29ce484222325ull ^ (uint64_t) nq ^ ((uint64_t) hasAA << 40);
    sig = hashWords(sig, mat3Di, kAlphabet * kAlphabet);
    if (hasAA) sig = hashWords(sig, matAA, kAlphabet * kAlphabet);
    for (int i = 0; i < nq; i++) {
        if (cR[i] == 0) continue;
        const size_t L = (size_t) q[i].L;
        sig = hashWords(sig ^ (uint64_t) i * 0x9E3779B97F4A7C15ull ^ L, q[i].q3Di, L);
        if (hasAA) sig = hashWords(sig, q[i].qAA, L);
        const int8_t *cbs[4] = {q[i].cb3Di_fwd, q[i].cbAA_fwd, q[i].cb3Di_rev, q[i].cbAA_rev};
        for (int c = 0; c < 4; c++) { if (cbs[c]) sig = hashWords(sig, cbs[c], L); else sig = (sig ^ 0x55) * 0x100000001B3ull; }
    }
    if (!sig) sig = 1;
    bool haveImages = ctx->s3Sig == sig && (int) ctx->s3ImgOff.size() == 3 * nq;
    for (int i = 0; i < nq && haveImages; i++)
        for (int shape = 0; shape < 3; shape++) if (nShape(i, shape) > 0 && ctx->s3ImgOff[3 * i + shape] == 0xffffffffu) haveImages = false;
    if (!haveImages) {
        ctx->s3Sig = 0;
        ctx->s3ImgOff.assign((size_t) 3 * nq, 0xffffffffu);       // [3 i + shape]: image of query i for 16 / 32 / 64 lanes per target pair
        size_t imgDw = 0, dataBytes = 0;
        int nImg = 0, maxDw = 0;
        for (int i = 0; i < nq; i++) {
            if (nSel(i) == 0) continue;
            for (int shape = 0; shape < 3; shape++) {
                if (nShape(i, shape) == 0) continue;
                const int HL = kShapeHL[shape], R = (q[i].L + HL - 1) / HL;
                const size_t one = (size_t) 2 * sw3ImageBytes(R, HL, hasAA) / 4;
                if (imgDw + one >= (1ull << 32)) { ctx->err = "fsgpu_sw_multi_dir_c: images of one call exceed 16 GiB"; return FSGPU_E_NOMEM; }
                ctx->s3ImgOff[3 * i + shape] = (uint32_t) imgDw; imgDw += one; maxDw = std::max(maxDw, (int) one);
                nImg++;
            }
            dataBytes += ((size_t) 6 * q[i].L + 15) / 16 * 16;
        }
        const size_t descBytes = ((size_t) nImg * sizeof(Sw3ImgQuery) + 15) / 16 * 16, matOff = descBytes, dataOff0 = matOff + 1024;
        if ((rc = ensurePinned(ctx, ctx->hS3build, dataOff0 + dataBytes)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->s3build, dataOff0 + dataBytes)) != FSGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->s3img, imgDw * 4)) != FSGPU_OK) return rc;
        unsigned char *hb = (unsigned char *) ctx->hS3build.p;
        Sw3ImgQuery *hd = (Sw3ImgQuery *) hb;
        memcpy(hb + matOff, mat3Di, kAlphabet * kAlphabet);
        if (hasAA) memcpy(hb + matOff + 512, matAA, kAlphabet * kAlphabet);
        size_t dpos = dataOff0;
        int k = 0;
        for (int i = 0; i < nq; i++) {
            if (nSel(i) == 0) continue;
            const size_t L = (size_t) q[i].L;
            for (int shape = 0; shape < 3; shape++) {
                if (ctx->s3ImgOff[3 * i + shape] == 0xffffffffu) continue;
                const int HL = kShapeHL[shape];
                hd[k].imgOff = ctx->s3ImgOff[3 * i + shape]; hd[k].dataOff = (uint32_t) (dpos - dataOff0); hd[k].L = (uint32_t) L;
                hd[k].R = (uint16_t) ((L + HL - 1) / HL); hd[k].HL = (uint16_t) HL;
                k++;
            }
            unsigned char *d = hb + dpos;
            memcpy(d, q[i].q3Di, L);
            if (hasAA) memcpy(d + L, q[i].qAA, L); else memset(d + L, 0, L);
            const int8_t *cbs[4] = {q[i].cb3Di_fwd, q[i].cbAA_fwd, q[i].cb3Di_rev, q[i].cbAA_rev};
            for (int c = 0; c < 4; c++) { if (cbs[c]) memcpy(d + (2 + c) * L, cbs[c], L); else memset(d + (2 + c) * L, 0, L); }
            dpos += (6 * L + 15) / 16 * 16;
            if (dpos - dataOff0 >= (1ull << 32)) { ctx->err = "fsgpu_sw_multi_dir_c: query data of one call exceeds 4 GiB"; return FSGPU_E_NOMEM; }
        }
        HIPCHK(hipMemcpyAsync(ctx->s3build.p, hb, dpos, hipMemcpyHostToDevice, S));
        const unsigned char *db = (const unsigned char *) ctx->s3build.p;
        rc = fsgpuLaunchSw3Image(ctx, (const Sw3ImgQuery *) db, nImg, maxDw, db + dataOff0, (const int8_t *) (db + matOff), (const int8_t *) (db + matOff + 512),
                                 (uint32_t *) ctx->s3img.p, hasAA, S);
        if (rc != FSGPU_OK) return rc;
        ctx->s3Sig = sig;
    }
    // ---- workgroup descriptors: one launch per (lanes per target pair, rows-per-lane range 1..4 / 5..8 / 9..12 / 13..16) ----
    struct Part { int q, key, R, first, n; };       // pairs [first, first + n) of query q's sorted list run with R rows per lane in launch group `key`
    std::vector<Part> parts;
    // launch groups: lanes per target pair x kernel (R = 1..8 / 9..16) x LDS occupancy class of the R range of four (the dynamic LDS of a
    // launch is that of its largest R: a query of 9 rows per lane must not take the 106 KB of one with 16 and lose its second workgroup per CU)
    // (a class whose largest member still fits three workgroups per CU shares its launch with the smaller ones: the register classes of a
    // search batch then run as one or two launches, each with a single long-target tail)
    auto occOf = [&](int HL, int R) { return std::min(3, (160 * 1024) / sw3LdsBytes(std::min(sw3MaxR(HL), (R + 3) / 4 * 4), HL, hasAA, 4)); };
    auto keyOf = [&](int shape, int R) { return shape * 12 + ((R - 1) / 8) * 4 + occOf(kShapeHL[shape], R); };
    for (int i = 0; i < nq; i++) {
        if (nSel(i) == 0) continue;
        for (int shape = 2; shape >= 0; shape--) {
            if (nShape(i, shape) == 0) continue;
            const int R = (q[i].L + kShapeHL[shape] - 1) / kShapeHL[shape];
            parts.push_back({i, keyOf(shape, R), R, firstOfShape(i, shape), nShape(i, shape)});
        }
    }
    struct Group { int key, HL, rlo, maxR, waves, lds; size_t blk0, nblk; };
    std::vector<Group> groups;
    size_t nBlocks = 0;
    // (workgroups of one or two waves for lists that fit them -- all-vs-all's ~8 pairs per query -- were measured and lost: 1.30 against 1.08 ms per
    // batch of 1024 solo; the 34 KB image of a workgroup then arrives through 64 lanes, and the extra launch groups queue behind each other)
    for (int key = 35; key >= 0; key--) {           // the 64-lane groups (the long targets) first
        Group g{key, kShapeHL[key / 12], ((key % 12) / 4) * 8 + 1, 0, 0, 0, nBlocks, 0};
        for (const Part &pt : parts) if (pt.key == key) g.maxR = std::max(g.maxR, pt.R);
        if (g.maxR == 0) continue;
        g.waves = sw3Waves(g.maxR, g.HL, hasAA);
        g.lds = sw3LdsBytes(g.maxR, g.HL, hasAA, g.waves);
        const size_t ppb = (size_t) g.waves * 2 * (64 / g.HL);
        for (const Part &pt : parts) if (pt.key == key) g.nblk += ((size_t) pt.n + ppb - 1) / ppb;
        nBlocks += g.nblk;
        groups.push_back(g);
    }
    const size_t descOff = (total * 4 + 15) / 16 * 16;
    if ((rc = ensurePinned(ctx, ctx->hS3pass, descOff + nBlocks * sizeof(SwBlockDesc))) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->s3pass, descOff + nBlocks * sizeof(SwBlockDesc))) != FSGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->s3res, total * 16 * nDirs)) != FSGPU_OK) return rc;
    if ((rc = ensurePinned(ctx, ctx->hS3res, total * 16 * nDirs)) != FSGPU_OK) return rc;
    uint32_t *hTids = (uint32_t *) ctx->hS3pass.p;
    SwBlockDesc *hBlk = (SwBlockDesc *) ((unsigned char *) ctx->hS3pass.p + descOff);
    for (int i = 0; i < nq; i++) {
        const uint32_t *p = perm.data() + sbase[i];
        for (int k = 0; k < nSel(i); k++) hTids[sbase[i] + k] = q[i].targetIds[p[k]];
    }
    {
        double cells = 0, pairs = 0, winsts = 0;
        for (Group &g : groups) {
            const int ppb = g.waves * 2 * (64 / g.HL), ppw = 2 * (64 / g.HL);
            size_t bp = g.blk0;
            for (const Part &pt : parts) {
                if (pt.key != g.key) continue;
                const int i = pt.q, L = q[i].L, lanes = (L + pt.R - 1) / pt.R;
                for (int p0 = 0; p0 < pt.n; p0 += ppb) {
                    SwBlockDesc &d = hBlk[bp++];
                    d.imgOff = ctx->s3ImgOff[3 * i + g.key / 12]; d.firstPair = (uint32_t) (sbase[i] + pt.first + p0); d.nPairs = (uint16_t) std::min(ppb, pt.n - p0);
                    d.rowsInTile = (uint16_t) L; d.segLen = (uint32_t) ((L + 15) / 16);
                }
                // accounting in the units of the kernel's roofline: DP cells and the VALU wave-instructions its waves issue (a wave runs
                // (longest of its targets) + lanes - 1 steps of 14 packed instructions per register row + 16 around them [+ the AA adds])
                const uint32_t *tp = hTids + sbase[i] + pt.first;
                const double perStep = 14.0 * pt.R + 16.0 + (hasAA ? 2.0 * sw3Dw(pt.R) + 4.0 : 0.0);
                for (int k = 0; k < pt.n; k++) {
                    const int lt = len[tp[k]];
                    cells += (double) L * lt;
                    if ((k % ppb) % ppw == 0 && lt > 0) winsts += (double) (lt + lanes - 1) * perStep;
                }
                pairs += pt.n;
            }
            // first pair of a workgroup is its longest: the workgroups with the most work per wave first (steps x instructions per step: a query of 15
            // rows per lane runs twice the instructions per column of one with 7; FSGPU_SW3_LPT=0: by target length alone, as until round 5)
            static const bool byWork = [] { const char *e = getenv("FSGPU_SW3_LPT"); return !(e && atoi(e) == 0); }();
            const int hl = g.HL;
            auto work = [&](const SwBlockDesc &x) {
                const long lt = len[hTids[x.firstPair]];
                if (!byWork) return lt;
                const long R = ((long) x.rowsInTile + hl - 1) / hl;
                return (lt + hl) * (15 * R + 18);
            };
            std::stable_sort(hBlk + g.blk0, hBlk + g.blk0 + g.nblk, [&](const SwBlockDesc &x, const SwBlockDesc &y) { return work(x) > work(y); });
        }
        ctx->swDirCells[slot] = clCells + cells * nDirs; ctx->swDirPairs[slot] = clPairs + pairs * nDirs; ctx->swDirWaveSteps[slot] = clSteps + winsts * nDirs;
    }
    HIPCHK(hipMemcpyAsync(ctx->s3pass.p, ctx->hS3pass.p, descOff + nBlocks * sizeof(SwBlockDesc), hipMemcpyHostToDevice, S));
    // FSGPU_SW_EXCLUSIVE=1 (A/B measurement): the pass takes its turn in the database's chain of scan batches (DbStore::lastScanDone) --
    // it starts when the scan batch enqueued before it is done and the next scan batch starts behind it -- instead of co-running with them from a
    // high-priority stream.  Uploads and the image build above stay outside the chain.
    static const bool swExclusive = [] { const char *e = getenv("FSGPU_SW_EXCLUSIVE"); return e && atoi(e) != 0; }();
    std::unique_lock<std::mutex> chain(ctx->db->scanMutex, std::defer_lock);
    if (swExclusive) {
        if (!ctx->swChainEv) HIPCHK(hipEventCreateWithFlags(&ctx->swChainEv, hipEventDisableTiming));
        chain.lock();
        if (ctx->db->lastScanDone && ctx->db->lastScanDone != ctx->swChainEv) HIPCHK(hipStreamWaitEvent(S, ctx->db->lastScanDone, 0));
    }
    if (slot == 0 || !ctx->evValid[1]) HIPCHK(hipEventRecord(ctx->ev[2], S));
    HIPCHK(hipEventRecord(ctx->swDirEv[2 * slot], S));
    // every launch group gets a stream: their long-target tails overlap instead of queueing up
    // ... and, in the one-submission form, the two directions of a group: the forward and the reversed-query launch of all-vs-all's batches are one round of
    // waves each (0.69 ms apiece for 1024 queries x 8 pairs, the wavefront of the longest target) and ran one behind the other on the group's stream
    static const bool dirStreams = [] { const char *e = getenv("FSGPU_SW3_DIRSTREAMS"); return !(e && atoi(e) == 0); }();          // 0: as until round 5 (A/B)
    const size_t nStreams = std::min<size_t>(groups.size() * (size_t) (dirStreams ? nDirs : 1), dirStreams ? (size_t) fsgpu_ctx::kSwAux : 6);
    if (nStreams > 1) {
        if (!ctx->swAuxEv[fsgpu_ctx::kSwAux]) for (int i = 0; i <= fsgpu_ctx::kSwAux; i++) HIPCHK(hipEventCreateWithFlags(&ctx->swAuxEv[i], hipEventDisableTiming));
        for (size_t k = 1; k < nStreams; k++) if (!ctx->swAux[k]) { if (!ctx->swCuMask.empty()) HIPCHK(hipExtStreamCreateWithCUMask(&ctx->swAux[k], (uint32_t) ctx->swCuMask.size(), ctx->swCuMask.data())); else if (ctx->swHi) HIPCHK(hipStreamCreateWithPriority(&ctx->swAux[k], hipStreamNonBlocking, ctx->swHiPrio)); else HIPCHK(hipStreamCreateWithFlags(&ctx->swAux[k], hipStreamNonBlocking)); }
        HIPCHK(hipEventRecord(ctx->swAuxEv[fsgpu_ctx::kSwAux], S));
        for (size_t k = 1; k < nStreams; k++) HIPCHK(hipStreamWaitEvent(ctx->swAux[k], ctx->swAuxEv[fsgpu_ctx::kSwAux], 0));
    }
    for (size_t gx = 0; gx < groups.size() * (size_t) nDirs; gx++) {
        const size_t gi = gx % groups.size();
        const int d = (int) (gx / groups.size());
        const Group &g = groups[gi];
        const size_t k = (dirStreams ? gx : gi) % nStreams;
        hipStream_t gs = k == 0 ? S : ctx->swAux[k];
        Sw3Args sa;
        sa.aa = ctx->db->alnAA; sa.ss = ctx->db->aln3di; sa.offsets = ctx->db->dOffsets; sa.lengths = ctx->db->dLengths;
        sa.targetIds = (const uint32_t *) ctx->s3pass.p;
        sa.img = (const uint32_t *) ctx->s3img.p;
        sa.blocks = (const SwBlockDesc *) ((const unsigned char *) ctx->s3pass.p + descOff) + g.blk0;
        sa.go = (uint32_t) gapOpen | ((uint32_t) gapOpen << 16);
        sa.ge = (uint32_t) gapExtend | ((uint32_t) gapExtend << 16);
        sa.dir = dir0 + d;
        sa.res0 = (int32_t *) ctx->s3res.p + (size_t) d * total * 4;
        rc = hasAA ? fsgpuLaunchSw3AA(ctx, g.rlo, g.HL, sa, (int) g.nblk, g.waves, g.lds, gs) : fsgpuLaunchSw3NA(ctx, g.rlo, g.HL, sa, (int) g.nblk, g.waves, g.lds, gs);
        if (rc != FSGPU_OK) { for (size_t x = 1; x < nStreams; x++) (void) hipStreamSynchronize(ctx->swAux[x]); return rc; }
    }
    for (size_t k = 1; k < nStreams; k++) { HIPCHK(hipEventRecord(ctx->swAuxEv[k], ctx->swAux[k])); HIPCHK(hipStreamWaitEvent(S, ctx->swAuxEv[k], 0)); }
    HIPCHK(hipEventRecord(ctx->ev[3], S));
    HIPCHK(hipEventRecord(ctx->swDirEv[2 * slot + 1], S));
    if (swExclusive) {
        HIPCHK(hipEventRecord(ctx->swChainEv, S));
        ctx->db->lastScanDone = ctx->swChainEv;
        chain.unlock();
    }
    ctx->swDirValid[slot] = true;
    ctx->evValid[1] = true;
    HIPCHK(hipMemcpyAsync(ctx->hS3res.p, ctx->s3res.p, total * 16 * nDirs, hipMemcpyDeviceToHost, S));
    if ((rc = syncStreamOf(ctx, S)) != FSGPU_OK) return rc;
    for (int d = 0; d < nDirs; d++) {
        const fsgpu_swres *r0 = (const fsgpu_swres *) ctx->hS3res.p + (size_t) d * total;
        fsgpu_swres *dst = d == 0 ? out : out2;
        for (int i = 0; i < nq; i++)
            for (int k = 0; k < nSel(i); k++) dst[base[i] + perm[sbase[i] + k]] = r0[sbase[i] + k];
    }
    // int16-saturated pairs: the single-query path re-runs them with the int32 kernel (computes both directions, keeps `dir`)
    std::vector<fsgpu_swres> f2, r2;
    for (int i = 0; i < nq; i++) {
        const int ns = nSel(i);
        if (ns == 0) continue;
        std::vector<uint32_t> ids;
        std::vector<int> where;
        for (int k = 0; k < ns; k++) {
            const int j = selIdx(i, k);
            if (out[base[i] + j].score == 32767 || (dir == 2 && out2[base[i] + j].score == 32767)) { ids.push_back(q[i].targetIds[j]); where.push_back(j); }
        }
        if (ids.empty()) continue;
        Prof pr;
        profilesOf(i, pr);
        f2.resize(ids.size()); r2.resize(ids.size());
        rc = fsgpu_sw_batch(ctx, hasAA ? pr.aF.data() : nullptr, pr.sF.data(), hasAA ? pr.aR.data() : nullptr, pr.sR.data(), q[i].L, ids.data(), (int) ids.size(),
                            gapOpen, gapExtend, f2.data(), r2.data());
        if (rc != FSGPU_OK) return rc;
        for (size_t k = 0; k < ids.size(); k++) {
            if (dir == 2) {           // only the saturated direction is replaced (the other one's int16 result stands, as in two separate passes)
                if (out[base[i] + where[k]].score == 32767) out[base[i] + where[k]] = f2[k];
                if (out2[base[i] + where[k]].score == 32767) out2[base[i] + where[k]] = r2[k];
            } else out[base[i] + where[k]] = dir == 0 ? f2[k] : r2[k];
        }
    }
    drain.ok = true;
    return FSGPU_OK;
}

int fsgpu_sw_multi_dir_c(fsgpu_ctx *ctx, const int8_t *mat3Di, const int8_t *matAA, const fsgpu_sw_cquery *q, int nq, int gapOpen, int gapExtend, int dir,
                         const int32_t *const *sel, const int32_t *nsel, fsgpu_swres *out) {
    if (dir != 0 && dir != 1) return FSGPU_E_ARG;
    return sw3MultiImpl(ctx, mat3Di, matAA, q, nq, gapOpen, gapExtend, dir, sel, nsel, out, nullptr);
}

int fsgpu_sw_multi_c(fsgpu_ctx *ctx, const int8_t *mat3Di, const int8_t *matAA, const fsgpu_sw_cquery *q, int nq, int gapOpen, int gapExtend,
                     fsgpu_swres *fwd, fsgpu_swres *rev) {
    return sw3MultiImpl(ctx, mat3Di, matAA, q, nq, gapOpen, gapExtend, 2, nullptr, nullptr, fwd, rev);
}


// both directions of every pair: two fsgpu_sw_multi_dir passes (callers that gate between the passes save most of the second)
int fsgpu_sw_multi(fsgpu_ctx *ctx, const fsgpu_sw_query *q, int nq, int gapOpen, int gapExtend, fsgpu_swres *fwd, fsgpu_swres *rev) {
    if (!ctx || nq < 0 || (nq > 0 && (!q || !fwd || !rev))) return FSGPU_E_ARG;
    int rc = fsgpu_sw_multi_dir(ctx, q, nq, gapOpen, gapExtend, 0, nullptr, nullptr, fwd);
    if (rc != FSGPU_OK) return rc;
    return fsgpu_sw_multi_dir(ctx, q, nq, gapOpen, gapExtend, 1, nullptr, nullptr, rev);
}

} // extern "C"
