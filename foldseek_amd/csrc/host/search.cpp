// search.cpp -- per-query host logic of the two modules on the hot path:
//   prefilter:  runFilterOnGpu's per-query body            (reference M/src/prefiltering/ungappedprefilter.cpp:170-326)
//   align:      structurealign's per-query body + alignStructure
//                                                          (reference F/src/strucclustutils/structurealign.cpp:37-112,318-452)
// The DP itself runs on the device through fsgpu_gapless_scan / fsgpu_sw_batch; this file only prepares profiles,
// applies the reference's gates in the reference's order, and formats results.
#include "hostlib.h"
#include "block_aligner_abi.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <sched.h>

using namespace fsh;

struct fshost_search {
    fsgpu_ctx *ctx = nullptr;
    fshost_params par;
    Matrix matPref;      // 3Di, 2.0 bits, bias 0.0   (ungappedprefilter.cpp:541)
    Matrix mat3Di;       // 3Di, 2.1 bits             (structurealign.cpp:252)
    Matrix matAA;        // blosum62, 1.4 bits or 0.0 (structurealign.cpp:264-265)
    Evaluer evaluer;
    std::vector<uint32_t> keys;
    const uint8_t *data3di = nullptr, *dataAA = nullptr;   // caller-owned padded DB (host)
    const uint64_t *offsets = nullptr;
    const int32_t *lengths = nullptr;
    std::string err;
    // scratch
    std::vector<int8_t> pssm;
    std::vector<int16_t> pAAf, p3f, pAAr, p3r;
    std::vector<int8_t> cbAA, cbSS;
    std::vector<uint8_t> rAA, r3Di, tAA, t3Di;
    std::vector<fsgpu_swres> fwd, rev;
    std::string cigars;
    int64_t lastDeviceBt = 0, lastBtTasks = 0;  // last fshost_search_align_batch: accepted hits answered by the device aligner / all of them
    std::vector<int8_t> btTblAA, btTbl3;        // device backtrace (fsgpu_block_backtrace): the block aligner's two AAMatrix tables and the code -> letter maps
    std::vector<uint8_t> btLetAA, btLet3;
    std::vector<std::vector<int16_t>> kThr;     // fshost_search_kmer_batch: per-query k-mer thresholds / ungapped profiles of the last batch
    std::vector<std::vector<int8_t>> kProf;
    std::vector<fsgpu_kmer_query> kq;
    // host-side wall time of the last calls, seconds: [0] prefilter profile, [1] prefilter device call (incl. wait),
    // [2] align profiles + e-value net, [3] SW device call (incl. wait), [4] gates, [5] block-aligner backtrace
    double stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
static inline double nowSec() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- host worker pool ------------------------------------------------------------------------------------------------
// The block-aligner backtrace of an accepted hit costs ~20 us on a host core; with 50 accepted hits per query that is
// 1 ms per query -- more than the device needs for prefilter + SW of a query against 100k targets.  The reference spreads
// this over its OpenMP threads (one query per thread, structurealign.cpp:318); here the feeder thread that owns a batch
// hands the backtraces of ALL its queries to a process-wide pool and works on them itself until they are done.
// Size: fshost_set_host_workers(n) / FSGPU_HOST_WORKERS, default min(6, usable cores - 1) where "usable" honours the
// cgroup CPU quota (a container that shows 256 CPUs may own the time of 16).
namespace {

int usableCores() {
    int n = (int) std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    {   // the CPU set this process may run on (a rank pinned to its share of the node's cores)
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min(n, c); }
    }
    // the cgroup's CPU quota, read once (usableCores() is asked per batch since round 6)
    static const long long quota = [] {
        long long v = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64]; long long period = 0;
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) v = std::max<long long>(1, atoll(q) / period);
            fclose(f);
        }
        return v;
    }();
    if (quota > 0) n = (int) std::min<long long>(n, quota);
    return std::max(1, n);
}

class HostPool {
public:
    static HostPool &get() { static HostPool *p = new HostPool(); return *p; }      // leaked on purpose: workers may outlive static destructors
    // runs fn(0..n-1); the caller takes part and returns when all indices are done
    void parallelFor(int n, const std::function<void(int)> &fn) {
        if (n <= 0) return;
        Job job; job.n = n; job.fn = &fn;
        if (n > 1 && workers() > 0) {
            { std::lock_guard<std::mutex> g(m_); jobs_.push_back(&job); }
            cv_.notify_all();
        }
        for (;;) { const int i = job.next.fetch_add(1); if (i >= n) break; fn(i); job.done.fetch_add(1); }
        if (n > 1 && workers() > 0) {
            { std::lock_guard<std::mutex> g(m_); jobs_.erase(std::find(jobs_.begin(), jobs_.end(), &job)); }
            // the tail of the job is in the workers' hands: wait without burning the core they may need (a spinning owner per feeder thread
            // cost the all-vs-all module 10 % of its wall time on a 16-core quota)
            for (int spins = 0; job.done.load(std::memory_order_acquire) < n || job.inside.load(std::memory_order_acquire) > 0; spins++) {
                if (spins < 64) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
        }
    }
    void resize(int n) {
        std::lock_guard<std::mutex> g(m_);
        want_ = std::max(0, std::min(n, 64));
        while ((int) threads_.size() < want_) { const int id = (int) threads_.size(); threads_.emplace_back([this, id] { loop(id); }); threads_.back().detach(); }
        cv_.notify_all();
    }
    int workers() const { return want_; }
private:
    struct Job { int n = 0; const std::function<void(int)> *fn = nullptr; std::atomic<int> next{0}, done{0}, inside{0}; };
    HostPool() {
        const char *e = getenv("FSGPU_HOST_WORKERS");
        resize(e ? atoi(e) : std::min(6, usableCores() - 1));
    }
    void loop(int id) {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            Job *job = nullptr;
            if (id < want_) for (Job *j : jobs_) if (j->next.load(std::memory_order_relaxed) < j->n) { job = j; break; }
            if (!job) { cv_.wait(lk); continue; }
            job->inside.fetch_add(1);          // under the lock: the owner cannot retire the job before it sees us
            lk.unlock();
            for (;;) { const int i = job->next.fetch_add(1); if (i >= job->n) break; (*job->fn)(i); job->done.fetch_add(1, std::memory_order_release); }
            job->inside.fetch_sub(1, std::memory_order_release);
            lk.lock();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<Job *> jobs_;
    std::vector<std::thread> threads_;
    int want_ = 0;
};

} // namespace

// the process-wide pool for other translation units of the library (fsgpu_kmer.hip: the per-query tail of a k-mer batch)
void fshostParallelFor(int n, const std::function<void(int)> &fn) { HostPool::get().parallelFor(n, fn); }

extern "C" {
void fshost_set_host_workers(int n) { HostPool::get().resize(n); }
int fshost_host_workers(void) { return HostPool::get().workers(); }
int fshost_usable_cores(void) { return usableCores(); }
}

extern "C" {

void fshost_params_default(fshost_params *p) {
    p->maxResListLen = 1000;
    p->minDiagScoreThr = 30;
    p->compBiasCorrection = 1;
    p->prefCompBiasScale = 0.15f;
    p->alignmentType = 2;
    p->alnCompBiasScale = 0.5f;
    p->gapOpen = 10;
    p->gapExtend = 1;
    p->evalThr = 10.0;
    p->covThr = 0.0f;
    p->covMode = 0;
    p->addBacktrace = 0;
    p->maxAccept = INT_MAX;
    p->maxRejected = INT_MAX;
    p->seqIdThr = 0.0f;
    p->alnLenThr = 0;
    p->seqIdMode = 0;
    p->altAlignment = 0;
    p->skipUndefinedDiagonals = 0;
}

fshost_search *fshost_search_create(fsgpu_ctx *ctx, const fshost_params *p, const uint32_t *keys, const char *nnPath,
                                    const uint8_t *data3di, const uint8_t *dataAA, const uint64_t *offsets, const int32_t *lengths) {
    if (!ctx || !p || !data3di || !offsets || !lengths) return nullptr;
    fshost_search *s = new fshost_search();
    s->ctx = ctx;
    s->data3di = data3di; s->dataAA = dataAA; s->offsets = offsets; s->lengths = lengths;
    s->par = *p;
    const uint64_t n = fsgpu_db_size(ctx);
    bool ok = s->matPref.builtin(FSHOST_MAT_3DI, 2.0f, 0.0f) && s->mat3Di.builtin(FSHOST_MAT_3DI, 2.1f, 0.0f) &&
              s->matAA.builtin(FSHOST_MAT_BLOSUM62, p->alignmentType == 2 ? 1.4f : 0.0f, 0.0f);
    if (!ok) s->err = "matrix construction failed";
    if (ok && !s->evaluer.load(nnPath, fsgpu_db_residues(ctx), s->err)) ok = false;
    if (!ok) {
        // keep the handle so the caller can read the error
        s->ctx = nullptr;
        return s;
    }
    s->keys.resize(n);
    for (uint64_t i = 0; i < n; i++) s->keys[i] = keys ? keys[i] : (uint32_t) i;
    return s;
}

void fshost_search_free(fshost_search *s) { delete s; }
const char *fshost_search_error(const fshost_search *s) { return s ? s->err.c_str() : "null handle"; }

int fshost_search_prefilter(fshost_search *s, const uint8_t *q3di, int L, int64_t identityId, fsgpu_hit *hits) {
    if (!s || !s->ctx) return FSGPU_E_ARG;
    const double t0 = nowSec();
    s->pssm.resize((size_t) s->matPref.n * L);
    int cap = 0;
    int rc = prefilterProfile(s->matPref, q3di, L, s->par.compBiasCorrection != 0, s->par.prefCompBiasScale, s->pssm.data(), &cap);
    if (rc != FSGPU_OK) { s->err = "bad query residue code"; return rc; }
    int nout = 0;
    const double t1 = nowSec();
    rc = fsgpu_gapless_scan(s->ctx, s->pssm.data(), L, cap, s->par.minDiagScoreThr, identityId, s->par.maxResListLen, hits, &nout);
    s->stats[0] = t1 - t0; s->stats[1] = nowSec() - t1;
    if (rc != FSGPU_OK) { s->err = fsgpu_last_error(s->ctx); return rc; }
    return nout;
}

// nq queries through fsgpu_gapless_scan_multi: hits[q * maxResListLen ...], nhits[q]; same results as nq fshost_search_prefilter calls
int fshost_search_prefilter_batch(fshost_search *s, int nq, const uint8_t *const *q3di, const int *L, const int64_t *identityId,
                                  fsgpu_hit *hits, int *nhits) {
    if (!s || !s->ctx || nq < 0 || (nq > 0 && (!q3di || !L || !hits || !nhits))) return FSGPU_E_ARG;
    const double t0 = nowSec();
    std::vector<size_t> off(nq + 1, 0);
    for (int i = 0; i < nq; i++) { if (!q3di[i] || L[i] <= 0) return FSGPU_E_ARG; off[i + 1] = off[i] + (size_t) s->matPref.n * L[i]; }
    s->pssm.resize(off[nq]);
    std::vector<fsgpu_gapless_query> gq(nq);
    for (int i = 0; i < nq; i++) {
        int cap = 0;
        const int rc = prefilterProfile(s->matPref, q3di[i], L[i], s->par.compBiasCorrection != 0, s->par.prefCompBiasScale, s->pssm.data() + off[i], &cap);
        if (rc != FSGPU_OK) { s->err = "bad query residue code"; return rc; }
        gq[i].pssm = s->pssm.data() + off[i]; gq[i].L = L[i]; gq[i].scoreCap = cap; gq[i].identityId = identityId ? identityId[i] : -1;
    }
    const double t1 = nowSec();
    const int rc = fsgpu_gapless_scan_multi(s->ctx, gq.data(), nq, s->par.minDiagScoreThr, s->par.maxResListLen, hits, nhits);
    s->stats[0] = t1 - t0; s->stats[1] = nowSec() - t1;
    if (rc != FSGPU_OK) { s->err = fsgpu_last_error(s->ctx); return rc; }
    return FSGPU_OK;
}

} // extern "C"

// ---- helpers restating small reference functions -------------------------------------------------------------
static float computeCov(unsigned int startPos, unsigned int endPos, unsigned int len) {   // StructureSmithWaterman.cpp:2028
    return (std::min(len, std::max(startPos, endPos)) - std::min(startPos, endPos) + 1) / (float) len;
}
static bool canBeCovered(const float covThr, const int covMode, float queryLength, float targetLength) {   // Util.cpp:542
    switch (covMode) {
        case 0: return ((queryLength / targetLength >= covThr) && (targetLength / queryLength >= covThr));
        case 2: return ((targetLength / queryLength) >= covThr);
        case 1: return ((queryLength / targetLength) >= covThr);
        case 3: return ((targetLength / queryLength) >= covThr) && (targetLength / queryLength) <= 1.0;
        case 4: return ((queryLength / targetLength) >= covThr) && (queryLength / targetLength) <= 1.0;
        case 5: return (std::min(targetLength, queryLength) / std::max(targetLength, queryLength)) >= covThr;
        default: return true;
    }
}
static bool hasCoverage(float covThr, int covMode, float queryCov, float targetCov) {   // Util.cpp:561
    switch (covMode) {
        case 0: return ((queryCov >= covThr) && (targetCov >= covThr));
        case 2: return (queryCov >= covThr);
        case 1: return (targetCov >= covThr);
        default: return true;
    }
}
static bool compareHits(const fshost_result &first, const fshost_result &second) {   // Matcher.h:161
    if (first.eval != second.eval) return first.eval < second.eval;
    if (first.score != second.score) return first.score > second.score;
    if (first.dbLen != second.dbLen) return first.dbLen < second.dbLen;
    return first.dbKey < second.dbKey;
}

extern "C" {

} // extern "C"

namespace {

// per-query state of the alignment stage: position biases of the forward and the reversed query, e-value parameters; the int16 profiles
// only where a profile-based device call needs them (single-query path, --alt-ali)
struct AlignQuery {
    const uint8_t *qAA = nullptr, *q3di = nullptr;
    int L = 0;
    std::vector<int16_t> pAAf, p3f, pAAr, p3r;
    std::vector<int8_t> cbAA, cbSS, cbAAr, cbSSr;       // r: biases of the reversed query, indexed by reversed position
    double lambda = 0, mu = 0;
};

// the four word profiles from codes + biases (StructureSmithWaterman.cpp:1566-1640: matrix column + position bias)
void materializeProfiles(const fshost_search *s, AlignQuery &aq) {
    if (!aq.p3f.empty()) return;
    const int A = s->mat3Di.n, L = aq.L;
    aq.pAAf.resize((size_t) A * L); aq.p3f.resize((size_t) A * L); aq.pAAr.resize((size_t) A * L); aq.p3r.resize((size_t) A * L);
    for (int a = 0; a < A; a++)
        for (int i = 0; i < L; i++) {
            const int j = L - 1 - i;
            aq.pAAf[(size_t) a * L + i] = (int16_t) (s->matAA.tiny[a * A + aq.qAA[i]] + aq.cbAA[i]);
            aq.p3f[(size_t) a * L + i] = (int16_t) (s->mat3Di.tiny[a * A + aq.q3di[i]] + aq.cbSS[i]);
            aq.pAAr[(size_t) a * L + i] = (int16_t) (s->matAA.tiny[a * A + aq.qAA[j]] + aq.cbAAr[i]);
            aq.p3r[(size_t) a * L + i] = (int16_t) (s->mat3Di.tiny[a * A + aq.q3di[j]] + aq.cbSSr[i]);
        }
}

// structurealign.cpp:322-347: e-value network + the biases of the forward / reversed-query profiles (+ the profiles themselves on request)
int prepareAlign(fshost_search *s, AlignQuery &aq, std::vector<uint8_t> &rAA, std::vector<uint8_t> &r3Di, bool withProfiles = true, std::string *errOut = nullptr) {
    const fshost_params &par = s->par;
    std::string &serr = errOut ? *errOut : s->err;          // callers that run queries in parallel collect the message themselves
    // the block aligner takes NEGATIVE gap costs with open < extend (block-aligner scan_block.rs: "Gap costs must be negative!"; the
    // reference process dies in that assertion with --gap-extend 0) and the device SW reproduces the striped kernel for open > extend:
    // refuse here, with a message, instead of aborting in the backtrace
    if (!(par.gapExtend >= 1 && par.gapOpen > par.gapExtend && par.gapOpen < 32768)) {
        serr = "gap costs must satisfy gapOpen > gapExtend >= 1 (got " + std::to_string(par.gapOpen) + " / " + std::to_string(par.gapExtend) + ")";
        return FSGPU_E_UNSUPPORTED;
    }
    const int A = s->mat3Di.n, L = aq.L;
    s->evaluer.predictMuLambda(aq.q3di, L, A, &aq.lambda, &aq.mu);
    aq.cbAA.resize(L); aq.cbSS.resize(L); aq.cbAAr.resize(L); aq.cbSSr.resize(L);
    aq.pAAf.clear(); aq.p3f.clear(); aq.pAAr.clear(); aq.p3r.clear();
    rAA.assign(aq.qAA, aq.qAA + L); r3Di.assign(aq.q3di, aq.q3di + L);
    std::reverse(rAA.begin(), rAA.end());
    std::reverse(r3Di.begin(), r3Di.end());
    int rc = alignProfiles(s->matAA, s->mat3Di, aq.qAA, aq.q3di, L, par.compBiasCorrection != 0, par.alnCompBiasScale, nullptr, nullptr,
                           aq.cbAA.data(), aq.cbSS.data());
    if (rc == FSGPU_OK)
        rc = alignProfiles(s->matAA, s->mat3Di, rAA.data(), r3Di.data(), L, par.compBiasCorrection != 0, par.alnCompBiasScale,
                           nullptr, nullptr, aq.cbAAr.data(), aq.cbSSr.data());
    if (rc != FSGPU_OK) { serr = "bad query residue code"; return rc; }
    if (withProfiles) materializeProfiles(s, aq);
    return rc;
}

// the gates a pair passes BEFORE structurealign looks at its reversed-query score (structurealign.cpp:357-361 canBeCovered,
// :50-58 coverage and e-value of the forward alignment): exactly the conditions gateAlign applies below, in the same order
// and arithmetic, so that the reversed pass can be restricted to these pairs
bool needsReversePass(const fshost_search *s, const AlignQuery &aq, uint32_t tid, const fsgpu_swres &f) {
    const fshost_params &par = s->par;
    const int L = aq.L, Lt = s->lengths[tid];
    if (!canBeCovered(par.covThr, par.covMode, (float) L, (float) Lt)) return false;
    const float qCov = computeCov(0, f.qEnd, L), tCov = computeCov(0, f.dbEnd, Lt);
    if (!hasCoverage(par.covThr, par.covMode, qCov, tCov)) return false;
    const double evalue = s->evaluer.computeEvalueCorr((double) (uint32_t) f.score, aq.lambda, aq.mu);
    return !(evalue > par.evalThr);
}

// start position + backtrace of one pair on the host (alignStartPosBacktraceBlock); safe to call from any thread
void pairBacktrace(const fshost_search *s, const AlignQuery &aq, uint32_t tid, const fsgpu_swres &f, BlockAlnOut &bo) {
    static thread_local std::vector<uint8_t> tAA, t3Di;
    const fshost_params &par = s->par;
    const int Lt = s->lengths[tid];
    tAA.resize(Lt); t3Di.resize(Lt);
    for (int i = 0; i < Lt; i++) {          // padded-DB codes: +32 = soft-masked, same letter for the aligner
        uint8_t c = s->data3di[s->offsets[tid] + i];
        c = c >= 32 ? c - 32 : c;
        t3Di[i] = c > 20 ? 20 : c;
        uint8_t a = s->dataAA ? s->dataAA[s->offsets[tid] + i] : 20;
        a = a >= 32 ? a - 32 : a;
        tAA[i] = a > 20 ? 20 : a;
    }
    blockBacktrace(s->matAA, s->mat3Di, aq.qAA, aq.q3di, aq.cbAA.data(), aq.cbSS.data(), aq.L, tAA.data(), t3Di.data(), Lt, f.qEnd, f.dbEnd,
                   f.score, par.gapOpen, par.gapExtend, bo);
}

// the gates of alignStructure that come BEFORE the backtrace (structurealign.cpp:357-361 canBeCovered, :50-58 coverage and
// e-value of the forward alignment, :68-73 e-value of fwd - rev); fills score / evalue of the pair when it passes
bool passesScoreGates(const fshost_search *s, const AlignQuery &aq, uint32_t tid, const fsgpu_swres &f, const fsgpu_swres &r,
                      int32_t &score, double &evalue, bool *lookedAtRev) {
    const fshost_params &par = s->par;
    const int L = aq.L, Lt = s->lengths[tid];
    if (!canBeCovered(par.covThr, par.covMode, (float) L, (float) Lt)) return false;
    const float qCov = computeCov(0, f.qEnd, L), tCov = computeCov(0, f.dbEnd, Lt);
    if (!hasCoverage(par.covThr, par.covMode, qCov, tCov)) return false;
    evalue = s->evaluer.computeEvalueCorr((double) (uint32_t) f.score, aq.lambda, aq.mu);
    if (evalue > par.evalThr) return false;
    if (lookedAtRev) *lookedAtRev = true;
    score = f.score - r.score;
    evalue = s->evaluer.computeEvalueCorr(score, aq.lambda, aq.mu);
    return !(evalue > par.evalThr);
}

// --alt-ali (structurealign.cpp:115-138, 415-429): after an accepted hit, up to altAlignment more alignments of the SAME
// target, each on the target with the previous alignment's range [dbStartPos, dbEndPos) overwritten by X in both alphabets
// (cumulative), through the whole of alignStructure again (forward SW, coverage + e-value gates, reversed SW, e-value of the
// difference, block backtrace) and checkCriteria with isIdentity = false.  The masked sequence exists only here, so the
// pair goes to the device as an explicit sequence (fsgpu_sw_batch_seqs).  first = the accepted result; appends to results.
int alternativeAlignments(fshost_search *s, const AlignQuery &aq, uint32_t tid, const fshost_result &first, fshost_result *results, int &nres, int resCap) {
    const fshost_params &par = s->par;
    const bool useAA = par.alignmentType == 2;
    const int L = aq.L, Lt = s->lengths[tid];
    std::vector<uint8_t> tAA(Lt), t3Di(Lt);
    for (int i = 0; i < Lt; i++) {
        uint8_t c = s->data3di[s->offsets[tid] + i];
        c = c >= 32 ? c - 32 : c;
        t3Di[i] = c > 20 ? 20 : c;
        uint8_t a = s->dataAA ? s->dataAA[s->offsets[tid] + i] : 20;
        a = a >= 32 ? a - 32 : a;
        tAA[i] = a > 20 ? 20 : a;
    }
    fshost_result prev = first;
    const uint64_t offs[2] = {0, (uint64_t) Lt};
    const int32_t lens[1] = {Lt};
    for (int alt = 0; alt < par.altAlignment && nres < resCap; alt++) {
        // a failed block alignment leaves dbStartPos = -1: the reference's loop then starts one byte BEFORE the sequence
        // buffer; the in-range part of that is [0, dbEndPos)
        for (int pos = std::max(prev.dbStartPos, 0); pos < prev.dbEndPos && pos < Lt; ++pos) { tAA[pos] = 20; t3Di[pos] = 20; }
        fsgpu_swres f, r;
        int rc = fsgpu_sw_batch_seqs(s->ctx, useAA ? aq.pAAf.data() : nullptr, aq.p3f.data(), useAA ? aq.pAAr.data() : nullptr, aq.p3r.data(), L,
                                     tAA.data(), t3Di.data(), offs, lens, 1, par.gapOpen, par.gapExtend, &f, &r);
        if (rc != FSGPU_OK) { s->err = fsgpu_last_error(s->ctx); return rc; }
        // ---- alignStructure on the masked target ----
        float qCov = computeCov(0, f.qEnd, L), tCov = computeCov(0, f.dbEnd, Lt);
        if (!hasCoverage(par.covThr, par.covMode, qCov, tCov)) break;
        double evalue = s->evaluer.computeEvalueCorr((double) (uint32_t) f.score, aq.lambda, aq.mu);
        if (evalue > par.evalThr) break;
        const int32_t score = f.score - r.score;
        evalue = s->evaluer.computeEvalueCorr(score, aq.lambda, aq.mu);
        if (evalue > par.evalThr) break;
        BlockAlnOut bo;
        blockBacktrace(s->matAA, s->mat3Di, aq.qAA, aq.q3di, aq.cbAA.data(), aq.cbSS.data(), L, tAA.data(), t3Di.data(), Lt, f.qEnd, f.dbEnd,
                       f.score, par.gapOpen, par.gapExtend, bo);
        fshost_result a;
        memset(&a, 0, sizeof(a));
        int qStart = -1, dbStart = -1;
        float seqId = 0.0f;
        if (bo.ok) {
            qStart = bo.qStart; dbStart = bo.dbStart;
            qCov = computeCov(qStart, f.qEnd, L);
            tCov = computeCov(dbStart, f.dbEnd, Lt);
        }
        unsigned int alnLength = std::max(abs(f.qEnd - qStart), abs(f.dbEnd - dbStart)) + 1;
        if (bo.backtrace.size() > 0) {
            alnLength = bo.backtrace.size();
            const int den = par.seqIdMode == 1 ? std::min(L, Lt) : par.seqIdMode == 2 ? std::max(L, Lt) : (int) alnLength;
            seqId = static_cast<float>(bo.identicalAA) / static_cast<float>(den);
        }
        a.dbKey = s->keys[tid]; a.score = score; a.qcov = qCov; a.dbcov = tCov; a.seqId = seqId; a.eval = evalue;
        a.alnLength = alnLength; a.qStartPos = qStart; a.qEndPos = f.qEnd; a.qLen = L; a.dbStartPos = dbStart; a.dbEndPos = f.dbEnd;
        a.dbLen = Lt; a.backtraceOff = (uint32_t) s->cigars.size(); a.backtraceLen = (uint32_t) bo.backtrace.size();
        // Alignment::checkCriteria(altRes, false, ...)
        if (!((a.eval <= par.evalThr) && (a.seqId >= par.seqIdThr) && hasCoverage(par.covThr, par.covMode, a.qcov, a.dbcov) && (int) a.alnLength >= par.alnLenThr)) break;
        s->cigars.append(bo.backtrace);
        s->cigars.push_back('\0');
        results[nres++] = a;
        prev = a;
    }
    return FSGPU_OK;
}

// alignStructure gates + backtrace + checkCriteria + ordering for one query (structurealign.cpp:37-112,350-445).
// pre / preIdx: backtraces computed ahead by the worker pool (preIdx[k] = index into pre, -1 = none); without them the
// backtrace of a pair is computed here, when the loop reaches it (the --max-accept / --max-rejected path).
int gateAlign(fshost_search *s, const AlignQuery &aq, int64_t identityId, const uint32_t *targetIds, int n, const fsgpu_swres *fwd,
              const fsgpu_swres *rev, fshost_result *results, double &tBack, const BlockAlnOut *pre, const int *preIdx) {
    const fshost_params &par = s->par;
    const int L = aq.L;
    const int resCap = n * (1 + std::max(0, par.altAlignment));
    int passedNum = 0, rejected = 0, nres = 0;
    BlockAlnOut local;
    for (int k = 0; k < n && passedNum < par.maxAccept && rejected < par.maxRejected; k++) {
        const uint32_t tid = targetIds[k];
        const bool isIdentity = ((int64_t) tid == identityId);
        if (tid >= s->keys.size()) { s->err = "target id out of range"; return FSGPU_E_ARG; }
        const int Lt = s->lengths[tid];
        // ---- alignStructure ----
        const fsgpu_swres &f = fwd[k];
        int32_t score = 0;
        double evalue = 0;
        bool looked = false;
        const bool pass = passesScoreGates(s, aq, tid, f, rev[k], score, evalue, &looked);
        if (looked) s->stats[6] += 1.0;             // pairs whose reversed-query score is actually looked at
        if (!pass) { rejected++; continue; }
        // start position + backtrace on the host (block aligner), only for hits that survived both gates
        const BlockAlnOut *bop;
        if (pre && preIdx && preIdx[k] >= 0) {
            bop = &pre[preIdx[k]];
        } else {
            const double tb0 = nowSec();
            pairBacktrace(s, aq, tid, f, local);
            tBack += nowSec() - tb0;
            bop = &local;
        }
        const BlockAlnOut &bo = *bop;
        fshost_result r;
        memset(&r, 0, sizeof(r));
        int qStart = -1, dbStart = -1;
        float seqId = 0.0f;
        float qCov = computeCov(0, f.qEnd, L), tCov = computeCov(0, f.dbEnd, Lt);
        if (bo.ok) {
            qStart = bo.qStart; dbStart = bo.dbStart;
            qCov = computeCov(qStart, f.qEnd, L);
            tCov = computeCov(dbStart, f.dbEnd, Lt);
        }
        unsigned int alnLength = std::max(abs(f.qEnd - qStart), abs(f.dbEnd - dbStart)) + 1;   // Matcher::computeAlnLength
        if (bo.backtrace.size() > 0) {
            alnLength = bo.backtrace.size();
            // Util::computeSeqId (M/src/commons/Util.cpp:597-607)
            const int den = par.seqIdMode == 1 ? std::min(L, Lt) : par.seqIdMode == 2 ? std::max(L, Lt) : (int) alnLength;
            seqId = static_cast<float>(bo.identicalAA) / static_cast<float>(den);
        }
        r.dbKey = s->keys[tid]; r.score = score; r.qcov = qCov; r.dbcov = tCov; r.seqId = seqId; r.eval = evalue;
        r.alnLength = alnLength; r.qStartPos = qStart; r.qEndPos = f.qEnd; r.qLen = L; r.dbStartPos = dbStart; r.dbEndPos = f.dbEnd;
        r.dbLen = Lt; r.backtraceOff = (uint32_t) s->cigars.size(); r.backtraceLen = (uint32_t) bo.backtrace.size();
        // Alignment::checkCriteria (Alignment.cpp:548)
        const bool evalOk = (r.eval <= par.evalThr);
        const bool seqIdOK = (r.seqId >= par.seqIdThr);
        const bool covOK = hasCoverage(par.covThr, par.covMode, r.qcov, r.dbcov);
        const bool alnLenOK = (int) r.alnLength >= par.alnLenThr;
        if (isIdentity || (evalOk && seqIdOK && covOK && alnLenOK)) {
            s->cigars.append(bo.backtrace);
            s->cigars.push_back('\0');
            results[nres++] = r;
            if (par.altAlignment > 0) {
                const int rc = alternativeAlignments(s, aq, tid, r, results, nres, resCap);
                if (rc != FSGPU_OK) return rc;
            }
            passedNum++;
            rejected = 0;
        } else {
            rejected++;
        }
    }
    if (nres > 1) std::sort(results, results + nres, compareHits);
    return nres;
}

// Backtraces of every pair of the given queries that passes the score gates, computed ahead of gateAlign by the host pool (the calling
// thread takes part) and the device aligner together.  Only when the accept / reject limits cannot cut a query short (their defaults):
// otherwise gateAlign computes them one by one, like the reference, and stops where the reference stops.
struct PreBacktrace {
    std::vector<BlockAlnOut> outs;
    std::vector<std::vector<int>> idx;       // [query][pair] -> outs index or -1
    double seconds = 0;                       // wall time of the parallel section
    size_t onDevice = 0;                      // hits the device aligner answered
};
void precomputeBacktraces(const fshost_search *s, const std::vector<AlignQuery> &aq, const uint32_t *const *targetIds, const int *n,
                          const fsgpu_swres *fwd, const fsgpu_swres *rev, PreBacktrace &pb) {
    const int nq = (int) aq.size();
    pb.idx.assign(nq, {});
    if (s->par.maxAccept != INT_MAX || s->par.maxRejected != INT_MAX) return;
    struct Task { int q, k; size_t base; };
    std::vector<Task> tasks;
    size_t base = 0;
    for (int i = 0; i < nq; i++) {
        pb.idx[i].assign(n[i], -1);
        for (int k = 0; k < n[i]; k++) {
            const uint32_t tid = targetIds[i][k];
            if (tid >= s->keys.size()) continue;
            int32_t score; double evalue;
            if (passesScoreGates(s, aq[i], tid, fwd[base + k], rev[base + k], score, evalue, nullptr)) {
                pb.idx[i][k] = (int) tasks.size();
                tasks.push_back({i, k, base});
            }
        }
        base += (size_t) n[i];
    }
    pb.outs.resize(tasks.size());
    const double t0 = nowSec();
    // ---- who answers a hit.  FSGPU_DEVICE_BACKTRACE = 1 / 0 forces the device aligner (k_btrace.hpp; what it hands back goes to the host) / the host
    // aligner (read per call: the tests switch it inside one process).  2: BOTH, sharing one list -- the pool's threads take hits from its front one at
    // a time, a helper thread feeds the device aligner chunks from its back (each at most half of what is left, at least kDevMin hits: the kernel's
    // latency is that of its longest alignment, ~5-20 ms, whatever the chunk), until the two meet.  (Round 5 chose one side per batch from constants of
    // one box, the first version of round 6 from per-hit costs measured in the process; both sent next to nothing to the device on 16 cores.)
    const char *envDev = getenv("FSGPU_DEVICE_BACKTRACE");
    // Measured (all-vs-all, 606 k accepted hits; profiles/r06_backtrace_modes.txt).  16 cores, 8 feeder threads, module query loop: host only 2.45 s,
    // shared 3.95 s with 294 k hits on the device -- k_block_backtrace keeps every SIMD issuing for the 10-20 ms of a chunk, and the k-mer batches'
    // chains of short kernels and the SW passes of the other feeders queue behind it (their waits doubled).  One rank's 2 cores of an 8-rank node
    // (bench.py --emulate-rank-share 8): host only 19.1 k queries/s, shared 30.1 k, device only 48.3 k -- two cores have no time to give.
    // Hence the default by the cores per GPU: <= 4 -> the device aligner (1), more -> the host aligner (0); 2 = shared on request.
    const int envMode = envDev && *envDev ? atoi(envDev) : -1;
    // cores per GPU: what this process may use divided by the devices it drives; a rank of a multi-process job is told its share (bench.py sets
    // FSGPU_CORES_PER_GPU = usable cores / local ranks: the cgroup quota and the affinity mask are the whole job's)
    const int coresPerGpu = [] {
        const char *e = getenv("FSGPU_CORES_PER_GPU");
        if (e && *e && atoi(e) > 0) return atoi(e);
        return std::max(1, usableCores() / std::max(1, fsgpu_live_devices()));
    }();
    const int mode = envMode == 0 || envMode == 1 || envMode == 2 ? envMode : (coresPerGpu <= 4 ? 1 : 0);          // 0 host, 1 device, 2 shared
    const fshost_params &par = s->par;
    const bool deviceCan = s->dataAA != nullptr && !tasks.empty() && par.gapOpen > par.gapExtend && par.gapExtend >= 1 && par.gapOpen <= 127;
    // hits from which a batch is shared (FSGPU_BT_SHARE_MIN; the tests lower it), smallest / largest device chunk
    const size_t kShareMin = [] { const char *e = getenv("FSGPU_BT_SHARE_MIN"); const long v = e && *e ? atol(e) : 0; return (size_t) (v > 0 ? v : 1024); }();
    const size_t kDevMin = std::min<size_t>(512, kShareMin / 2), kDevMax = 8192;
    // (a handful of hits: the device call's fixed cost alone decides, unless the device was asked for)
    const bool useDevice = deviceCan && ((mode == 1 && (envMode == 1 || tasks.size() >= 64)) || (mode == 2 && tasks.size() >= kShareMin));
    // the shared list: [lo, hi) not taken yet
    std::atomic<uint64_t> range{(uint64_t) tasks.size()};          // lo << 32 | hi
    auto popFront = [&](size_t &t) {
        uint64_t r = range.load(std::memory_order_relaxed);
        for (;;) {
            const uint64_t lo = r >> 32, hi = r & 0xffffffffu;
            if (lo >= hi) return false;
            if (range.compare_exchange_weak(r, ((lo + 1) << 32) | hi, std::memory_order_acq_rel)) { t = (size_t) lo; return true; }
        }
    };
    auto popBack = [&](size_t &c0, size_t &c1, size_t traceBudget) {
        uint64_t r = range.load(std::memory_order_relaxed);
        for (;;) {
            const uint64_t lo = r >> 32, hi = r & 0xffffffffu;
            const uint64_t left = hi > lo ? hi - lo : 0;
            uint64_t take = mode == 1 ? std::min<uint64_t>(left, 16384) : std::min<uint64_t>(kDevMax, left / 2);
            if (take == 0 || (mode == 2 && take < kDevMin)) return false;
            // at most 1 GB of trace scratch per device call (64 B per row and column of an alignment's prefixes: a call full of 2000-residue pairs
            // would otherwise pin 6.7 GB per feeder context for good)
            size_t bytes = 0; uint64_t fit = 0;
            for (uint64_t t = hi; t > hi - take; t--) {
                const fsgpu_swres &f = fwd[tasks[t - 1].base + tasks[t - 1].k];
                const size_t bb = 64 * ((size_t) f.qEnd + (size_t) f.dbEnd + 2 + 256);
                if (fit > 0 && bytes + bb > traceBudget) break;
                bytes += bb; fit++;
            }
            take = fit;
            if (range.compare_exchange_weak(r, (lo << 32) | (hi - take), std::memory_order_acq_rel)) { c0 = (size_t) (hi - take); c1 = (size_t) hi; return true; }
        }
    };
    std::vector<int> handedBack;          // device: status 0 (block beyond its limit, trace slice too small, ...)
    std::mutex handedM;
    std::atomic<size_t> devDone{0};
    std::atomic<bool> devBroken{false};
    auto deviceLoop = [&]() {
        fshost_search *ms = const_cast<fshost_search *>(s);
        std::vector<fsgpu_bt_query> bq(nq);
        for (int i = 0; i < nq; i++) { bq[i].qAA = aq[i].qAA; bq[i].q3Di = aq[i].q3di; bq[i].cbAA = aq[i].cbAA.data(); bq[i].cbSS = aq[i].cbSS.data(); bq[i].L = aq[i].L; bq[i].reserved = 0; }
        std::vector<fsgpu_bt_task> bt;
        std::vector<fsgpu_bt_res> br;
        size_t c0, c1;
        while (!devBroken.load() && popBack(c0, c1, (size_t) 1 << 30)) {
            bt.resize(c1 - c0); br.resize(c1 - c0);
            for (size_t t = c0; t < c1; t++) {
                const Task &tk = tasks[t];
                const fsgpu_swres &f = fwd[tk.base + tk.k];
                bt[t - c0] = fsgpu_bt_task{(uint32_t) tk.q, targetIds[tk.q][tk.k], f.qEnd, f.dbEnd, f.score};
            }
            const char *btBase = nullptr;
            if (fsgpu_block_backtrace(s->ctx, ms->btTblAA.data(), ms->btTbl3.data(), ms->btLetAA.data(), ms->btLet3.data(), bq.data(), nq, bt.data(), (int) bt.size(),
                                      par.gapOpen, par.gapExtend, br.data(), &btBase) != FSGPU_OK) {
                // (scratch did not fit, ...): the host answers this chunk and the device takes no more
                std::lock_guard<std::mutex> g(handedM);
                for (size_t t = c0; t < c1; t++) handedBack.push_back((int) t);
                devBroken.store(true);
                break;
            }
            size_t ok = 0;
            for (size_t t = c0; t < c1; t++) {
                const fsgpu_bt_res &r = br[t - c0];
                BlockAlnOut &o = pb.outs[t];
                o = BlockAlnOut();
                if (r.status == 1) { o.ok = true; o.qStart = r.qStart; o.dbStart = r.dbStart; o.identicalAA = (unsigned int) r.identicalAA; o.backtrace.assign(btBase + r.btOff, (size_t) r.btLen); ok++; }
                else if (r.status == 2) ok++;
                else { std::lock_guard<std::mutex> g(handedM); handedBack.push_back((int) t); }
            }
            devDone.fetch_add(ok);
        }
    };
    if (useDevice) {
        fshost_search *ms = const_cast<fshost_search *>(s);
        if (ms->btTblAA.empty()) {
            // the two AAMatrix tables exactly as blockBacktrace fills them (block_set_aamatrix per letter pair, StructureSmithWaterman.cpp:428-447)
            AAMatrix *ma = block_new_simple_aamatrix(1, -1), *m3 = block_new_simple_aamatrix(1, -1);
            for (int a = 0; a < s->matAA.n; a++)
                for (int b = 0; b < s->matAA.n; b++) block_set_aamatrix(ma, (uint8_t) s->matAA.letters[a], (uint8_t) s->matAA.letters[b], (int8_t) s->matAA.sub[a * s->matAA.n + b]);
            for (int a = 0; a < s->mat3Di.n; a++)
                for (int b = 0; b < s->mat3Di.n; b++) block_set_aamatrix(m3, (uint8_t) s->mat3Di.letters[a], (uint8_t) s->mat3Di.letters[b], (int8_t) s->mat3Di.sub[a * s->mat3Di.n + b]);
            ms->btTblAA.assign(block_aamatrix_scores(ma), block_aamatrix_scores(ma) + 27 * 32);
            ms->btTbl3.assign(block_aamatrix_scores(m3), block_aamatrix_scores(m3) + 27 * 32);
            block_free_aamatrix(ma); block_free_aamatrix(m3);
            auto letterIdx = [](char c) { return (uint8_t) (((c >= 'a' && c <= 'z') ? c - 32 : c) - 'A'); };
            ms->btLetAA.assign(21, 23); ms->btLet3.assign(21, 23);
            for (int a = 0; a < std::min(21, s->matAA.n); a++) ms->btLetAA[a] = letterIdx(s->matAA.letters[a]);
            for (int a = 0; a < std::min(21, s->mat3Di.n); a++) ms->btLet3[a] = letterIdx(s->mat3Di.letters[a]);
        }
    }
    auto hostOne = [&](size_t t) {
        const Task &tk = tasks[t];
        pairBacktrace(s, aq[tk.q], targetIds[tk.q][tk.k], fwd[tk.base + tk.k], pb.outs[t]);
    };
    if (useDevice && mode == 1) {
        deviceLoop();                          // everything the device takes; what it cannot take or hands back follows below
    } else if (useDevice) {
        // one batch at a time shares its list with the device (the feeders' batches overlap: eight aligner calls in flight were what doubled the other
        // kernels' waits), and with ONE workgroup per CU: the aligner's workgroups live for the whole call and hold 48 KB of LDS each
        static std::mutex deviceAlignerBusy;
        std::unique_lock<std::mutex> mine(deviceAlignerBusy, std::try_to_lock);
        std::thread helper;
        if (mine.owns_lock()) {
            static const int fp = [] { const char *e = getenv("FSGPU_BT_SHARED_WG_PER_CU"); const int v = e && *e ? atoi(e) : 1; return v >= 0 && v <= 16 ? v : 1; }();
            (void) fsgpu_block_backtrace_footprint(s->ctx, fp);
            helper = std::thread(deviceLoop);  // sleeps in the device call's stream wait almost all of its time
        }
        const int slots = HostPool::get().workers() + 1;
        HostPool::get().parallelFor(slots, [&](int) { size_t t; while (popFront(t)) hostOne(t); });
        if (helper.joinable()) { helper.join(); (void) fsgpu_block_backtrace_footprint(s->ctx, 0); }
    }
    {
        // the host's part where nothing was shared, plus whatever the device left or handed back
        size_t t;
        while (popFront(t)) handedBack.push_back((int) t);
        HostPool::get().parallelFor((int) handedBack.size(), [&](int h) { hostOne((size_t) handedBack[h]); });
    }
    pb.onDevice = devDone.load();
    pb.seconds = nowSec() - t0;
}

} // namespace

extern "C" {

int fshost_search_align(fshost_search *s, const uint8_t *qAA, const uint8_t *q3di, int L, int64_t identityId,
                        const uint32_t *targetIds, int n, fshost_result *results) {
    if (!s || !s->ctx || !qAA || !q3di || L <= 0 || n < 0) return FSGPU_E_ARG;
    const fshost_params &par = s->par;
    const bool useAA = par.alignmentType == 2;
    const double t0 = nowSec();
    AlignQuery aq;
    aq.qAA = qAA; aq.q3di = q3di; aq.L = L;
    int rc = prepareAlign(s, aq, s->rAA, s->r3Di);
    if (rc != FSGPU_OK) return rc;
    s->fwd.resize(n); s->rev.resize(n);
    const double t1 = nowSec();
    rc = fsgpu_sw_batch(s->ctx, useAA ? aq.pAAf.data() : nullptr, aq.p3f.data(), useAA ? aq.pAAr.data() : nullptr, aq.p3r.data(), L,
                        targetIds, n, par.gapOpen, par.gapExtend, s->fwd.data(), s->rev.data());
    if (rc != FSGPU_OK) { s->err = fsgpu_last_error(s->ctx); return rc; }
    const double t2 = nowSec();
    double tBack = 0;
    s->cigars.clear();
    PreBacktrace pb;
    {
        std::vector<AlignQuery> one(1);
        one[0] = std::move(aq);
        precomputeBacktraces(s, one, &targetIds, &n, s->fwd.data(), s->rev.data(), pb);
        aq = std::move(one[0]);
    }
    tBack += pb.seconds;
    const int nres = gateAlign(s, aq, identityId, targetIds, n, s->fwd.data(), s->rev.data(), results, tBack, pb.outs.data(), pb.idx[0].empty() ? nullptr : pb.idx[0].data());
    s->stats[2] = t1 - t0; s->stats[3] = t2 - t1; s->stats[5] = tBack; s->stats[4] = nowSec() - t2 - tBack;
    return nres;
}

// The same for nq queries with one forward and one reversed device pass (fsgpu_sw_multi_dir): results[q] must hold n[q] entries, nres[q] receives the
// number of accepted alignments of query q.  Backtraces of all queries stay valid until the next align call.
int fshost_search_align_batch(fshost_search *s, int nq, const uint8_t *const *qAA, const uint8_t *const *q3di, const int *L,
                              const int64_t *identityId, const uint32_t *const *targetIds, const int *n,
                              fshost_result *const *results, int *nres) {
    if (!s || !s->ctx || nq < 0 || (nq > 0 && (!qAA || !q3di || !L || !targetIds || !n || !results || !nres))) return FSGPU_E_ARG;
    const fshost_params &par = s->par;
    const bool useAA = par.alignmentType == 2;
    const double t0 = nowSec();
    std::vector<AlignQuery> aq(nq);
    std::vector<fsgpu_sw_cquery> dq(nq);
    size_t total = 0;
    for (int i = 0; i < nq; i++) if (!qAA[i] || !q3di[i] || L[i] <= 0 || n[i] < 0) return FSGPU_E_ARG;
    {
        // e-value network + the two composition-bias passes of every query: 25-50 us each, 1024 of them per all-vs-all batch -- over the pool, in
        // slices of 16 queries (round 5 ran them one after the other on the feeder thread: "profiles" in the module timing)
        std::atomic<int> firstErr{FSGPU_OK};
        std::mutex errM;
        const int slice = 16, nSlices = (nq + slice - 1) / slice;
        HostPool::get().parallelFor(nSlices, [&](int sl) {
            static thread_local std::vector<uint8_t> rAA, r3Di;
            for (int i = sl * slice; i < std::min(nq, (sl + 1) * slice); i++) {
                aq[i].qAA = qAA[i]; aq[i].q3di = q3di[i]; aq[i].L = L[i];
                std::string err;
                const int rc = prepareAlign(s, aq[i], rAA, r3Di, par.altAlignment > 0, &err);      // --alt-ali re-aligns through the profile-based call
                if (rc != FSGPU_OK) { std::lock_guard<std::mutex> g(errM); if (firstErr.load() == FSGPU_OK) { firstErr.store(rc); s->err = err; } }
            }
        });
        if (firstErr.load() != FSGPU_OK) return firstErr.load();
    }
    for (int i = 0; i < nq; i++) {
        dq[i].qAA = qAA[i]; dq[i].q3Di = q3di[i];
        dq[i].cbAA_fwd = aq[i].cbAA.data(); dq[i].cb3Di_fwd = aq[i].cbSS.data(); dq[i].cbAA_rev = aq[i].cbAAr.data(); dq[i].cb3Di_rev = aq[i].cbSSr.data();
        dq[i].L = L[i]; dq[i].n = n[i]; dq[i].targetIds = targetIds[i];
        total += (size_t) n[i];
    }
    const int8_t *m3 = s->mat3Di.tiny.data(), *mA = useAA ? s->matAA.tiny.data() : nullptr;
    // FSGPU_SW_PROFILES=1: the profile-based device entry (word profiles built here, LDS images built on the host, k_sw2) instead of the
    // compact one -- kept for A/B measurements of the two device paths (tools/sw2_probe.py)
    static const bool viaProfiles = [] { const char *e = getenv("FSGPU_SW_PROFILES"); return e && *e && *e != '0'; }();
    std::vector<fsgpu_sw_query> pq;
    if (viaProfiles) {
        pq.resize(nq);
        for (int i = 0; i < nq; i++) {
            materializeProfiles(s, aq[i]);
            pq[i].pAA_fwd = useAA ? aq[i].pAAf.data() : nullptr; pq[i].p3Di_fwd = aq[i].p3f.data();
            pq[i].pAA_rev = useAA ? aq[i].pAAr.data() : nullptr; pq[i].p3Di_rev = aq[i].p3r.data();
            pq[i].L = L[i]; pq[i].n = n[i]; pq[i].targetIds = targetIds[i];
        }
    }
    auto pass = [&](int dir, const int32_t *const *sel, const int32_t *nsel, fsgpu_swres *out) {
        return viaProfiles ? fsgpu_sw_multi_dir(s->ctx, pq.data(), nq, par.gapOpen, par.gapExtend, dir, sel, nsel, out)
                           : fsgpu_sw_multi_dir_c(s->ctx, m3, mA, dq.data(), nq, par.gapOpen, par.gapExtend, dir, sel, nsel, out);
    };
    s->fwd.resize(total); s->rev.assign(total, fsgpu_swres{0, 0, 0, 0});
    const double t1 = nowSec();
    // A small batch (the all-vs-all steps: a few thousand pairs per call) is bound by round trips, not by DP cells: both directions of
    // every pair in one submission.  Otherwise: forward pass over every pair, the reversed-query pass only over the pairs whose forward
    // score passes the gates (3-16 % of a search's hit lists).  FSGPU_SW_ONEPASS_PAIRS overrides the limit (0 = never).
    static const size_t onePassPairs = [] { const char *e = getenv("FSGPU_SW_ONEPASS_PAIRS"); return e ? (size_t) atoll(e) : (size_t) 16384; }();
    int rc;
    if (!viaProfiles && total > 0 && total <= onePassPairs) {
        rc = fsgpu_sw_multi_c(s->ctx, m3, mA, dq.data(), nq, par.gapOpen, par.gapExtend, s->fwd.data(), s->rev.data());
        if (rc != FSGPU_OK) { s->err = fsgpu_last_error(s->ctx); return rc; }
        size_t b = 0, any = 0;
        for (int i = 0; i < nq; i++) {
            for (int k = 0; k < n[i]; k++) {
                const uint32_t tid = targetIds[i][k];
                if (tid >= s->keys.size()) { s->err = "target id out of range"; return FSGPU_E_ARG; }
                if (needsReversePass(s, aq[i], tid, s->fwd[b + k])) any++;
                else s->rev[b + k] = fsgpu_swres{0, 0, 0, 0};          // what the two-pass form leaves for pairs structurealign never reverses
            }
            b += (size_t) n[i];
        }
        s->stats[7] = (double) any;
    } else {
    rc = pass(0, nullptr, nullptr, s->fwd.data());
    if (rc != FSGPU_OK) { s->err = fsgpu_last_error(s->ctx); return rc; }
    {
        std::vector<std::vector<int32_t>> sel(nq);
        std::vector<const int32_t *> selp(nq);
        std::vector<int32_t> nsel(nq);
        size_t b = 0, any = 0;
        for (int i = 0; i < nq; i++) {
            for (int k = 0; k < n[i]; k++) {
                const uint32_t tid = targetIds[i][k];
                if (tid >= s->keys.size()) { s->err = "target id out of range"; return FSGPU_E_ARG; }
                if (needsReversePass(s, aq[i], tid, s->fwd[b + k])) sel[i].push_back(k);
            }
            selp[i] = sel[i].data(); nsel[i] = (int32_t) sel[i].size(); any += sel[i].size();
            b += (size_t) n[i];
        }
        s->stats[7] = (double) any;
        if (any) {
            rc = pass(1, selp.data(), nsel.data(), s->rev.data());
            if (rc != FSGPU_OK) { s->err = fsgpu_last_error(s->ctx); return rc; }
        }
    }
    }
    const double t2 = nowSec();
    double tBack = 0;
    s->cigars.clear();
    PreBacktrace pb;
    precomputeBacktraces(s, aq, targetIds, n, s->fwd.data(), s->rev.data(), pb);
    tBack += pb.seconds;
    size_t base = 0;
    for (int i = 0; i < nq; i++) {
        nres[i] = gateAlign(s, aq[i], identityId ? identityId[i] : -1, targetIds[i], n[i], s->fwd.data() + base, s->rev.data() + base, results[i], tBack,
                            pb.outs.data(), pb.idx[i].empty() ? nullptr : pb.idx[i].data());
        if (nres[i] < 0) return nres[i];
        base += (size_t) n[i];
    }
    s->stats[2] = t1 - t0; s->stats[3] = t2 - t1; s->stats[5] = tBack; s->stats[4] = nowSec() - t2 - tBack;
    s->lastDeviceBt = (int64_t) pb.onDevice; s->lastBtTasks = (int64_t) pb.outs.size();
    return FSGPU_OK;
}

// structurerescorediagonal's per-query body (F/src/strucclustutils/structurerescorediagonal.cpp:50-156, 300-370) for nq queries:
// the Kadane scans run on the device (fsgpu_diag_rescore), here are the module's gates in the module's order.
// diagonals[q][k] is the prefilter line's third column (a short, as parsePrefilterHit reads it).
// Util::canBeCovered (M/src/commons/Util.cpp:542-559) for the three modes runSplit applies it to (Prefiltering.cpp:880-887)
static bool kmerCanBeCovered(float covThr, int covMode, float q, float t) {
    switch (covMode) {
        case 0: return (q / t >= covThr) && (t / q >= covThr);
        case 2: return (t / q) >= covThr;
        case 5: return (std::min(t, q) / std::max(t, q)) >= covThr;
        default: return true;
    }
}

int fshost_search_kmer_batch(fshost_search *s, const fshost_matrix *mKmer, const fshost_matrix *mUngapped, const fsgpu_kmer_search_params *sp,
                             int kmerThr, int spaced, int nq, const uint8_t *const *qAA, const uint8_t *const *q3di, const int *L,
                             const int64_t *prefIdentity, const int64_t *alnIdentity, fsgpu_kmer_hit *hits, int32_t *nhits, int32_t *status,
                             uint32_t *keptIds, int32_t *nkept, fshost_result *results, int32_t *nres, double *seconds) {
    if (!s || !s->ctx || !mKmer || !mUngapped || !sp || nq < 0 || (nq > 0 && (!qAA || !q3di || !L || !hits || !nhits || !status || !keptIds || !nkept || !results || !nres))) return FSGPU_E_ARG;
    if (sp->maxResListLen <= 0) { s->err = "fshost_search_kmer_batch: maxResListLen >= 1 required"; return FSGPU_E_ARG; }
    const fshost_params &par = s->par;
    const size_t cap = (size_t) sp->maxResListLen, rcap = cap * (size_t) (1 + std::max(0, par.altAlignment));
    double t0 = nowSec();
    if ((int) s->kThr.size() < nq) { s->kThr.resize(nq); s->kProf.resize(nq); }
    s->kq.resize(std::max(nq, 1));
    for (int k = 0; k < nq; k++) {
        if (L[k] < 0 || (L[k] > 0 && (!q3di[k] || !qAA[k]))) { s->err = "fshost_search_kmer_batch: bad query"; return FSGPU_E_ARG; }
        s->kThr[k].resize((size_t) L[k] + 1); s->kProf[k].resize((size_t) L[k] * 21 + 1);
        fshost_kmer_query_prepare(mKmer, mUngapped, q3di[k], L[k], par.compBiasCorrection, par.prefCompBiasScale, kmerThr, 6, spaced, s->kThr[k].data(), s->kProf[k].data());
        fsgpu_kmer_query &q = s->kq[k];
        q.seq = q3di[k]; q.kmerThr = s->kThr[k].data(); q.profile = s->kProf[k].data(); q.L = L[k]; q.reserved = 0; q.identity = prefIdentity ? prefIdentity[k] : -1;
    }
    double t1 = nowSec();
    if (seconds) seconds[0] = t1 - t0;
    t0 = t1;
    if (nq > 0) {
        const int rc = fsgpu_kmer_search(s->ctx, sp, s->kq.data(), nq, hits, nhits, status, nullptr);
        if (rc != FSGPU_OK) { s->err = fsgpu_last_error(s->ctx); return rc; }
    }
    t1 = nowSec();
    if (seconds) seconds[1] = t1 - t0;
    t0 = t1;
    const bool cov = par.covThr > 0.0 && (par.covMode == 0 || par.covMode == 2 || par.covMode == 5);
    std::vector<const uint8_t *> lA, l3;
    std::vector<const uint32_t *> lT;
    std::vector<fshost_result *> lR;
    std::vector<int> lL, lN, lres, lq;
    std::vector<int64_t> lI;
    for (int k = 0; k < nq; k++) {
        nres[k] = 0; nkept[k] = 0;
        if (status[k] < 0) continue;
        uint32_t *ids = keptIds + (size_t) k * cap;
        int n = 0;
        for (int h = 0; h < nhits[k]; h++) {
            const fsgpu_kmer_hit &hit = hits[(size_t) k * cap + h];
            if (cov && !kmerCanBeCovered((float) par.covThr, par.covMode, (float) L[k], (float) s->lengths[hit.id])) continue;
            ids[n++] = hit.id;
        }
        nkept[k] = n;
        if (n == 0 || L[k] <= 0) continue;
        lq.push_back(k); lA.push_back(qAA[k]); l3.push_back(q3di[k]); lT.push_back(ids); lR.push_back(results + (size_t) k * rcap);
        lL.push_back(L[k]); lN.push_back(n); lI.push_back(alnIdentity ? alnIdentity[k] : -1);
    }
    t1 = nowSec();
    if (seconds) seconds[2] = t1 - t0;
    t0 = t1;
    if (!lq.empty()) {
        lres.assign(lq.size(), 0);
        const int rc = fshost_search_align_batch(s, (int) lq.size(), lA.data(), l3.data(), lL.data(), lI.data(), lT.data(), lN.data(), lR.data(), lres.data());
        if (rc != FSGPU_OK) return rc;
        for (size_t j = 0; j < lq.size(); j++) nres[lq[j]] = lres[j];
    }
    if (seconds) seconds[3] = nowSec() - t0;
    return FSGPU_OK;
}

int fshost_search_rescore_diagonal_batch(fshost_search *s, int nq, const uint8_t *const *qAA, const uint8_t *const *q3di, const int *L,
                                         const int64_t *identityId, const uint32_t *const *targetIds, const int16_t *const *diagonals,
                                         const int *n, fshost_result *const *results, int *nres) {
    if (!s || !s->ctx || nq < 0 || (nq > 0 && (!qAA || !q3di || !L || !targetIds || !diagonals || !n || !results || !nres))) return FSGPU_E_ARG;
    const fshost_params &par = s->par;
    std::vector<uint8_t> catA, cat3;
    std::vector<uint64_t> qOff(nq + 1, 0);
    std::vector<int32_t> qLen(nq);
    std::vector<fsgpu_diag_pair> pairs;
    std::vector<size_t> base(nq + 1, 0);
    for (int i = 0; i < nq; i++) {
        if (!qAA[i] || !q3di[i] || L[i] <= 0 || n[i] < 0) return FSGPU_E_ARG;
        qLen[i] = L[i];
        qOff[i + 1] = qOff[i] + (uint64_t) ((L[i] + 3) / 4 * 4);
        catA.resize(qOff[i + 1], 20); cat3.resize(qOff[i + 1], 20);
        memcpy(catA.data() + qOff[i], qAA[i], L[i]); memcpy(cat3.data() + qOff[i], q3di[i], L[i]);
        for (int k = 0; k < n[i]; k++) {
            if (targetIds[i][k] >= s->keys.size()) { s->err = "target id out of range"; return FSGPU_E_ARG; }
            pairs.push_back({(uint32_t) i, targetIds[i][k], (int32_t) diagonals[i][k]});
        }
        base[i + 1] = pairs.size();
    }
    std::vector<fsgpu_diag_res> dr(pairs.size());
    const double t1 = nowSec();
    if (!pairs.empty()) {
        const int rc = fsgpu_diag_rescore(s->ctx, catA.data(), cat3.data(), qOff.data(), qLen.data(), nq, s->mat3Di.sub.data(), s->matAA.sub.data(),
                                          pairs.data(), (int64_t) pairs.size(), dr.data());
        if (rc != FSGPU_OK) { s->err = fsgpu_last_error(s->ctx); return rc; }
    }
    const double t2 = nowSec();
    s->cigars.clear();
    for (int i = 0; i < nq; i++) {
        double lambda = 0, mu = 0;
        s->evaluer.predictMuLambda(q3di[i], L[i], s->mat3Di.n, &lambda, &mu);
        const int Lq = L[i];
        int passedNum = 0, rejected = 0, cnt = 0;
        for (int k = 0; k < n[i] && passedNum < par.maxAccept && rejected < par.maxRejected; k++) {
            const uint32_t tid = targetIds[i][k];
            const bool isIdentity = identityId && ((int64_t) tid == identityId[i]);
            const int Lt = s->lengths[tid];
            if (!canBeCovered(par.covThr, par.covMode, (float) Lq, (float) Lt)) { rejected++; continue; }
            const fsgpu_diag_res &d = dr[base[i] + k];
            if (d.status == FSGPU_DIAG_UNDEFINED || d.status == FSGPU_DIAG_NO_OVERLAP) {
                if (par.skipUndefinedDiagonals) continue;
                s->err = "structurerescorediagonal: query " + std::to_string(i) + " target key " + std::to_string(s->keys[tid]) + " diagonal " + std::to_string(diagonals[i][k]) +
                         (d.status == FSGPU_DIAG_UNDEFINED ? ": negative diagonal with a target longer than the query -- the reference's reverse pass reads past the query here "
                                                             "(structurerescorediagonal.cpp:96-99), its result is undefined"
                                                           : ": diagonal outside the sequences (the reference indexes position -1 here)");
                return FSGPU_E_UNSUPPORTED;
            }
            const int diagonal = diagonals[i][k];
            const int dist = abs(diagonal);
            const int32_t score = d.score - d.revScore;
            const double evalue = s->evaluer.computeEvalueCorr(score, lambda, mu);
            int qStart, qEnd, dbStart, dbEnd;
            if (diagonal >= 0) { qStart = d.startPos + dist; qEnd = d.endPos + dist; dbStart = d.startPos; dbEnd = d.endPos; }
            else { qStart = d.startPos; qEnd = d.endPos; dbStart = d.startPos + dist; dbEnd = d.endPos + dist; }
            const unsigned int alnLength = std::max(abs(qEnd - qStart), abs(dbEnd - dbStart)) + 1;
            const float queryCov = computeCov(qStart, qEnd, Lq), targetCov = computeCov(dbStart, dbEnd, Lt);
            if (!hasCoverage(par.covThr, par.covMode, queryCov, targetCov)) { rejected++; continue; }
            if (evalue > par.evalThr) { rejected++; continue; }
            const int den = par.seqIdMode == 1 ? std::min(Lq, Lt) : par.seqIdMode == 2 ? std::max(Lq, Lt) : (int) alnLength;
            fshost_result r;
            memset(&r, 0, sizeof(r));
            r.dbKey = s->keys[tid]; r.score = score; r.qcov = queryCov; r.dbcov = targetCov;
            r.seqId = static_cast<float>(d.identicalAA) / static_cast<float>(den);
            r.eval = evalue; r.alnLength = alnLength; r.qStartPos = qStart; r.qEndPos = qEnd; r.qLen = Lq;
            r.dbStartPos = dbStart; r.dbEndPos = dbEnd; r.dbLen = Lt;
            r.backtraceOff = (uint32_t) s->cigars.size(); r.backtraceLen = par.addBacktrace ? alnLength : 0;
            const bool ok = (r.eval <= par.evalThr) && (r.seqId >= par.seqIdThr) && hasCoverage(par.covThr, par.covMode, r.qcov, r.dbcov) && (int) r.alnLength >= par.alnLenThr;
            if (isIdentity || ok) {
                if (par.addBacktrace) s->cigars.append(alnLength, 'M');
                s->cigars.push_back('\0');
                results[i][cnt++] = r;
                passedNum++;
                rejected = 0;
            } else {
                rejected++;
            }
        }
        if (cnt > 1) std::sort(results[i], results[i] + cnt, compareHits);
        nres[i] = cnt;
    }
    s->stats[3] = t2 - t1; s->stats[4] = nowSec() - t2;
    return FSGPU_OK;
}

// alignStartPosBacktrace for a sequence query (StructureSmithWaterman.cpp:540-739): start position by the
// reverse pass on the device, CIGAR by the banded DP on the host (banded_backtrace.cpp).  (qEnd, dbEnd, score) = the forward
// alignScoreEndPos result of the pair.  Returns 1 and fills the outputs; 0 when the reverse pass does not reproduce `score`
// (the reference prints "Score of forward/backward SW differ" and exits) or the trace-back fails; < 0 on errors.
int fshost_search_startpos_backtrace(fshost_search *s, const uint8_t *qAA, const uint8_t *q3di, int L, uint32_t targetId, int qEnd, int dbEnd,
                                     int score, int *qStart, int *dbStart, unsigned int *identicalAA, char *backtrace, size_t btCap) {
    if (!s || !s->ctx || !qAA || !q3di || L <= 0 || !qStart || !dbStart || !backtrace || btCap == 0) return FSGPU_E_ARG;
    if (targetId >= s->keys.size() || qEnd < 0 || qEnd >= L || dbEnd < 0 || dbEnd >= s->lengths[targetId]) { s->err = "startpos_backtrace: bad end position"; return FSGPU_E_ARG; }
    const fshost_params &par = s->par;
    const bool useAA = par.alignmentType == 2;
    const int A = s->mat3Di.n;
    AlignQuery aq;
    aq.qAA = qAA; aq.q3di = q3di; aq.L = L;
    int rc = prepareAlign(s, aq, s->rAA, s->r3Di);                 // forward profiles give the rounded biases cbAA / cbSS
    if (rc != FSGPU_OK) return rc;
    // reversed query prefix [0, qEnd] with its position biases (createQueryProfile on query_*_rev_sequence + queryOffset, :617-620)
    const int Lp = qEnd + 1, Lt = dbEnd + 1;
    std::vector<int16_t> pA((size_t) A * Lp), p3((size_t) A * Lp);
    for (int a = 0; a < A; a++)
        for (int i = 0; i < Lp; i++) {
            const int src = qEnd - i;
            pA[(size_t) a * Lp + i] = (int16_t) (s->matAA.tiny[(size_t) a * A + qAA[src]] + aq.cbAA[src]);
            p3[(size_t) a * Lp + i] = (int16_t) (s->mat3Di.tiny[(size_t) a * A + q3di[src]] + aq.cbSS[src]);
        }
    std::vector<uint8_t> tA(Lt), t3(Lt), fA(Lt), f3(Lt);
    for (int k = 0; k < Lt; k++) {
        uint8_t c = s->data3di[s->offsets[targetId] + k]; c = c >= 32 ? c - 32 : c; f3[k] = c > 20 ? 20 : c;
        uint8_t a = s->dataAA ? s->dataAA[s->offsets[targetId] + k] : 20; a = a >= 32 ? a - 32 : a; fA[k] = a > 20 ? 20 : a;
    }
    for (int k = 0; k < Lt; k++) { t3[k] = f3[dbEnd - k]; tA[k] = fA[dbEnd - k]; }
    const uint64_t offs[2] = {0, (uint64_t) Lt};
    const int32_t lens[1] = {Lt};
    fsgpu_swres f, r;
    rc = fsgpu_sw_batch_seqs(s->ctx, useAA ? pA.data() : nullptr, p3.data(), useAA ? pA.data() : nullptr, p3.data(), Lp, tA.data(), t3.data(), offs, lens, 1,
                             par.gapOpen, par.gapExtend, &f, &r);
    if (rc != FSGPU_OK) { s->err = fsgpu_last_error(s->ctx); return rc; }
    if (f.score != score) return 0;
    const int qs = qEnd - f.qEnd, ds = dbEnd - f.dbEnd;
    *qStart = qs; *dbStart = ds;
    std::string path;
    const int qLen = qEnd - qs + 1, dbLen = dbEnd - ds + 1;
    if (!bandedBacktrace(s->matAA, s->mat3Di, qAA + qs, q3di + qs, aq.cbAA.data() + qs, aq.cbSS.data() + qs, qLen, fA.data() + ds, f3.data() + ds, dbLen,
                         score, par.gapOpen, par.gapExtend, path))
        return 0;
    // computerBacktrace (:746-773)
    unsigned int ids = 0;
    int qp = qs, tp = ds;
    for (char c : path) {
        if (c == 'M') { ids += fA[tp] == qAA[qp]; qp++; tp++; }
        else if (c == 'I') qp++;
        else tp++;
    }
    if (identicalAA) *identicalAA = ids;
    if (path.size() + 1 > btCap) { s->err = "startpos_backtrace: backtrace buffer too small"; return FSGPU_E_ARG; }
    memcpy(backtrace, path.c_str(), path.size() + 1);
    return 1;
}

const char *fshost_search_backtrace(const fshost_search *s, const fshost_result *r) { return s->cigars.c_str() + r->backtraceOff; }

void fshost_search_backtrace_counts(const fshost_search *s, int64_t *onDevice, int64_t *all) {
    if (onDevice) *onDevice = s ? s->lastDeviceBt : 0;
    if (all) *all = s ? s->lastBtTasks : 0;
}
void fshost_search_stats(const fshost_search *s, double *out8) { memcpy(out8, s->stats, sizeof(s->stats)); }

void fshost_search_last_sw(const fshost_search *s, const fsgpu_swres **fwd, const fsgpu_swres **rev) {
    *fwd = s->fwd.data();
    *rev = s->rev.data();
}

// ---- text formats --------------------------------------------------------------------------------------------
static char *putU64(char *p, uint64_t v) {
    char tmp[24];
    int k = 0;
    do { tmp[k++] = (char) ('0' + v % 10); v /= 10; } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}
static char *putI32(char *p, int32_t v) {
    if (v < 0) { *p++ = '-'; return putU64(p, (uint64_t) (-(int64_t) v)); }
    return putU64(p, (uint64_t) v);
}

size_t fshost_format_prefilter_hit(char *buf, uint32_t key, int score, int diagonal) {
    char *p = putU64(buf, key);
    *p++ = '\t';
    p = putI32(p, score);
    *p++ = '\t';
    p = putI32(p, (int32_t) (short) diagonal);
    *p++ = '\n';
    *p = '\0';
    return (size_t) (p - buf);
}

size_t fshost_format_result(char *buf, const fshost_result *r, const char *backtrace, int addBacktrace) {
    char *p = putU64(buf, r->dbKey);
    *p++ = '\t';
    p = putI32(p, r->score);
    *p++ = '\t';
    // Util::fastSeqIdToBuffer (Util.cpp:251-279)
    if (r->seqId == 1.0) {
        // the reference prints "1.00": fastSeqIdToBuffer writes "1.000\0" but, unlike the Itoa routines, returns the address
        // OF the terminator, so resultToBuffer's `*(tmpBuff-1) = '\t'` lands on the last zero (Util.cpp:252-263, Matcher.cpp:288-289)
        memcpy(p, "1.00", 4); p += 4;
    } else {
        *p++ = '0'; *p++ = '.';
        if (r->seqId < 0.10) *p++ = '0';
        if (r->seqId < 0.01) *p++ = '0';
        p = putI32(p, (int) (r->seqId * 1000));
    }
    *p++ = '\t';
    p += snprintf(p, 32, "%.3E", r->eval);
    *p++ = '\t';
    p = putI32(p, r->qStartPos); *p++ = '\t';
    p = putI32(p, r->qEndPos); *p++ = '\t';
    p = putI32(p, (int32_t) r->qLen); *p++ = '\t';
    p = putI32(p, r->dbStartPos); *p++ = '\t';
    p = putI32(p, r->dbEndPos); *p++ = '\t';
    p = putI32(p, (int32_t) r->dbLen);
    if (addBacktrace) {
        *p++ = '\t';
        // Matcher::compressAlignment (Matcher.cpp:168-186): run-length encode, first state 'M'
        const char *bt = backtrace ? backtrace : "";
        char state = 'M';
        size_t counter = 0;
        for (size_t i = 0; bt[i]; ++i) {
            if (bt[i] != state) {
                p = putU64(p, counter);
                *p++ = state;
                state = bt[i];
                counter = 1;
            } else {
                counter++;
            }
        }
        p = putU64(p, counter);
        *p++ = state;
    }
    *p++ = '\n';
    *p = '\0';
    return (size_t) (p - buf);
}

} // extern "C"
