// marv_shim.cpp -- `class Marv` (include/marv.h == M/lib/libmarv/src/marv.h:6-58) implemented over the fsgpu_* C ABI.
// Compiled by oracle/build_ref_full.sh into the reference binary in place of libmarv; not part of libfsgpu.so (it would
// drag a C++ class ABI across the C boundary) -- a maintainer who wants `--gpu 1` on AMD adds this one file.
#include "marv.h"
#include "fsgpu.h"
#include "fshost.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

struct ShimDb {                       // what loadDb hands back: borrowed host pointers (the caller keeps the mmap alive,
    const uint8_t *data;              // ungappedprefilter.cpp:124-158) + whether it is resident on the device already
    std::vector<uint64_t> offsets;
    const int32_t *lengths;
    size_t n, bytes;
};

// libmarv uses every visible device inside ONE Marv object and shards the TARGETS over them (marv.cu:57-60,
// cudasw4.cuh:1477-1553), because its caller hands it one query at a time.  Same here: device k holds the targets
// k, k + N, k + 2N, ... of the (length-sorted) database -- every shard sees the whole length range --, a scan runs on all
// devices at once (asynchronous halves of the C ABI, one host thread) and the per-device top lists are merged in the CPU path's
// order.  FSGPU_MARV_SHARDS=n puts n shards on device 0 (tests on a one-GPU box).
struct Shard {
    fsgpu_ctx *ctx = nullptr;
    std::vector<uint64_t> offsets;        // of this shard's targets in the shard's own packed buffer
    std::vector<int32_t> lengths;
    std::vector<uint32_t> globalId;
    std::vector<fsgpu_hit> hits;
};

struct ShimState {
    std::vector<Shard> shards;
    const ShimDb *resident = nullptr;
    size_t maxSeqs = 0;
    int maxSeqLength = 0;
    std::vector<fsgpu_hit> merged;
    std::chrono::steady_clock::time_point timer;
};

[[noreturn]] void die(const std::string &msg) {          // libmarv aborts on device errors (CUERR); so does the shim
    fprintf(stderr, "Marv (fsgpu): %s\n", msg.c_str());
    exit(EXIT_FAILURE);
}

} // namespace

Marv::Marv(size_t dbEntries, int alphabetSize, int maxSeqLength, size_t maxSeqs, Marv::AlignmentType alignmentType)
    : dbEntries(dbEntries), alphabetSize(alphabetSize), cudasw(nullptr), dbmanager(nullptr), alignmentType(alignmentType) {
    if (alignmentType != AlignmentType::GAPLESS)
        die("only AlignmentType::GAPLESS (--prefilter-mode 1 / --gpu 1 without gapped rescoring) is implemented on this device path");
    if (alphabetSize != FSGPU_ALPHABET) die("alphabet size " + std::to_string(alphabetSize) + " is not supported (21: amino acids / 3Di + X)");
    ShimState *s = new ShimState();
    s->maxSeqs = maxSeqs;
    s->maxSeqLength = maxSeqLength;
    std::vector<int> devs = getDeviceIds();
    if (devs.empty()) die("no GPU visible");
    if (const char *e = getenv("FSGPU_MARV_SHARDS")) devs.assign((size_t) std::max(1, atoi(e)), devs[0]);
    if (devs.size() > dbEntries && dbEntries > 0) devs.resize(dbEntries);
    s->shards.resize(devs.size());
    for (size_t k = 0; k < devs.size(); k++) {
        if (fsgpu_create(devs[k], &s->shards[k].ctx) != FSGPU_OK) die(fsgpu_last_error(nullptr));
        s->shards[k].hits.resize(std::max<size_t>(maxSeqs, 1));
    }
    cudasw = s;
}

Marv::~Marv() {
    ShimState *s = static_cast<ShimState *>(cudasw);
    if (s) { for (Shard &sh : s->shards) fsgpu_destroy(sh.ctx); delete s; }
}

std::vector<int> Marv::getDeviceIds() {
    std::vector<int> ids;
    for (int i = 0; i < fsgpu_device_count(); i++) ids.push_back(i);
    return ids;
}

void *Marv::loadDb(char *data, size_t *offset, int32_t *length, size_t dbByteSize) {
    ShimDb *db = new ShimDb();            // never freed by the reference either (marv.cu:95-99)
    db->data = reinterpret_cast<const uint8_t *>(data);
    db->offsets.assign(offset, offset + dbEntries + 1);
    db->offsets[dbEntries] = dbByteSize;  // the reference's last offset is offsets[n-1] + lengths[n-1]: unpadded; ours = end of the buffer
    db->lengths = length;
    db->n = dbEntries;
    db->bytes = dbByteSize;
    return db;
}

void *Marv::loadDb(char *, size_t, void *otherdb) { return otherdb; }

void Marv::setDb(void *dbhandle) {
    ShimState *s = static_cast<ShimState *>(cudasw);
    const ShimDb *db = static_cast<const ShimDb *>(dbhandle);
    if (!db) die("setDb: null database handle");
    if (s->resident != db) {
        const size_t N = s->shards.size();
        std::vector<uint8_t> packed;              // one shard's entries, contiguous, offsets rebased: a device holds ITS targets only
        for (size_t k = 0; k < N; k++) {
            Shard &sh = s->shards[k];
            sh.offsets.clear(); sh.lengths.clear(); sh.globalId.clear();
            if (N == 1) {                         // the whole database: no host copy, the caller's buffer goes up as it is
                sh.offsets = db->offsets;
                sh.lengths.assign(db->lengths, db->lengths + db->n);
                sh.globalId.resize(db->n);
                for (size_t i = 0; i < db->n; i++) sh.globalId[i] = (uint32_t) i;
                if (db->n && fsgpu_db_load(sh.ctx, db->data, nullptr, sh.offsets.data(), sh.lengths.data(), db->n, db->bytes) != FSGPU_OK) die(fsgpu_last_error(sh.ctx));
                continue;
            }
            size_t bytes = 0;
            auto extent = [&](size_t i) { return ((size_t) db->lengths[i] + 3) & ~(size_t) 3; };   // residues + padding to 4 (makepaddedseqdb.cpp:88-89)
            for (size_t i = k; i < db->n; i += N) bytes += extent(i);
            packed.resize(bytes);
            size_t at = 0;
            for (size_t i = k; i < db->n; i += N) {
                const size_t len = extent(i);
                // a database not written by makepaddedseqdb may end an entry without its padding: never read past the caller's buffer,
                // the missing bytes are X like the padding would be
                if (db->offsets[i] > db->bytes || (size_t) db->lengths[i] > db->bytes - db->offsets[i]) die("target database: an entry lies outside the data buffer");
                const size_t have = std::min<size_t>(len, db->bytes - db->offsets[i]);
                memcpy(packed.data() + at, db->data + db->offsets[i], have);
                if (have < len) memset(packed.data() + at + have, 20, len - have);
                sh.offsets.push_back(at); sh.lengths.push_back(db->lengths[i]); sh.globalId.push_back((uint32_t) i);
                at += len;
            }
            sh.offsets.push_back(at);
            if (sh.lengths.empty()) continue;
            if (fsgpu_db_load(sh.ctx, packed.data(), nullptr, sh.offsets.data(), sh.lengths.data(), sh.lengths.size(), bytes) != FSGPU_OK) die(fsgpu_last_error(sh.ctx));
        }
        s->resident = db;
    }
    dbmanager = dbhandle;
}

// libmarv shares device allocations between processes through CUDA IPC handles here; the shim's server keeps the DB in its
// own context and serves scans over the shared-memory protocol (gpuserver.cpp), so there is nothing to import
void Marv::setDbWithAllocation(void *dbhandle, const std::string &) { setDb(dbhandle); }
std::string Marv::getDbMemoryHandle() { return std::string(); }

void Marv::printInfo() {
    ShimState *s = static_cast<ShimState *>(cudasw);
    for (const Shard &sh : s->shards)
        fprintf(stderr, "Marv (fsgpu): device %d, %llu targets, %llu residues resident\n", fsgpu_device(sh.ctx),
                (unsigned long long) fsgpu_db_size(sh.ctx), (unsigned long long) fsgpu_db_residues(sh.ctx));
}
void Marv::prefetch() {}                  // setDb already placed the database in HBM
void Marv::startTimer() { static_cast<ShimState *>(cudasw)->timer = std::chrono::steady_clock::now(); }
void Marv::stopTimer() {}

Marv::Stats Marv::scan(const char *sequence, size_t sequenceLength, int8_t *pssm, Result *results) {
    ShimState *s = static_cast<ShimState *>(cudasw);
    Stats st;
    st.results = 0; st.numOverflows = 0; st.seconds = 0; st.gcups = 0;
    const size_t L = sequenceLength;
    if (L == 0) return st;
    if (!s->resident) die("scan before setDb");
    // The CPU kernel saturates at 255 - bias with bias = |min(matrix)| + |min(rounded composition bias)|
    // (StripedSmithWaterman.cpp:1375-1406); the caller passes pssm[a][i] = matrix[a][q_i] + round(bias_i) only.
    // (1) A profile that decomposes exactly over a matrix this library carries (3di.out, whose X row is zero, and blosum62.out,
    //     whose X row is -1; both at the prefilter's 2.0 bits) gets that matrix's true minimum and the exact per-position bias
    //     pssm[X][i] - matrix[X][q_i], even if the query lacks the residues of a minimal pair.
    // (2) Any other profile is read under the assumption matrix[X][*] == 0 (then pssm[X][i] IS the rounded bias and
    //     pssm[a][i] - pssm[X][i] the matrix entry).  A non-zero X row would shift both terms silently, so the assumption is
    //     tested where the profile allows it (query residue X: the whole column must equal the bias; and a rounded composition
    //     bias is small) and the scan is refused when it fails.
    const int A = alphabetSize, X = A - 1;
    int cbMin = 0, matMin = 0;
    for (size_t i = 0; i < L; i++) if ((unsigned char) sequence[i] >= A) die("query residue code out of range");
    bool known = false;
    for (int which : {FSHOST_MAT_3DI, FSHOST_MAT_BLOSUM62}) {
        fshost_matrix *m = fshost_matrix_create(which, 2.0f, 0.0f);
        if (!m) continue;
        const int16_t *sub = fshost_matrix_scores(m);
        bool same = fshost_matrix_size(m) == A;
        int cbm = 0;
        for (size_t i = 0; i < L && same; i++) {
            const int q = (unsigned char) sequence[i], cb = (int) pssm[(size_t) X * L + i] - (int) sub[X * A + q];
            cbm = std::min(cbm, cb);
            for (int a = 0; a < A && same; a++) same = ((int) pssm[(size_t) a * L + i] - cb) == (int) sub[a * A + q];
        }
        if (same) {
            for (int k = 0; k < A * A; k++) matMin = std::min(matMin, (int) sub[k]);
            cbMin = cbm; known = true;
        }
        fshost_matrix_free(m);
        if (same) break;
    }
    if (!known) {
        for (size_t i = 0; i < L; i++) {
            const int q = (unsigned char) sequence[i];
            const int cb = pssm[(size_t) X * L + i];
            if (abs(cb) > 8) die("profile of an unknown substitution matrix whose X row is not a plain composition bias: the CPU path's saturation cap cannot be derived from it");
            cbMin = std::min(cbMin, cb);
            for (int a = 0; a < A; a++) {
                matMin = std::min(matMin, (int) pssm[(size_t) a * L + i] - cb);
                if (q == X && (int) pssm[(size_t) a * L + i] != cb) die("profile does not come from a substitution matrix with a zero X row (profile queries are not supported)");
            }
        }
    }
    const int cap = 255 - (abs(matMin) + abs(cbMin));
    const auto t0 = std::chrono::steady_clock::now();
    // all devices scan their shard of the targets concurrently; every shard returns ITS top maxSeqs, the global top maxSeqs
    // is among them
    for (Shard &sh : s->shards)
        if (!sh.lengths.empty() && fsgpu_gapless_launch(sh.ctx, pssm, (int) L, std::max(cap, 0), -1, -1, (int) s->maxSeqs) != FSGPU_OK) die(fsgpu_last_error(sh.ctx));
    s->merged.clear();
    uint64_t residues = 0;
    for (Shard &sh : s->shards) {
        if (sh.lengths.empty()) continue;
        int n = 0;
        if (fsgpu_gapless_finish(sh.ctx, sh.hits.data(), &n) != FSGPU_OK) die(fsgpu_last_error(sh.ctx));
        for (int k = 0; k < n; k++) s->merged.push_back({sh.globalId[sh.hits[k].id], sh.hits[k].score});
        residues += fsgpu_db_residues(sh.ctx);
    }
    if (s->shards.size() > 1)
        std::sort(s->merged.begin(), s->merged.end(), [](const fsgpu_hit &a, const fsgpu_hit &b) { return a.score != b.score ? a.score > b.score : a.id < b.id; });
    const int nout = (int) std::min(s->merged.size(), s->maxSeqs);
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int k = 0; k < nout; k++) results[k] = Result(s->merged[k].id, s->merged[k].score, 0, 0);
    st.results = (size_t) nout;
    st.gcups = st.seconds > 0 ? (double) residues * (double) L / st.seconds * 1e-9 : 0;
    return st;
}
