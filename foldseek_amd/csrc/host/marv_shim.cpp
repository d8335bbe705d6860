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

struct ShimState {
    fsgpu_ctx *ctx = nullptr;
    const ShimDb *resident = nullptr;
    size_t maxSeqs = 0;
    int maxSeqLength = 0;
    std::vector<fsgpu_hit> hits;
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
    if (fsgpu_create(0, &s->ctx) != FSGPU_OK) die(fsgpu_last_error(nullptr));
    s->hits.resize(std::max<size_t>(maxSeqs, 1));
    cudasw = s;
}

Marv::~Marv() {
    ShimState *s = static_cast<ShimState *>(cudasw);
    if (s) { fsgpu_destroy(s->ctx); delete s; }
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
        if (fsgpu_db_load(s->ctx, db->data, nullptr, db->offsets.data(), db->lengths, db->n, db->bytes) != FSGPU_OK) die(fsgpu_last_error(s->ctx));
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
    fprintf(stderr, "Marv (fsgpu): device %d, %llu targets, %llu residues resident\n", fsgpu_device(s->ctx),
            (unsigned long long) fsgpu_db_size(s->ctx), (unsigned long long) fsgpu_db_residues(s->ctx));
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
    // (StripedSmithWaterman.cpp:1375-1406).  Both come out of what the caller passes: the X row of every MMseqs matrix is
    // zero, so pssm[X][i] is position i's rounded composition bias, and pssm[a][i] - pssm[X][i] is matrix[a][q_i]; the
    // matrix is symmetric, so the minimum over all entries with one index in the query is the matrix minimum as soon
    // as a residue of a minimal pair occurs in the query (see below for the matrices known to this library); a query that
    // fails the consistency test (profile queries: no zero X row semantics) is refused.
    const int A = alphabetSize, X = A - 1;
    int cbMin = 0, matMin = 0;
    for (size_t i = 0; i < L; i++) {
        const int q = (unsigned char) sequence[i];
        if (q >= A) die("query residue code out of range");
        const int cb = pssm[(size_t) X * L + i];
        cbMin = std::min(cbMin, cb);
        for (int a = 0; a < A; a++) matMin = std::min(matMin, (int) pssm[(size_t) a * L + i] - cb);
        if (q == X && (pssm[(size_t) 0 * L + i] - cb) != 0) die("profile does not come from a substitution matrix with a zero X row (profile queries are not supported)");
    }
    // Exact for the matrices this library carries (3di.out / blosum62.out at the prefilter's 2.0 bits): when every derived
    // entry agrees with one of them, its true minimum is used even if the query lacks the residues that reach it.
    for (int which : {FSHOST_MAT_3DI, FSHOST_MAT_BLOSUM62}) {
        fshost_matrix *m = fshost_matrix_create(which, 2.0f, 0.0f);
        if (!m) continue;
        const int16_t *sub = fshost_matrix_scores(m);
        bool same = fshost_matrix_size(m) == A;
        for (size_t i = 0; i < L && same; i++) {
            const int q = (unsigned char) sequence[i], cb = pssm[(size_t) X * L + i];
            for (int a = 0; a < A && same; a++) same = ((int) pssm[(size_t) a * L + i] - cb) == (int) sub[a * A + q];
        }
        if (same) for (int k = 0; k < A * A; k++) matMin = std::min(matMin, (int) sub[k]);
        fshost_matrix_free(m);
        if (same) break;
    }
    const int cap = 255 - (abs(matMin) + abs(cbMin));
    const auto t0 = std::chrono::steady_clock::now();
    int nout = 0;
    if (fsgpu_gapless_scan(s->ctx, pssm, (int) L, std::max(cap, 0), -1, -1, (int) s->maxSeqs, s->hits.data(), &nout) != FSGPU_OK) die(fsgpu_last_error(s->ctx));
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int k = 0; k < nout; k++) results[k] = Result(s->hits[k].id, s->hits[k].score, 0, 0);
    st.results = (size_t) nout;
    st.gcups = st.seconds > 0 ? (double) fsgpu_db_residues(s->ctx) * (double) L / st.seconds * 1e-9 : 0;
    return st;
}
