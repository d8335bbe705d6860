// ref_kmer_harness.cpp -- TEST INFRASTRUCTURE ONLY (part of oracle/_ref).
//
// extern "C" driver around the REFERENCE's own k-mer prefilter classes, compiled from the sources where they
// lie under /root/reference by oracle/Makefile (nothing is copied into this repository):
//   IndexTable (M/src/prefiltering/IndexTable.h:67-493), SequenceLookup, Masker (M/src/commons/Masker.cpp)
//   ExtendedSubstitutionMatrix::calcScoreMatrix (ExtendedSubstitutionMatrix.cpp:20-69)
//   KmerGenerator::generateKmerList (KmerGenerator.cpp:108-184)
//   QueryMatcher::matchQuery / match (QueryMatcher.cpp:103-376), CacheFriendlyOperations, UngappedAlignment
//
// IndexBuilder::fillDatabase (IndexBuilder.cpp:56-271) and Prefiltering::runSplit (Prefiltering.cpp:755-982)
// read their sequences through DBReader, which drags in the whole DB/CLI layer; the two loops below call the
// same IndexTable / Masker / QueryMatcher methods in the same order on in-memory ASCII sequences instead.
#include "QueryMatcher.h"
#include "IndexTable.h"
#include "SequenceLookup.h"
#include "ExtendedSubstitutionMatrix.h"
#include "SubstitutionMatrix.h"
#include "Masker.h"
#include "Sequence.h"
#include "Parameters.h"
#include "Util.h"
#include "ref_resources.h"

#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

std::string mat3di() {
    return std::string("mat3di.out:") + std::string((const char *) ref_mat3di_out, ref_mat3di_out_len);
}

// getScoreLookup (IndexBuilder.cpp:11-22)
char *idScores(BaseMatrix &m) {
    char *l = new char[m.alphabetSize];
    for (int a = 0; a < m.alphabetSize; a++) l[a] = (char) m.subMatrix[a][a];
    return l;
}

} // namespace

extern "C" {

struct RefKpfParams {
    int32_t kmerSize, spaced, kmerThr, maxResListLen, compBias, minDiagScoreThr, maskLowerCase, maskNrepeats;
    float compBiasScale;
    int32_t bins;                 // 0: as the reference picks from Util::getL2CacheSize(); else force this BINSIZE
    int64_t maxDbMatches;         // 0: reference value 2*max(1e6,N); else override (to exercise the overflow path at test sizes)
    int64_t foundDiagonalsSize;   // 0: reference value max(1e6,N)
    int32_t noDiagScore;          // 1: diagonalScoring = false (--diag-score 0): the per-target k-mer match count is the score
    int32_t pad;
};
struct RefKpfHit { uint32_t id; int32_t score; uint16_t diag; uint16_t pad; };

} // extern "C"

namespace {

// QueryMatcher with the two host-dependent sizes exposed (protected members of the reference class)
struct KpfMatcher : QueryMatcher {
    KpfMatcher(IndexTable *it, SequenceLookup *sl, BaseMatrix *k, BaseMatrix *u, short thr, int ks, size_t dbSize, unsigned maxLen,
            size_t maxHits, bool cb, float cbs, unsigned minDiag, const RefKpfParams &p)
        : QueryMatcher(it, sl, k, u, thr, ks, dbSize, maxLen, maxHits, cb, cbs, p.noDiagScore == 0, minDiag, false, false, NULL, Parameters::DBTYPE_AMINO_ACIDS) {
        if (p.bins != 0 && (unsigned) p.bins != activeCounter) {
            deleteDiagonalMatcher(activeCounter);
            // initDiagonalMatcher picks BINSIZE x as the first x with dbsize/x < L2 (QueryMatcher.cpp:460-488)
            const uint64_t l2 = Util::getL2CacheSize();
            size_t fake = (p.bins == 2) ? 2 : (size_t) l2 * (p.bins / 2);
            initDiagonalMatcher(fake, maxDbMatches);
            if ((int) activeCounter != p.bins) abort();
        }
        if (p.maxDbMatches > 0 && (size_t) p.maxDbMatches <= maxDbMatches) {
            maxDbMatches = p.maxDbMatches;
            lastSequenceHit = databaseHits + maxDbMatches;
        }
        if (p.foundDiagonalsSize > 0 && (size_t) p.foundDiagonalsSize <= foundDiagonalsSize) foundDiagonalsSize = p.foundDiagonalsSize;
    }
    unsigned bins() const { return activeCounter; }
};

struct Kpf {
    RefKpfParams p;
    SubstitutionMatrix *kmerSubMat, *ungappedSubMat;
    ScoreMatrix three, two;
    IndexTable *indexTable;
    SequenceLookup *lookup;
    size_t n;
    unsigned maxLen;
    int alphabetSize;
};

} // namespace

extern "C" {

uint64_t ref_l2_cache_size() { return Util::getL2CacheSize(); }

// Prefiltering::Prefiltering (matrices, :62-69,220-225) + getIndexTable (:544-583) + IndexBuilder::fillDatabase.
void *ref_kpf_create(const RefKpfParams *pp, const char *tcat, const int64_t *toff, const int32_t *tlen, int64_t n, int threads) {
    Kpf *h = new Kpf();
    h->p = *pp;
    std::string m = mat3di();
    h->kmerSubMat = new SubstitutionMatrix(m.c_str(), 8.0, -0.2f);
    h->ungappedSubMat = new SubstitutionMatrix(m.c_str(), 2.0, -0.2f);
    h->alphabetSize = h->kmerSubMat->alphabetSize;
    h->kmerSubMat->alphabetSize = h->alphabetSize - 1;
    h->two = ExtendedSubstitutionMatrix::calcScoreMatrix(*h->kmerSubMat, 2);
    h->three = ExtendedSubstitutionMatrix::calcScoreMatrix(*h->kmerSubMat, 3);
    h->kmerSubMat->alphabetSize = h->alphabetSize;
    h->n = (size_t) n;
    unsigned maxLen = 1;
    size_t aaSize = 0;
    for (int64_t i = 0; i < n; i++) { maxLen = std::max(maxLen, (unsigned) tlen[i]); aaSize += tlen[i]; }
    h->maxLen = maxLen + 2;

    const int seqType = Parameters::DBTYPE_AMINO_ACIDS;
    Sequence tseq(h->maxLen, seqType, h->kmerSubMat, pp->kmerSize, pp->spaced != 0, pp->compBias != 0, true, "");
    h->indexTable = new IndexTable(h->alphabetSize - 1, pp->kmerSize, false);
    h->lookup = new SequenceLookup((size_t) n, aaSize);
    std::vector<size_t> seqOff((size_t) n + 1, 0);
    size_t tableSize = 0;
    for (int64_t i = 0; i < n; i++) {
        seqOff[i + 1] = seqOff[i] + tlen[i];
        if (Util::overlappingKmers(tlen[i], tseq.getEffectiveKmerSize() > 0)) tableSize++;
    }
    char *idScoreLookup = idScores(*h->kmerSubMat);
    const int kmerThr = pp->kmerThr;   // localKmerThr == kmerThr for sequence-sequence searches (Prefiltering.cpp:555-557)
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        Masker masker(*h->kmerSubMat);
        Indexer idxer((unsigned) h->indexTable->getAlphabetSize(), pp->kmerSize);
        Sequence s(h->maxLen, seqType, h->kmerSubMat, pp->kmerSize, pp->spaced != 0, false, true, "");
        unsigned int *buffer = (unsigned int *) malloc(h->maxLen * sizeof(unsigned int));
#pragma omp for schedule(dynamic, 100)
        for (int64_t id = 0; id < n; id++) {
            s.resetCurrPos();
            s.mapSequence((size_t) id, (DBKeyType) id, tcat + toff[id], (unsigned) tlen[id]);
            masker.maskSequence(s, false, 0.999995, pp->maskLowerCase != 0, pp->maskNrepeats);
            h->indexTable->addKmerCount(&s, &idxer, buffer, kmerThr, idScoreLookup);
            h->lookup->addSequence(s.numSequence, s.L, (size_t) id, seqOff[id]);
        }
        free(buffer);
    }
    h->indexTable->initMemory(tableSize);
    h->indexTable->init();
#pragma omp parallel
    {
        Sequence s(h->maxLen, seqType, h->kmerSubMat, pp->kmerSize, pp->spaced != 0, false, true, "");
        Indexer idxer((unsigned) h->indexTable->getAlphabetSize(), pp->kmerSize);
        size_t bufferSize = h->maxLen;
        IndexEntryLocalTmp *buffer = (IndexEntryLocalTmp *) malloc(bufferSize * sizeof(IndexEntryLocalTmp));
#pragma omp for schedule(dynamic, 100)
        for (int64_t id = 0; id < n; id++) {
            s.resetCurrPos();
            s.mapSequence((size_t) id, (DBKeyType) id, h->lookup->getSequence((size_t) id));
            h->indexTable->addSequence(&s, &idxer, &buffer, bufferSize, kmerThr, idScoreLookup);
        }
        free(buffer);
    }
    delete[] idScoreLookup;
    h->indexTable->revertPointer();
    h->indexTable->sortDBSeqLists();
    return h;
}

// Test knob: adds delta to every entry of the UNGAPPED matrix (UngappedAlignment::createProfile reads it per query), so that diagonals
// score 0 and the elements keepMaxScoreElementOnly hands on with score 0 appear (--min-ungapped-score 0 with diagonal scores); the index and
// the similar-k-mer lists keep the k-mer matrix.
void ref_kpf_shift_ungapped(void *hv, int delta) {
    Kpf *h = (Kpf *) hv;
    for (int a = 0; a < h->ungappedSubMat->alphabetSize; a++)
        for (int b = 0; b < h->ungappedSubMat->alphabetSize; b++) h->ungappedSubMat->subMatrix[a][b] = (short) (h->ungappedSubMat->subMatrix[a][b] + delta);
}

// query-time parameters may change between runs (everything except kmerSize / spaced / kmerThr / masking)
void ref_kpf_set_params(void *hv, const RefKpfParams *pp) { ((Kpf *) hv)->p = *pp; }

void ref_kpf_free(void *hv) {
    Kpf *h = (Kpf *) hv;
    delete h->indexTable;
    delete h->lookup;
    ExtendedSubstitutionMatrix::freeScoreMatrix(h->three);
    ExtendedSubstitutionMatrix::freeScoreMatrix(h->two);
    delete h->kmerSubMat;
    delete h->ungappedSubMat;
    delete h;
}

// matrices as the prefilter builds them: which 0 = k-mer matrix (8 bits, -0.2), 1 = ungapped matrix (2 bits, -0.2)
int ref_kpf_submat(void *hv, int which, int16_t *sub) {
    Kpf *h = (Kpf *) hv;
    SubstitutionMatrix *m = which == 0 ? h->kmerSubMat : h->ungappedSubMat;
    for (int i = 0; i < m->alphabetSize; i++) for (int j = 0; j < m->alphabetSize; j++) sub[i * m->alphabetSize + j] = m->subMatrix[i][j];
    return m->alphabetSize;
}

// row `idx` of the 3-mer (which=3) or 2-mer (which=2) extended matrix: scores and indices, `size` elements
int64_t ref_kpf_scorematrix_row(void *hv, int which, int64_t idx, int16_t *score, uint32_t *index) {
    Kpf *h = (Kpf *) hv;
    ScoreMatrix &m = which == 3 ? h->three : h->two;
    for (size_t z = 0; z < m.elementSize; z++) { score[z] = m.score[idx * m.rowSize + z]; index[z] = m.index[idx * m.rowSize + z]; }
    return (int64_t) m.elementSize;
}

int64_t ref_kpf_index_entries(void *hv) { return (int64_t) ((Kpf *) hv)->indexTable->getTableEntriesNum(); }

// index list of one k-mer
int64_t ref_kpf_index_list(void *hv, int64_t kmer, uint32_t *seqId, uint16_t *pos, int64_t cap) {
    Kpf *h = (Kpf *) hv;
    size_t sz;
    IndexEntryLocal *e = h->indexTable->getDBSeqList((size_t) kmer, &sz);
    for (size_t i = 0; i < sz && (int64_t) i < cap; i++) { seqId[i] = e[i].seqId; pos[i] = e[i].position_j; }
    return (int64_t) sz;
}

// whole offsets table (tableSize + 1 entries) -> caller buffer
int64_t ref_kpf_index_offsets(void *hv, uint64_t *out, int64_t cap) {
    Kpf *h = (Kpf *) hv;
    const size_t ts = h->indexTable->getTableSize();
    if (out != NULL) for (size_t i = 0; i <= ts && (int64_t) i < cap; i++) out[i] = h->indexTable->getOffsets()[i];
    return (int64_t) ts;
}

// masked numeric target as stored in the SequenceLookup
int ref_kpf_masked(void *hv, int64_t id, uint8_t *out) {
    Kpf *h = (Kpf *) hv;
    std::pair<const unsigned char *, const unsigned int> s = h->lookup->getSequence((size_t) id);
    memcpy(out, s.first, s.second);
    return (int) s.second;
}

// similar k-mer list of one numeric k-mer at threshold thr (KmerGenerator::generateKmerList)
int64_t ref_kpf_kmer_list(void *hv, const uint8_t *kmer, int thr, uint64_t *out, int64_t cap) {
    Kpf *h = (Kpf *) hv;
    KmerGenerator g(h->p.kmerSize, h->indexTable->getAlphabetSize(), (short) thr);
    g.setDivideStrategy(&h->three, &h->two);
    std::pair<size_t *, size_t> l = g.generateKmerList(kmer);
    for (size_t i = 0; i < l.second && (int64_t) i < cap; i++) out[i] = l.first[i];
    return (int64_t) l.second;
}

// Prefiltering::runSplit per-query body (:847-917) for nq ASCII queries; identity[q] = target id treated as the
// query itself (sameQTDB / includeIdentical) or -1.  out: nq x maxResListLen hits, cnt[q]; stats[q*4..] =
// kmersPerPos, dbMatches, diagonalOverflow, bins.  Returns the wall-clock seconds of the OpenMP region.
double ref_kpf_run(void *hv, const char *qcat, const int64_t *qoff, const int32_t *qlen, int64_t nq, const int64_t *identity,
                   int threads, RefKpfHit *out, int32_t *cnt, double *stats) {
    Kpf *h = (Kpf *) hv;
    const RefKpfParams &p = h->p;
    unsigned qMax = 1;
    for (int64_t i = 0; i < nq; i++) qMax = std::max(qMax, (unsigned) qlen[i]);
    qMax += 2;
    const size_t maxRes = std::min((size_t) p.maxResListLen, h->n);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel
    {
        Sequence seq(qMax, Parameters::DBTYPE_AMINO_ACIDS, h->kmerSubMat, p.kmerSize, p.spaced != 0, p.compBias != 0, true, "");
        KpfMatcher matcher(h->indexTable, h->lookup, h->kmerSubMat, h->ungappedSubMat, (short) p.kmerThr, p.kmerSize, h->n,
                        std::max(h->maxLen, qMax), maxRes, p.compBias != 0, p.compBiasScale, (unsigned) p.minDiagScoreThr, p);
        matcher.setSubstitutionMatrix(&h->three, &h->two);
#pragma omp for schedule(dynamic, 1)
        for (int64_t q = 0; q < nq; q++) {
            seq.mapSequence((size_t) q, (DBKeyType) q, qcat + qoff[q], (unsigned) qlen[q]);
            const DBLocalId tid = identity != NULL && identity[q] >= 0 ? (DBLocalId) identity[q] : DB_LOCAL_ID_INVALID;
            std::pair<hit_t *, size_t> r = matcher.matchQuery(&seq, tid, false);
            cnt[q] = (int32_t) r.second;
            for (size_t i = 0; i < r.second; i++) {
                RefKpfHit &o = out[q * (int64_t) p.maxResListLen + i];
                o.id = (uint32_t) r.first[i].seqId; o.score = r.first[i].prefScore; o.diag = r.first[i].diagonal; o.pad = 0;
            }
            if (stats != NULL) {
                stats[q * 4 + 0] = matcher.getStatistics()->kmersPerPos;
                stats[q * 4 + 1] = (double) matcher.getStatistics()->dbMatches;
                stats[q * 4 + 2] = (double) matcher.getStatistics()->diagonalOverflow;
                stats[q * 4 + 3] = (double) matcher.bins();
            }
        }
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

} // extern "C"
