/* fs_kmer_oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported or called by the product path.
 *
 * Plain-C, scalar, sequential restatement of the reference's k-mer prefilter,
 * literal enough to serve as the specification of every ordering rule the GPU path has to reproduce:
 *   Masker::maskSequence / maskRepeats            M/src/commons/Masker.cpp:15-57,84-120
 *   IndexTable::addKmerCount / addSequence        M/src/prefiltering/IndexTable.h:129-171,364-415
 *   ExtendedSubstitutionMatrix::calcScoreMatrix   M/src/prefiltering/ExtendedSubstitutionMatrix.cpp:20-69
 *   KmerGenerator::generateKmerList               M/src/prefiltering/KmerGenerator.cpp:108-217
 *   QueryMatcher::match / matchQuery / getResult  M/src/prefiltering/QueryMatcher.cpp:243-376,103-240,401-457
 *   CacheFriendlyOperations<B>::*                 M/src/prefiltering/CacheFriendlyOperations.cpp:38-384
 *   UngappedAlignment::createProfile / scoring    M/src/prefiltering/UngappedAlignment.cpp:388-428,45-57,430-443
 * Pinned to the compiled reference (oracle/_ref, ref_kmer_harness.cpp) by tests/test_kmer_oracle_vs_ref.py
 * and to the committed fixtures under tests/golden/.
 *
 * Sequences are numeric codes 0..20 (20 = X); +32 marks a soft-masked (lower-case) residue.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define FKO_X 20
#define FKO_MAX_KMER_RESULT_SIZE (262144 * 32)     /* KmerGenerator.h:45 */

typedef struct {
    int32_t kmerSize, spaced, kmerThr, maxResListLen, compBias, minDiagScoreThr, maskLowerCase, maskNrepeats;
    float compBiasScale;
    int32_t bins;                 /* BINSIZE of CacheFriendlyOperations (QueryMatcher::initDiagonalMatcher); 0 = derive from l2CacheSize */
    int64_t maxDbMatches;         /* 0 = 2*max(1e6,N)  (QueryMatcher.cpp:45) */
    int64_t foundDiagonalsSize;   /* 0 = max(1e6,N)    (QueryMatcher.cpp:44) */
    uint64_t l2CacheSize;         /* Util::getL2CacheSize() of the host being emulated */
    int32_t noDiagScore;          /* 1 = diagonalScoring == false (--diag-score 0): k-mer match counts as scores (QueryMatcher.cpp:215-232) */
    int32_t pad;
} fko_params;

typedef struct { uint32_t id; int32_t score; uint16_t diag; uint16_t pad; } fko_hit;

typedef struct { uint32_t seqId; uint16_t pos; } Entry;                    /* IndexEntryLocal */
typedef struct { uint32_t id; uint16_t diagonal; uint8_t count; } Counter; /* CounterResult */

typedef struct {
    int size, rowSize;         /* alphabet^k */
    int16_t *score;            /* [size][size] sorted descending, stable in permutation order */
    uint32_t *index;
} ScoreMat;

typedef struct {
    fko_params p;
    int alphabet;              /* 21 */
    int16_t kmerSub[21 * 21], ungSub[21 * 21];
    double pback[21];
    ScoreMat three, two;
    /* index */
    int64_t n;
    uint64_t tableSize;
    uint64_t *offsets;         /* tableSize + 1 */
    Entry *entries;
    uint64_t nEntries;
    /* SequenceLookup: masked numeric targets */
    uint8_t *lookup;
    int64_t *lookOff;          /* n + 1 */
    int pattern[32], patternSize, kpos[16];
} fko;

/* ---- spaced patterns (M/src/commons/Sequence.h:18-60); only the sizes the structure prefilter can pick ---- */
static void set_pattern(fko *h) {
    static const int s6[] = {1, 1, 0, 1, 0, 1, 0, 0, 1, 1};
    static const int s7[] = {1, 1, 0, 1, 0, 1, 0, 0, 1, 0, 1, 1};
    int k = h->p.kmerSize, n = 0;
    if (!h->p.spaced) { h->patternSize = k; for (int i = 0; i < k; i++) h->pattern[i] = 1; }
    else if (k == 6) { h->patternSize = 10; for (int i = 0; i < 10; i++) h->pattern[i] = s6[i]; }
    else { h->patternSize = 12; for (int i = 0; i < 12; i++) h->pattern[i] = s7[i]; }
    for (int i = 0; i < h->patternSize; i++) if (h->pattern[i]) h->kpos[n++] = i;
}

/* ---- Masker::maskRepeats + lower-case masking + finalizeMasking ------------------------------------------ */
void fko_mask_target(const uint8_t *codes, int L, int maskLower, int maskNrepeats, uint8_t *out) {
    for (int i = 0; i < L; i++) out[i] = codes[i] >= 32 ? (uint8_t) (codes[i] - 32) : codes[i];
    if (maskNrepeats > 0) {
        unsigned repeatCount = 0;
        int startOfRepeat = -1;
        char previousChar = '\0';
        for (unsigned pos = 0; pos < (unsigned) L; ++pos) {
            char c = (char) out[pos];
            if (c == previousChar) {
                repeatCount++;
            } else {
                if (repeatCount > (unsigned) maskNrepeats)
                    for (unsigned i = (unsigned) startOfRepeat; i < pos; ++i) out[i] = FKO_X;   /* startOfRepeat == -1 wraps: no-op */
                repeatCount = 1; startOfRepeat = (int) pos; previousChar = c;
            }
        }
        if (repeatCount > (unsigned) maskNrepeats)
            for (unsigned i = (unsigned) startOfRepeat; i < (unsigned) L; ++i) out[i] = FKO_X;
    }
    if (maskLower) for (int i = 0; i < L; i++) if (codes[i] >= 32) out[i] = FKO_X;
}

/* ---- ExtendedSubstitutionMatrix::calcScoreMatrix --------------------------------------------------------- */
static void build_scoremat(ScoreMat *m, const int16_t *sub, int alpha, int k) {
    int size = 1;
    for (int i = 0; i < k; i++) size *= alpha;
    m->size = size; m->rowSize = size;
    m->score = (int16_t *) malloc((size_t) size * size * sizeof(int16_t));
    m->index = (uint32_t *) malloc((size_t) size * size * sizeof(uint32_t));
    /* permutation order: first position outermost (createCartesianProduct); index = sum a_i * alpha^i */
    int *permIdx = (int *) malloc(size * sizeof(int));
    uint8_t *perm = (uint8_t *) malloc((size_t) size * k);
    for (int r = 0; r < size; r++) {
        int rem = r, idx = 0, pw = 1;
        for (int i = k - 1; i >= 0; i--) { perm[r * k + i] = (uint8_t) (rem % alpha); rem /= alpha; }
        for (int i = 0; i < k; i++) { idx += perm[r * k + i] * pw; pw *= alpha; }
        permIdx[r] = idx;
    }
    int16_t *tmp = (int16_t *) malloc(size * sizeof(int16_t));
    int *cnt = (int *) malloc(65536 * sizeof(int));
    for (int i = 0; i < size; i++) {
        int lo = INT_MAX, hi = INT_MIN;
        for (int j = 0; j < size; j++) {
            int s = 0;
            for (int z = 0; z < k; z++) s += sub[perm[i * k + z] * 21 + perm[j * k + z]];
            tmp[j] = (int16_t) s;
            if (s < lo) lo = s;
            if (s > hi) hi = s;
        }
        /* stable descending sort == counting sort over the score range, scanning j in permutation order */
        int range = hi - lo + 1;
        memset(cnt, 0, range * sizeof(int));
        for (int j = 0; j < size; j++) cnt[hi - tmp[j]]++;
        int acc = 0;
        for (int r = 0; r < range; r++) { int c = cnt[r]; cnt[r] = acc; acc += c; }
        int16_t *rs = m->score + (size_t) permIdx[i] * size;
        uint32_t *ri = m->index + (size_t) permIdx[i] * size;
        for (int j = 0; j < size; j++) { int d = cnt[hi - tmp[j]]++; rs[d] = tmp[j]; ri[d] = (uint32_t) permIdx[j]; }
    }
    free(cnt); free(tmp); free(perm); free(permIdx);
}

/* ---- KmerGenerator ------------------------------------------------------------------------------------- */
typedef struct {
    int steps, divide[8];
    const ScoreMat *mat[8];
    int16_t *outScore[2];
    uint64_t *outIndex[2];
} KmerGen;

static uint64_t ipow(uint64_t b, int e) { uint64_t r = 1; while (e-- > 0) r *= b; return r; }

static void kmergen_init(KmerGen *g, const fko *h) {
    /* setDivideStrategy(three, two) incl. the final std::reverse (KmerGenerator.cpp:44-87) */
    int k = h->p.kmerSize, t = k / 3, n = 0;
    int div[8]; const ScoreMat *mm[8];
    switch (k % 3) {
        case 0: for (int i = 0; i < t; i++) { div[n] = 3; mm[n++] = &h->three; } break;
        case 1: for (int i = 0; i < t - 1; i++) { div[n] = 3; mm[n++] = &h->three; }
                div[n] = 2; mm[n++] = &h->two; div[n] = 2; mm[n++] = &h->two; break;
        default: for (int i = 0; i < t; i++) { div[n] = 3; mm[n++] = &h->three; }
                div[n] = 2; mm[n++] = &h->two; break;
    }
    g->steps = n;
    for (int i = 0; i < n; i++) { g->divide[i] = div[n - 1 - i]; g->mat[i] = mm[n - 1 - i]; }
    for (int i = 0; i < 2; i++) {
        g->outScore[i] = (int16_t *) malloc((size_t) FKO_MAX_KMER_RESULT_SIZE * sizeof(int16_t));
        g->outIndex[i] = (uint64_t *) malloc((size_t) FKO_MAX_KMER_RESULT_SIZE * sizeof(uint64_t));
    }
}
static void kmergen_free(KmerGen *g) { for (int i = 0; i < 2; i++) { free(g->outScore[i]); free(g->outIndex[i]); } }

/* generateKmerList (addIdentity == false); returns the list and its length */
static const uint64_t *kmer_list(KmerGen *g, const fko *h, const uint8_t *kmer, short threshold, size_t *len) {
    const int alpha = h->alphabet - 1;
    size_t kmerIndex[8], stepMul[8];
    short highest[8], possibleRest[8];
    int before = 0;
    for (int i = 0; i < g->steps; i++) {
        size_t idx = 0, pw = 1;
        for (int z = 0; z < g->divide[i]; z++) { idx += kmer[before + z] * pw; pw *= alpha; }
        kmerIndex[i] = idx;
        stepMul[i] = (size_t) ipow(alpha, before);
        highest[i] = g->mat[i]->score[idx * g->mat[i]->rowSize];
        before += g->divide[i];
    }
    possibleRest[g->steps - 1] = 0;
    for (int i = g->steps - 1; i >= 1; i--) possibleRest[i - 1] = (short) (highest[i] + possibleRest[i]);

    short cutoff1 = (short) (threshold - possibleRest[0]);
    const ScoreMat *in = g->mat[0];
    size_t sizeIn = (size_t) in->size;
    const int16_t *inScore = &in->score[kmerIndex[0] * in->rowSize];
    for (size_t pos = 0; pos < (size_t) in->rowSize && inScore[pos] >= cutoff1; pos++)
        g->outIndex[1][pos] = in->index[kmerIndex[0] * in->rowSize + pos];
    const uint64_t *inIndex = g->outIndex[1];
    int i;
    for (i = 0; i < g->steps - 1; i++) {
        const ScoreMat *nx = g->mat[i + 1];
        const int16_t *s2 = &nx->score[kmerIndex[i + 1] * nx->rowSize];
        const uint32_t *i2 = &nx->index[kmerIndex[i + 1] * nx->rowSize];
        int16_t *os = g->outScore[i % 2];
        uint64_t *oi = g->outIndex[i % 2];
        /* calculateArrayProduct (KmerGenerator.cpp:187-217) */
        size_t counter = 0;
        for (size_t a = 0; a < sizeIn; a++) {
            const short sa = inScore[a];
            const uint64_t ka = inIndex[a];
            if (sa < cutoff1) break;
            const short cutoff2 = (short) (threshold - sa - possibleRest[i + 1]);
            for (size_t b = 0; b < (size_t) nx->size && (counter + 1 < (size_t) FKO_MAX_KMER_RESULT_SIZE) && s2[b] >= cutoff2; b++) {
                os[counter] = (int16_t) (sa + s2[b]);
                oi[counter] = ka + (uint64_t) i2[b] * stepMul[i + 1];
                counter++;
            }
            if (counter + 1 >= (size_t) FKO_MAX_KMER_RESULT_SIZE) break;
        }
        inScore = os; inIndex = oi; cutoff1 = -1000; sizeIn = counter;
    }
    *len = sizeIn;
    return g->outIndex[(i - 1) % 2];
}

/* ---- k-mer iteration (Sequence::hasNextKmer / nextKmer / kmerContainsX) --------------------------------- */
static int kmer_at(const fko *h, const uint8_t *seq, int L, int pos, uint8_t *kmer) {   /* returns 0 past the end, 1 ok, 2 contains X */
    if (pos + h->patternSize > L) return 0;
    int x = 0;
    for (int i = 0; i < h->p.kmerSize; i++) { kmer[i] = seq[pos + h->kpos[i]]; if (kmer[i] == FKO_X) x = 1; }
    return x ? 2 : 1;
}

static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *) a, y = *(const uint32_t *) b; return x < y ? -1 : x > y; }
typedef struct { uint32_t kmer; uint32_t seqId; uint16_t pos; } TmpEntry;
static int cmp_tmp(const void *a, const void *b) {   /* IndexEntryLocalTmp::comapreByIdAndPos: kmer, seqId, position */
    const TmpEntry *x = (const TmpEntry *) a, *y = (const TmpEntry *) b;
    if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
    if (x->seqId != y->seqId) return x->seqId < y->seqId ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}
static int cmp_entry(const void *a, const void *b) {
    const Entry *x = (const Entry *) a, *y = (const Entry *) b;
    if (x->seqId != y->seqId) return x->seqId < y->seqId ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}

/* k-mers of one masked target that enter the index: no X, self score >= kmerThr (IndexTable.h:139-149) */
static size_t target_kmers(const fko *h, const uint8_t *seq, int L, uint32_t seqId, TmpEntry *buf) {
    size_t n = 0;
    uint8_t kmer[16];
    const int alpha = h->alphabet - 1;
    for (int pos = 0;; pos++) {
        int r = kmer_at(h, seq, L, pos, kmer);
        if (r == 0) break;
        if (r == 2) continue;
        if (h->p.kmerThr > 0) {
            int score = 0;
            for (int z = 0; z < h->p.kmerSize; z++) score += (char) h->kmerSub[kmer[z] * 21 + kmer[z]];
            if (score < h->p.kmerThr) continue;
        }
        uint64_t idx = 0, pw = 1;
        for (int z = 0; z < h->p.kmerSize; z++) { idx += kmer[z] * pw; pw *= alpha; }
        buf[n].kmer = (uint32_t) idx; buf[n].seqId = seqId; buf[n].pos = (uint16_t) pos; n++;
    }
    return n;
}

void fko_free(void *hv) {
    fko *h = (fko *) hv;
    if (!h) return;
    free(h->three.score); free(h->three.index); free(h->two.score); free(h->two.index);
    free(h->offsets); free(h->entries); free(h->lookup); free(h->lookOff);
    free(h);
}

/* Prefiltering ctor matrices + IndexBuilder::fillDatabase */
void *fko_create(const fko_params *p, const int16_t *kmerSub21, const double *pback21, const int16_t *ungSub21,
                 const uint8_t *tcodes, const int64_t *toff, const int32_t *tlen, int64_t n) {
    fko *h = (fko *) calloc(1, sizeof(fko));
    h->p = *p;
    h->alphabet = 21;
    memcpy(h->kmerSub, kmerSub21, sizeof(h->kmerSub));
    memcpy(h->ungSub, ungSub21, sizeof(h->ungSub));
    memcpy(h->pback, pback21, sizeof(h->pback));
    set_pattern(h);
    build_scoremat(&h->three, h->kmerSub, 20, 3);
    build_scoremat(&h->two, h->kmerSub, 20, 2);
    h->n = n;
    h->tableSize = ipow(20, p->kmerSize);
    h->offsets = (uint64_t *) calloc(h->tableSize + 1, sizeof(uint64_t));
    h->lookOff = (int64_t *) malloc((n + 1) * sizeof(int64_t));
    int maxLen = 1;
    h->lookOff[0] = 0;
    for (int64_t i = 0; i < n; i++) { h->lookOff[i + 1] = h->lookOff[i] + tlen[i]; if (tlen[i] > maxLen) maxLen = tlen[i]; }
    h->lookup = (uint8_t *) malloc((size_t) h->lookOff[n] + 1);
    TmpEntry *buf = (TmpEntry *) malloc((size_t) maxLen * sizeof(TmpEntry));
    /* pass 1: mask, count distinct k-mers per sequence */
    for (int64_t id = 0; id < n; id++) {
        uint8_t *m = h->lookup + h->lookOff[id];
        fko_mask_target(tcodes + toff[id], tlen[id], p->maskLowerCase, p->maskNrepeats, m);
        size_t c = target_kmers(h, m, tlen[id], (uint32_t) id, buf);
        qsort(buf, c, sizeof(TmpEntry), cmp_tmp);
        uint32_t prev = UINT32_MAX;
        for (size_t i = 0; i < c; i++) { if (buf[i].kmer != prev) h->offsets[buf[i].kmer]++; prev = buf[i].kmer; }
    }
    /* init(): exclusive prefix sum */
    uint64_t off = 0;
    for (uint64_t k = 0; k < h->tableSize; k++) { uint64_t c = h->offsets[k]; h->offsets[k] = off; off += c; }
    h->offsets[h->tableSize] = off;
    h->nEntries = off;
    h->entries = (Entry *) malloc((off ? off : 1) * sizeof(Entry));
    uint64_t *cursor = (uint64_t *) malloc(h->tableSize * sizeof(uint64_t));
    memcpy(cursor, h->offsets, h->tableSize * sizeof(uint64_t));
    /* pass 2: fill; one entry per (k-mer, sequence) = the smallest position (sorted by kmer, seqId, pos; first wins) */
    for (int64_t id = 0; id < n; id++) {
        size_t c = target_kmers(h, h->lookup + h->lookOff[id], tlen[id], (uint32_t) id, buf);
        qsort(buf, c, sizeof(TmpEntry), cmp_tmp);
        uint32_t prev = UINT32_MAX;
        for (size_t i = 0; i < c; i++) {
            if (buf[i].kmer != prev) { Entry *e = &h->entries[cursor[buf[i].kmer]++]; e->seqId = buf[i].seqId; e->pos = buf[i].pos; }
            prev = buf[i].kmer;
        }
    }
    /* sortDBSeqLists: by (seqId, pos) -- sequential fill already is, kept for the parallel-fill semantics */
    for (uint64_t k = 0; k < h->tableSize; k++) {
        uint64_t c = h->offsets[k + 1] - h->offsets[k];
        if (c > 1) qsort(h->entries + h->offsets[k], c, sizeof(Entry), cmp_entry);
    }
    free(cursor); free(buf);
    (void) cmp_u32;
    return h;
}

/* query-time parameters may change between queries (everything except kmerSize / spaced / kmerThr / masking) */
void fko_set_params(void *hv, const fko_params *p) { ((fko *) hv)->p = *p; }

/* ---- accessors for piecewise pinning -------------------------------------------------------------------- */
int64_t fko_index_entries(void *hv) { return (int64_t) ((fko *) hv)->nEntries; }
int64_t fko_index_list(void *hv, int64_t kmer, uint32_t *seqId, uint16_t *pos, int64_t cap) {
    fko *h = (fko *) hv;
    uint64_t c = h->offsets[kmer + 1] - h->offsets[kmer];
    for (uint64_t i = 0; i < c && (int64_t) i < cap; i++) { seqId[i] = h->entries[h->offsets[kmer] + i].seqId; pos[i] = h->entries[h->offsets[kmer] + i].pos; }
    return (int64_t) c;
}
int64_t fko_index_offsets(void *hv, uint64_t *out, int64_t cap) {
    fko *h = (fko *) hv;
    if (out) for (uint64_t i = 0; i <= h->tableSize && (int64_t) i < cap; i++) out[i] = h->offsets[i];
    return (int64_t) h->tableSize;
}
/* flat copy of the index: offsets (tableSize+1), entry seqIds / positions (nEntries) */
void fko_index_copy(void *hv, uint64_t *offsets, uint32_t *seqId, uint16_t *pos) {
    fko *h = (fko *) hv;
    memcpy(offsets, h->offsets, (h->tableSize + 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < h->nEntries; i++) { seqId[i] = h->entries[i].seqId; pos[i] = h->entries[i].pos; }
}
int fko_masked(void *hv, int64_t id, uint8_t *out) {
    fko *h = (fko *) hv;
    int L = (int) (h->lookOff[id + 1] - h->lookOff[id]);
    memcpy(out, h->lookup + h->lookOff[id], L);
    return L;
}
int64_t fko_scorematrix_row(void *hv, int which, int64_t idx, int16_t *score, uint32_t *index) {
    fko *h = (fko *) hv;
    ScoreMat *m = which == 3 ? &h->three : &h->two;
    memcpy(score, m->score + idx * m->rowSize, m->size * sizeof(int16_t));
    memcpy(index, m->index + idx * m->rowSize, m->size * sizeof(uint32_t));
    return m->size;
}
int64_t fko_kmer_list(void *hv, const uint8_t *kmer, int thr, uint64_t *out, int64_t cap) {
    fko *h = (fko *) hv;
    KmerGen g;
    kmergen_init(&g, h);
    size_t len;
    const uint64_t *l = kmer_list(&g, h, kmer, (short) thr, &len);
    for (size_t i = 0; i < len && (int64_t) i < cap; i++) out[i] = l[i];
    kmergen_free(&g);
    return (int64_t) len;
}

/* ---- CacheFriendlyOperations<B> -------------------------------------------------------------------------- */
typedef struct {
    unsigned B, shift;
    uint8_t *dup;              /* duplicateBitArray, indexed by id >> shift */
    size_t dupSize;
    Counter *frame;            /* bins, laid out as a stable partition by (id & (B-1)) */
    size_t *binStart;          /* B + 1 */
    size_t cap;
    Counter *tmp;
} Cfo;

static void cfo_init(Cfo *c, unsigned B, size_t n) {
    c->B = B; c->shift = 0;
    while ((1u << c->shift) < B) c->shift++;
    c->dupSize = (n >> c->shift) + 2;
    c->dup = (uint8_t *) calloc(c->dupSize, 1);
    c->binStart = (size_t *) calloc(B + 1, sizeof(size_t));
    c->cap = 0; c->frame = NULL; c->tmp = NULL;
}
static void cfo_free(Cfo *c) { free(c->dup); free(c->binStart); free(c->frame); free(c->tmp); }
static void cfo_reserve(Cfo *c, size_t n) {
    if (n > c->cap) { c->cap = n * 2 + 16; c->frame = (Counter *) realloc(c->frame, c->cap * sizeof(Counter)); c->tmp = (Counter *) realloc(c->tmp, c->cap * sizeof(Counter)); }
}
/* hashElements: the fixed-size bins + "overflow -> double binSize and redo" of the reference are a stable partition */
static void cfo_hash(Cfo *c, const Counter *in, size_t N) {
    cfo_reserve(c, N);
    memset(c->binStart, 0, (c->B + 1) * sizeof(size_t));
    for (size_t i = 0; i < N; i++) c->binStart[(in[i].id & (c->B - 1)) + 1]++;
    for (unsigned b = 0; b < c->B; b++) c->binStart[b + 1] += c->binStart[b];
    size_t *cur = (size_t *) malloc(c->B * sizeof(size_t));
    memcpy(cur, c->binStart, c->B * sizeof(size_t));
    for (size_t i = 0; i < N; i++) c->frame[cur[in[i].id & (c->B - 1)]++] = in[i];
    free(cur);
}

/* findDuplicates(IndexEntryLocal **input ...) with computeTotalScore == false.
 * hits: the databaseHits chunk as (id, diagonal) in arrival order. */
static size_t cfo_find_duplicates(Cfo *c, const Counter *hits, size_t nHits, Counter *output, size_t outputSize) {
    cfo_hash(c, hits, nHits);
    memset(c->dup, 0, c->dupSize);
    size_t dbl = 0;
    for (unsigned bin = 0; bin < c->B; bin++) {
        const Counter *bs = c->frame + c->binStart[bin];
        const size_t sz = c->binStart[bin + 1] - c->binStart[bin];
        size_t ec = 0;
        for (size_t n = 0; n < sz; n++) {
            const size_t hb = bs[n].id >> c->shift;
            const uint8_t cur = (uint8_t) bs[n].diagonal, prev = c->dup[hb];
            c->tmp[ec].id = bs[n].id; c->tmp[ec].diagonal = bs[n].diagonal;
            ec += (cur == prev) ? 1 : 0;
            c->dup[hb] = cur;
        }
        if (dbl + ec >= outputSize) return dbl;
        for (size_t n = ec; n-- > 0;) c->dup[c->tmp[n].id >> c->shift] = (uint8_t) (((uint8_t) c->tmp[n].diagonal) + 1);
        for (size_t n = 0; n < ec; n++) {
            const size_t hb = c->tmp[n].id >> c->shift;
            output[dbl].id = c->tmp[n].id; output[dbl].count = 0; output[dbl].diagonal = c->tmp[n].diagonal;
            dbl += (c->dup[hb] != (uint8_t) c->tmp[n].diagonal) ? 1 : 0;
            c->dup[hb] = (uint8_t) c->tmp[n].diagonal;
        }
        for (size_t n = 0; n < sz; n++) c->dup[bs[n].id >> c->shift] = 0;
    }
    return dbl;
}
/* findDuplicates with computeTotalScore == true (CacheFriendlyOperations.cpp:217-241): the candidates of a bin are counted per target
 * (saturating at 255) and every target is handed on once -- at its first candidate, with that candidate's diagonal. */
static size_t cfo_find_duplicates_total(Cfo *c, const Counter *hits, size_t nHits, Counter *output, size_t outputSize) {
    cfo_hash(c, hits, nHits);
    memset(c->dup, 0, c->dupSize);
    size_t dbl = 0;
    for (unsigned bin = 0; bin < c->B; bin++) {
        const Counter *bs = c->frame + c->binStart[bin];
        const size_t sz = c->binStart[bin + 1] - c->binStart[bin];
        size_t ec = 0;
        for (size_t n = 0; n < sz; n++) {
            const size_t hb = bs[n].id >> c->shift;
            const uint8_t cur = (uint8_t) bs[n].diagonal, prev = c->dup[hb];
            c->tmp[ec].id = bs[n].id; c->tmp[ec].diagonal = bs[n].diagonal;
            ec += (cur == prev) ? 1 : 0;
            c->dup[hb] = cur;
        }
        if (dbl + ec >= outputSize) return dbl;
        for (size_t n = 0; n < ec; n++) c->dup[c->tmp[n].id >> c->shift] = 0;
        for (size_t n = 0; n < ec; n++) { uint8_t *d = &c->dup[c->tmp[n].id >> c->shift]; *d = (uint8_t) (*d + (*d < 255 ? 1 : 0)); }
        for (size_t n = 0; n < ec; n++) {
            const size_t hb = c->tmp[n].id >> c->shift;
            output[dbl].id = c->tmp[n].id; output[dbl].count = c->dup[hb]; output[dbl].diagonal = c->tmp[n].diagonal;
            dbl += (c->dup[hb] != 0) ? 1 : 0;
            c->dup[hb] = 0;
        }
        for (size_t n = 0; n < sz; n++) c->dup[bs[n].id >> c->shift] = 0;
    }
    return dbl;
}
/* mergeElementsByScore -> mergeScoreDuplicates (CacheFriendlyOperations.cpp:52-58, :150-180), the merge of the per-refill lists without
 * diagonal scoring.  Restated as the reference executes it, not as its comments describe it: per bin the counts of a target are summed
 * (saturating at 255) into its byte, every element of the bin is handed on with the byte's CURRENT value, and the byte is then left at
 * that element's diagonal byte (not zero).  So a target present in both lists comes out twice -- first with the sum, then with the low
 * byte of the first entry's diagonal as its "count" (dropped only when that byte is 0) -- and a byte left behind by one bin is the
 * starting value of the targets of later bins that share id >> shift. */
static size_t cfo_merge_score(Cfo *c, Counter *io, size_t N) {
    cfo_hash(c, io, N);
    size_t dbl = 0;
    for (unsigned bin = 0; bin < c->B; bin++) {
        const Counter *bs = c->frame + c->binStart[bin];
        const size_t sz = c->binStart[bin + 1] - c->binStart[bin];
        for (size_t n = 0; n < sz; n++) {
            uint8_t *d = &c->dup[bs[n].id >> c->shift];
            const uint8_t cur = bs[n].count, db = *d;
            *d = (cur > 0xFF - db) ? 0xFF : (uint8_t) (db + cur);
        }
        for (size_t n = 0; n < sz; n++) {
            uint8_t *d = &c->dup[bs[n].id >> c->shift];
            io[dbl].id = bs[n].id; io[dbl].count = *d; io[dbl].diagonal = bs[n].diagonal;
            dbl += (*d != 0) ? 1 : 0;
            *d = (uint8_t) bs[n].diagonal;
        }
    }
    return dbl;
}
static size_t cfo_merge_diag(Cfo *c, Counter *io, size_t N) {              /* mergeDiagonalDuplicates */
    cfo_hash(c, io, N);
    size_t dbl = 0;
    for (unsigned bin = 0; bin < c->B; bin++) {
        const Counter *bs = c->frame + c->binStart[bin];
        const size_t sz = c->binStart[bin + 1] - c->binStart[bin];
        for (size_t n = sz; n-- > 0;) c->dup[bs[n].id >> c->shift] = (uint8_t) (((uint8_t) bs[n].diagonal) + 1);
        for (size_t n = 0; n < sz; n++) {
            const size_t hb = bs[n].id >> c->shift;
            io[dbl] = bs[n];
            dbl += (c->dup[hb] != (uint8_t) bs[n].diagonal) ? 1 : 0;
            c->dup[hb] = (uint8_t) bs[n].diagonal;
        }
    }
    return dbl;
}
static size_t cfo_merge_diag_keep_scored(Cfo *c, Counter *io, size_t N) {  /* mergeDiagonalKeepScoredHitsDuplicates */
    cfo_hash(c, io, N);
    size_t dbl = 0;
    for (unsigned bin = 0; bin < c->B; bin++) {
        const Counter *bs = c->frame + c->binStart[bin];
        const size_t sz = c->binStart[bin + 1] - c->binStart[bin];
        for (size_t n = 0; n < sz; n++) c->dup[bs[n].id >> c->shift] = (uint8_t) (((uint8_t) bs[n].diagonal) + 1);
        for (size_t n = sz; n-- > 0;) {
            const size_t hb = bs[n].id >> c->shift;
            io[dbl] = bs[n];
            dbl += (io[dbl].count != 0 || c->dup[hb] != (uint8_t) bs[n].diagonal) ? 1 : 0;
            c->dup[hb] = (uint8_t) bs[n].diagonal;
        }
    }
    return dbl;
}
static size_t cfo_keep_max(Cfo *c, Counter *io, size_t N) {                /* keepMaxScoreElementOnly */
    cfo_hash(c, io, N);
    size_t dbl = 0;
    memset(c->dup, 0, c->dupSize);
    for (unsigned bin = 0; bin < c->B; bin++) {
        const Counter *bs = c->frame + c->binStart[bin];
        const size_t sz = c->binStart[bin + 1] - c->binStart[bin];
        for (size_t n = 0; n < sz; n++) {
            const size_t hb = bs[n].id >> c->shift;
            if (bs[n].count > c->dup[hb]) c->dup[hb] = bs[n].count;
        }
        for (size_t n = 0; n < sz; n++) {
            const size_t hb = bs[n].id >> c->shift;
            io[dbl] = bs[n];
            const int found = c->dup[hb] == bs[n].count;
            dbl += found;
            c->dup[hb] = (uint8_t) (c->dup[hb] * (1 - found));
        }
    }
    return dbl;
}

/* ---- UngappedAlignment --------------------------------------------------------------------------------- */
static int scalar_diag(const int8_t *profile, unsigned len, const uint8_t *db) {   /* scalarDiagonalScoring */
    int max = 0, score = 0;
    for (unsigned pos = 0; pos < len; pos++) {
        score += profile[pos * 21 + db[pos]];
        score = score < 0 ? 0 : score;
        max = score > max ? score : max;
    }
    return max;
}
static int single_seq_score(const int8_t *profile, unsigned qLen, const uint8_t *db, unsigned dbLen, int diagonal, unsigned minDist) {
    int max = 0;                                                                   /* computeSingelSequenceScores */
    if (diagonal >= 0 && minDist < qLen) {
        unsigned len = dbLen < qLen - minDist ? dbLen : qLen - minDist;
        max = scalar_diag(profile + minDist * 21, len, db);
    } else if (diagonal < 0 && minDist < dbLen) {
        unsigned len = dbLen - minDist < qLen ? dbLen - minDist : qLen;
        max = scalar_diag(profile, len, db + minDist);
    }
    return max;
}
/* uncapped score of (target, 16-bit diagonal); sequences >= 32768 are outside the supported domain */
static int diag_score(const fko *h, const int8_t *profile, unsigned qLen, uint32_t id, uint16_t diagonal) {
    const uint8_t *db = h->lookup + h->lookOff[id];
    unsigned dbLen = (unsigned) (h->lookOff[id + 1] - h->lookOff[id]);
    uint16_t d1 = (uint16_t) (0 - diagonal), d2 = diagonal;
    unsigned minDist = d1 < d2 ? d1 : d2;
    return single_seq_score(profile, qLen, db, dbLen, (int) (short) diagonal, minDist);
}
static void align_counters(const fko *h, const int8_t *profile, unsigned qLen, Counter *r, size_t n) {   /* UngappedAlignment::align */
    for (size_t i = 0; i < n; i++) {
        if (r[i].count != 0) continue;
        int s = diag_score(h, profile, qLen, r[i].id, r[i].diagonal);
        r[i].count = (uint8_t) (s > 255 ? 255 : s);
    }
}

/* SubstitutionMatrix::calcLocalAaBiasCorrection against the k-mer matrix: the pinned restatement in fs_oracle.c */
void fso_comp_bias(const int16_t *sub, const double *pBack, int n, const uint8_t *seq, int N, float scale, float *out);
static void comp_bias(const fko *h, const uint8_t *q, int N, float scale, float *out) {
    fso_comp_bias(h->kmerSub, h->pback, 21, q, N, scale, out);
}

static unsigned pick_bins(const fko *h) {   /* QueryMatcher::initDiagonalMatcher (QueryMatcher.cpp:460-488) */
    if (h->p.bins) return (unsigned) h->p.bins;
    const uint64_t l2 = h->p.l2CacheSize ? h->p.l2CacheSize : 262144;
    for (unsigned x = 2; x <= 1024; x *= 2) if ((uint64_t) h->n / x < l2) return x;
    return 2048;
}

static int cmp_hit(const void *a, const void *b) {   /* hit_t::compareHitsByScoreAndId */
    const fko_hit *x = (const fko_hit *) a, *y = (const fko_hit *) b;
    int ax = abs(x->score), ay = abs(y->score);
    if (ax != ay) return ax > ay ? -1 : 1;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return 0;
}

static size_t radix_by_score(const unsigned *sizes, Counter *w, unsigned thr, const Counter *r, size_t n) {   /* radixSortByScoreSize */
    Counter *ptr[256];
    Counter *prev = w + n;
    for (int i = 0; i < 256; i++) { ptr[i] = prev - sizes[i]; prev = ptr[i]; }
    size_t above = 0;
    for (size_t i = 0; i < n; i++) if (r[i].count >= thr) { above++; *ptr[r[i].count]++ = r[i]; }
    return above;
}

/* QueryMatcher::matchQuery.  q: numeric query codes (mask flag ignored).  identity: target id of the query itself or -1.
 * stats[0..3] = kmersPerPos, dbMatches, diagonalOverflow, bins.  Returns the number of hits (<= maxResListLen),
 * or -1 when the reference would take its unstable std::sort branch (resultSize >= foundDiagonalsSize/2). */
int fko_query(void *hv, const uint8_t *qcodes, int L, int64_t identity, fko_hit *out, double *stats) {
    fko *h = (fko *) hv;
    const fko_params *p = &h->p;
    const size_t big = h->n > 1000000 ? (size_t) h->n : 1000000;
    const size_t foundSize = p->foundDiagonalsSize ? (size_t) p->foundDiagonalsSize : big;
    const size_t maxDbMatches = p->maxDbMatches ? (size_t) p->maxDbMatches : big * 2;
    const size_t maxHits = (size_t) p->maxResListLen < (size_t) h->n ? (size_t) p->maxResListLen : (size_t) h->n;
    uint8_t *q = (uint8_t *) malloc(L + 1);
    for (int i = 0; i < L; i++) q[i] = qcodes[i] >= 32 ? (uint8_t) (qcodes[i] - 32) : qcodes[i];
    float *bias = (float *) calloc(L + 1, sizeof(float));
    if (p->compBias) comp_bias(h, q, L, p->compBiasScale, bias);
    /* UngappedAlignment::createProfile */
    int8_t *profile = (int8_t *) calloc((size_t) 21 * (L + 1), 1);
    for (int pos = 0; pos < L; pos++) {
        float c = bias[pos];
        c = (c < 0.0) ? c / 4 - 0.5 : c / 4 + 0.5;
        const char corr = (char) c;
        for (int a = 0; a < 21; a++) profile[pos * 21 + a] = (int8_t) (h->ungSub[q[pos] * 21 + a] + corr);
    }
    Cfo cfo;
    cfo_init(&cfo, pick_bins(h), (size_t) h->n);
    Counter *hits = (Counter *) malloc(maxDbMatches * sizeof(Counter));          /* databaseHits with the diagonal resolved */
    Counter *found = (Counter *) calloc(foundSize * 2 + 16, sizeof(Counter));
    KmerGen g;
    kmergen_init(&g, h);

    /* ---- QueryMatcher::match -------------------------------------------------------------------------- */
    size_t kmerListLen = 0, numMatches = 0, overflowNumMatches = 0, overflowHitCount = 0, nh = 0, hitCount = 0;
    int overflow = 0, aborted = 0, unsupported = 0;
    uint8_t kmer[16];
    for (int cur = 0; !aborted; cur++) {
        int r = kmer_at(h, q, L, cur, kmer);
        if (r == 0) break;
        float bc = 0;
        for (int i = 0; i < p->kmerSize; i++) bc += bias[cur + h->kpos[i]];
        if (r == 2) continue;
        short b = (short) ((bc < 0.0) ? bc - 0.5 : bc + 0.5);
        int kms = p->kmerThr - b;
        short thr = (short) (kms > 0 ? kms : 0);
        size_t len;
        const uint64_t *list = kmer_list(&g, h, kmer, thr, &len);
        kmerListLen += len;
        for (size_t kp = 0; kp < len; kp++) {
            const uint64_t o0 = h->offsets[list[kp]], sz = h->offsets[list[kp] + 1] - o0;
            if (nh + sz >= maxDbMatches) {
                overflow = 1;
                const size_t hc = p->noDiagScore ? cfo_find_duplicates_total(&cfo, hits, nh, found + overflowHitCount, foundSize - overflowHitCount)
                                                 : cfo_find_duplicates(&cfo, hits, nh, found + overflowHitCount, foundSize - overflowHitCount);
                if (overflowHitCount != 0 && p->noDiagScore) {
                    overflowHitCount = cfo_merge_score(&cfo, found, hc + overflowHitCount);      /* QueryMatcher.cpp:328-332 */
                } else if (overflowHitCount != 0) {
                    overflowHitCount = cfo_merge_diag_keep_scored(&cfo, found, hc + overflowHitCount);
                    align_counters(h, profile, (unsigned) L, found, overflowHitCount);
                    overflowHitCount = cfo_keep_max(&cfo, found, overflowHitCount);
                } else {
                    overflowHitCount = hc;
                }
                nh = 0;
                overflowNumMatches += numMatches;
                numMatches = 0;
                if (nh + sz >= maxDbMatches) { aborted = 1; break; }
            }
            for (uint64_t e = 0; e < sz; e++) {
                hits[nh].id = h->entries[o0 + e].seqId;
                hits[nh].diagonal = (uint16_t) (cur - h->entries[o0 + e].pos);
                hits[nh].count = 0;
                nh++;
            }
            numMatches += sz;
        }
    }
    if (unsupported) { kmergen_free(&g); cfo_free(&cfo); free(hits); free(found); free(profile); free(bias); free(q); return -3; }
    if (numMatches > 0) {
        hitCount = p->noDiagScore ? cfo_find_duplicates_total(&cfo, hits, nh, found + overflowHitCount, foundSize - overflowHitCount)
                                  : cfo_find_duplicates(&cfo, hits, nh, found + overflowHitCount, foundSize - overflowHitCount);
        if (overflowHitCount != 0) hitCount = p->noDiagScore ? cfo_merge_score(&cfo, found, overflowHitCount + hitCount)
                                                             : cfo_merge_diag(&cfo, found, overflowHitCount + hitCount);
    }
    if (stats) { stats[0] = (double) kmerListLen / (double) L; stats[1] = (double) (overflowNumMatches + numMatches); stats[2] = overflow; stats[3] = cfo.B; }
    if (p->noDiagScore) {
        /* ---- matchQuery without diagonal scoring (QueryMatcher.cpp:215-232): histogram of the counts (match(), :344-348), cut, radix
         * order by count, getResult<KMER_SCORE> (the query itself with UCHAR_MAX), final order by (score, id) ---- */
        int nr = 0;
        unsigned sz[256];
        memset(sz, 0, sizeof(sz));
        for (size_t i = 0; i < hitCount; i++) sz[found[i].count]++;
        unsigned t;
        { size_t fh = 0; for (t = 255; t > 0; t--) { fh += sz[t]; if (fh >= maxHits) break; } }
        if (t < (unsigned) p->minDiagScoreThr) t = (unsigned) p->minDiagScoreThr;
        if (hitCount >= foundSize / 2) nr = -1;               /* std::sort branch (:222-231): not modelled, like the diagonal-score mode */
        else {
            Counter *w2 = found + hitCount;
            const size_t cnt = radix_by_score(sz, w2, t, found, hitCount);
            size_t cur = 0;
            if (identity >= 0) { out[0].id = (uint32_t) identity; out[0].score = 255; out[0].diag = 0; out[0].pad = 0; cur = 1; }
            for (size_t i = 0; i < cnt && cur < maxHits; i++)
                if (w2[i].count >= t && (identity < 0 || (uint32_t) identity != w2[i].id)) {
                    out[cur].id = w2[i].id; out[cur].score = w2[i].count; out[cur].diag = w2[i].diagonal; out[cur].pad = 0;
                    cur++;
                }
            if (cur > 1) {
                if (identity >= 0) qsort(out + 1, cur - 1, sizeof(fko_hit), cmp_hit);
                else qsort(out, cur, sizeof(fko_hit), cmp_hit);
            }
            nr = (int) cur;
        }
        kmergen_free(&g); cfo_free(&cfo); free(hits); free(found); free(profile); free(bias); free(q);
        return nr;
    }

    /* ---- matchQuery post-processing (diagonalScoring, amino acids) ------------------------------------- */
    int nres = 0;
    size_t resultSize = hitCount;
    align_counters(h, profile, (unsigned) L, found, resultSize);
    unsigned sizes[256];
    Counter *rd = found, *wr;
    resultSize = cfo_keep_max(&cfo, rd, resultSize);
    wr = found + resultSize;
    memset(sizes, 0, sizeof(sizes));
    for (size_t i = 0; i < resultSize; i++) sizes[rd[i].count]++;
    unsigned thr;
    { size_t fh = 0; for (thr = 255; thr > 0; thr--) { fh += sizes[thr]; if (fh >= maxHits) break; } }   /* computeScoreThreshold */
    if (thr < (unsigned) p->minDiagScoreThr) thr = (unsigned) p->minDiagScoreThr;
    if (resultSize >= foundSize / 2) { nres = -1; goto done; }
    {
        const int truncated = thr >= 255;
        size_t cnt = radix_by_score(sizes, wr, thr, rd, resultSize);
        { Counter *t = rd; rd = wr; wr = t; }
        int rescale = 0;
        unsigned useThr = thr;
        if (truncated) {
            /* rescoreHits (QueryMatcher.cpp:563-589) */
            memset(sizes, 0, sizeof(sizes));
            int maxSelf = single_seq_score(profile, (unsigned) L, q, (unsigned) L, 0, 0);
            maxSelf -= 255;
            if (maxSelf < 1) maxSelf = 1;
            if (maxSelf > 65535) maxSelf = 65535;
            const float fmax = (float) maxSelf;
            size_t elements = 0;
            for (size_t i = 0; i < cnt && rd[i].count >= 255; i++) {
                unsigned ns = (unsigned) diag_score(h, profile, (unsigned) L, rd[i].id, rd[i].diagonal);
                ns -= 255;
                float sc = (float) (ns < 65535u ? ns : 65535u);
                rd[i].count = (uint8_t) ((sc / fmax) * (float) 255 + 0.5);
                sizes[rd[i].count] += 1;
                elements++;
            }
            cnt = radix_by_score(sizes, wr, 0, rd, elements);
            { Counter *t = rd; rd = wr; wr = t; }
            rescale = maxSelf;
            useThr = 0;
        }
        /* getResult<UNGAPPED_DIAGONAL_SCORE> */
        size_t cur = 0;
        if (identity >= 0) { out[0].id = (uint32_t) identity; out[0].score = 65535; out[0].diag = 0; out[0].pad = 0; cur = 1; }
        for (size_t i = 0; i < cnt && cur < maxHits; i++) {
            if (rd[i].count >= useThr && (identity < 0 || (uint32_t) identity != rd[i].id)) {
                out[cur].id = rd[i].id; out[cur].score = rd[i].count; out[cur].diag = rd[i].diagonal; out[cur].pad = 0;
                if (rescale != 0) out[cur].score = (int32_t) (255u + ((unsigned) rd[i].count * (unsigned) rescale / 255u));
                else if (rd[i].count >= 255) out[cur].score = diag_score(h, profile, (unsigned) L, rd[i].id, rd[i].diagonal);
                cur++;
            }
        }
        if (cur > 1) {
            if (identity >= 0) qsort(out + 1, cur - 1, sizeof(fko_hit), cmp_hit);
            else qsort(out, cur, sizeof(fko_hit), cmp_hit);
        }
        nres = (int) cur;
    }
done:
    kmergen_free(&g);
    cfo_free(&cfo);
    free(hits); free(found); free(profile); free(bias); free(q);
    return nres;
}
