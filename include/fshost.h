/*
 * fshost.h -- host-side (CPU) half of the hot path, C ABI.
 *
 * Everything here is cheap per-query or per-hit work that the reference also does on the CPU and that must be
 * bit-identical for the device kernels' inputs/outputs to mean the same thing: substitution matrices, composition
 * bias, query profiles, the e-value network, hit gating and the result text format.  It lives in libfsgpu.so next
 * to the device entry points of fsgpu.h; none of it runs a DP on the CPU.
 *
 * Reference code each function mirrors (M/ = lib/mmseqs, F/ = repository root of the reference):
 *   fshost_matrix_*            SubstitutionMatrix ctor / readProbMatrix      M/src/commons/SubstitutionMatrix.cpp:12-58,317-421
 *                              BaseMatrix::generateSubMatrix                 M/src/commons/BaseMatrix.cpp:96-159
 *   fshost_comp_bias           SubstitutionMatrix::calcLocalAaBiasCorrection M/src/commons/SubstitutionMatrix.cpp:79-109
 *   fshost_prefilter_profile   runFilterOnGpu profile + ssw_init bias        M/src/prefiltering/ungappedprefilter.cpp:195-203,
 *                                                                            M/src/alignment/StripedSmithWaterman.cpp:1375-1406
 *   fshost_align_profiles      StructureSmithWaterman::ssw_init              F/src/commons/StructureSmithWaterman.cpp:1556-1640
 *   fshost_evaluer_*           EvalueNeuralNet                               F/src/strucclustutils/EvalueNeuralNet.{h,cpp}
 *   fshost_search_*            runFilterOnGpu + structurealign per-query body M/src/prefiltering/ungappedprefilter.cpp:41-326,
 *                                                                            F/src/strucclustutils/structurealign.cpp:318-452
 */
#ifndef FSHOST_H
#define FSHOST_H

#include <stddef.h>
#include <stdint.h>
#include "fsgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fshost_matrix fshost_matrix;
typedef struct fshost_evaluer fshost_evaluer;

enum { FSHOST_MAT_3DI = 0, FSHOST_MAT_BLOSUM62 = 1 };

/* built-in matrices (the parameter data compiled in from foldseek_amd/data/fs_params.h) */
fshost_matrix *fshost_matrix_create(int which, float bitFactor, float scoreBias);
/* user matrix in the .out text format (--sub-mat); NULL on parse error */
fshost_matrix *fshost_matrix_from_text(const char *text, float bitFactor, float scoreBias);
/* from an already scaled matrix: scores[n * n] (row-major, BaseMatrix::subMatrix) and background[n] (BaseMatrix::pBack) as the reference's
 * SubstitutionMatrix object holds them -- what an adapter inside the reference has at hand.  No letters: numeric codes only. */
fshost_matrix *fshost_matrix_from_scores(const int16_t *scores, int n, const double *pBack);
void fshost_matrix_free(fshost_matrix *m);
int fshost_matrix_size(const fshost_matrix *m);                 /* alphabet size incl. X (21) */
/* the parameter file of a built-in matrix as text (what a precomputed index stores next to the matrix name, BaseMatrix::serialize,
 * M/src/commons/BaseMatrix.cpp:170-187); NULL for matrices without one.  *len = bytes without terminator. */
const char *fshost_matrix_text(int which, size_t *len);
const int16_t *fshost_matrix_scores(const fshost_matrix *m);    /* [n*n] bit-scaled substitution scores */
const double *fshost_matrix_background(const fshost_matrix *m); /* [n] BaseMatrix::pBack */
/* ASCII -> numeric codes with the reference's letter mapping (lower case = same letter; J->L, U/O->X, Z->E, B->D) */
void fshost_matrix_encode(const fshost_matrix *m, const char *ascii, int len, uint8_t *codes);
char fshost_matrix_letter(const fshost_matrix *m, int code);

void fshost_comp_bias(const fshost_matrix *m, const uint8_t *seq, int L, float scale, float *out);
void fshost_round_bias(const float *cb, int L, int8_t *out);

/* pssm: int8 [n][L]; scoreCap = 255 - bias of the CPU uint8 kernel.  compBias != 0 enables the correction. */
int fshost_prefilter_profile(const fshost_matrix *m3di, const uint8_t *q3di, int L, int compBias, float scale,
                             int8_t *pssm, int *scoreCap);

/* pAA/p3Di: int16 [n][L] linear word profiles; cbAA/cbSS (int8 [L], may be NULL) receive the rounded biases.
 * Both biases are computed against mAA (reference quirk, StructureSmithWaterman.cpp:1565,1570). */
int fshost_align_profiles(const fshost_matrix *mAA, const fshost_matrix *m3Di, const uint8_t *qAA, const uint8_t *q3Di,
                          int L, int compBias, float scale3Di, int16_t *pAA, int16_t *p3Di, int8_t *cbAA, int8_t *cbSS);

/* ---- k-mer prefilter query side (QueryMatcher::matchQuery :108-124 + match :262-270, UngappedAlignment::createProfile) ----
 * mKmer = SubstitutionMatrix(3di.out, 8.0, -0.2) (the seeding matrix), mUngapped = SubstitutionMatrix(3di.out, 2.0, -0.2).
 * kmerThr[i], i in [0, L - patternSize]: max(kmerThrBase - round(sum of bias over the k-mer's residues), 0);
 * profile[L][21] = mUngapped[q_i][a] + round(bias_i / 4).  patternSize/kmerSize as in fsgpu_kmer_index_params.
 * Returns the number of k-mer start positions written (0 when the query is shorter than the pattern). */
int fshost_kmer_query_prepare(const fshost_matrix *mKmer, const fshost_matrix *mUngapped, const uint8_t *q3di, int L,
                              int compBias, float scale, int kmerThrBase, int kmerSize, int spaced,
                              int16_t *kmerThr, int8_t *profile);
/* Prefiltering::getKmerThreshold for sequence-sequence searches (Prefiltering.cpp:1036-1096, externalThreshold of
 * F/src/FoldseekBase.cpp:585): -s sensitivity -> k-mer score threshold; < 0 for an unsupported k */
int fshost_kmer_threshold(float sensitivity, int kmerSize);

/* nnPath NULL: evalue_nn.bin next to libfsgpu.so (foldseek_amd/data/) */
fshost_evaluer *fshost_evaluer_create(const char *nnPath, uint64_t dbResidues);
void fshost_evaluer_free(fshost_evaluer *e);
void fshost_predict_mu_lambda(const fshost_evaluer *e, const uint8_t *q3di, unsigned int L, int alphabetSize,
                              double *lambda, double *mu);
double fshost_evalue_corr(const fshost_evaluer *e, double score, double lambda, double mu);

/* ---- one query through prefilter + structurealign (what bench.py times and the module executables call) ---- */
typedef struct {
    /* prefilter (ungappedprefilter defaults as set by Foldseek: F/src/workflow/StructureSearch.cpp:101-107) */
    int maxResListLen;        /* --max-seqs, 1000 */
    int minDiagScoreThr;      /* --min-ungapped-score, 30 */
    int compBiasCorrection;   /* --comp-bias-corr, 1 */
    float prefCompBiasScale;  /* 0.15 */
    /* structurealign (F/src/commons/LocalParameters.cpp:382-421) */
    int alignmentType;        /* 0: 3Di only, 2: 3Di+AA */
    float alnCompBiasScale;   /* 0.5 */
    int gapOpen, gapExtend;   /* 10, 1 */
    double evalThr;           /* -e, 10 */
    float covThr;             /* -c, 0.0 */
    int covMode;              /* 0 */
    int addBacktrace;         /* -a */
    int maxAccept, maxRejected; /* INT_MAX */
    float seqIdThr;           /* --min-seq-id 0 */
    int alnLenThr;            /* --min-aln-len 0 */
    int seqIdMode;            /* --seq-id-mode 0: identities / alignment length, 1: / shorter, 2: / longer sequence (Util::computeSeqId) */
    int altAlignment;         /* --alt-ali 0: up to this many alternative alignments per accepted hit (structurealign.cpp:115-138,415-429) */
    int skipUndefinedDiagonals; /* structurerescorediagonal: 0 = fail on a pair whose reference result is undefined (FSGPU_DIAG_UNDEFINED / NO_OVERLAP), 1 = drop it */
} fshost_params;

void fshost_params_default(fshost_params *p);

/* == Matcher::result_t minus the strings (M/src/alignment/Matcher.h:32-49) */
typedef struct {
    uint32_t dbKey;
    int32_t score;
    float qcov, dbcov, seqId;
    double eval;
    uint32_t alnLength;
    int32_t qStartPos, qEndPos;
    uint32_t qLen;
    int32_t dbStartPos, dbEndPos;
    uint32_t dbLen;
    uint32_t backtraceOff, backtraceLen;  /* into the cigar buffer of the search handle */
} fshost_result;

typedef struct fshost_search fshost_search;

/* Binds a device context (with a loaded DB) to matrices/e-value network; keys[n] maps target index -> DB key
 * (NULL: identity).  data3di/dataAA/offsets/lengths are the host buffers that were given to fsgpu_db_load (the
 * caller's mmap'd padded DB): like the reference's Marv handle (ungappedprefilter.cpp:124-158) they stay owned by
 * the caller and must outlive the handle; the backtrace of accepted hits reads target residues from them.
 * The handle owns its matrices; ctx is borrowed. */
fshost_search *fshost_search_create(fsgpu_ctx *ctx, const fshost_params *p, const uint32_t *keys, const char *nnPath,
                                    const uint8_t *data3di, const uint8_t *dataAA, const uint64_t *offsets,
                                    const int32_t *lengths);
void fshost_search_free(fshost_search *s);
const char *fshost_search_error(const fshost_search *s);

/* Prefilter one query: fills hits (capacity maxResListLen) sorted like the reference; returns count or < 0. */
int fshost_search_prefilter(fshost_search *s, const uint8_t *q3di, int L, int64_t identityId, fsgpu_hit *hits);
/* Prefilter nq queries with as few scan launches as their lengths allow (fsgpu_gapless_scan_multi): hits[q * maxResListLen ...],
 * nhits[q]; identityId may be NULL.  Same hit lists as nq fshost_search_prefilter calls.  Returns 0 or < 0. */
int fshost_search_prefilter_batch(fshost_search *s, int nq, const uint8_t *const *q3di, const int *L, const int64_t *identityId,
                                  fsgpu_hit *hits, int *nhits);
/* Align one query against a hit list (target indices); results (capacity n * (1 + altAlignment)) sorted like structurealign writes them.
 * Returns number of accepted alignments or < 0. */
int fshost_search_align(fshost_search *s, const uint8_t *qAA, const uint8_t *q3di, int L, int64_t identityId,
                        const uint32_t *targetIds, int n, fshost_result *results);
/* The same for nq queries with one forward and one reversed device pass (fsgpu_sw_multi_dir): results[q] has room for n[q] * (1 + altAlignment) entries, nres[q] receives
 * the number of accepted alignments of query q; identityId may be NULL.  Returns 0 or < 0. */
int fshost_search_align_batch(fshost_search *s, int nq, const uint8_t *const *qAA, const uint8_t *const *q3di, const int *L,
                              const int64_t *identityId, const uint32_t *const *targetIds, const int *n,
                              fshost_result *const *results, int *nres);
/* The per-batch body of the fused `search` module with the k-mer prefilter, in ONE call: query profiles and k-mer thresholds
 * (fshost_kmer_query_prepare), the device prefilter (fsgpu_kmer_search: QueryMatcher::matchQuery, M/src/prefiltering/QueryMatcher.cpp:103-240),
 * runSplit's coverage pre-filter of its hits (Util::canBeCovered for --cov-mode 0 / 2 / 5, M/src/prefiltering/Prefiltering.cpp:880-887) and the
 * structure alignment of what is left (fshost_search_align_batch: F/src/strucclustutils/structurealign.cpp:284-452).  mKmer / mUngapped as for
 * fshost_kmer_query_prepare.  prefIdentity[q]: target id of the query itself for the prefilter's self hit (-1: none); alnIdentity[q]: the index
 * structurealign compares with the target's (-1: none).  Outputs, cap = sp->maxResListLen per query: hits [nq * cap] with nhits[q] prefilter hits
 * (before the coverage pre-filter), status[q] (FSGPU_KMER_*), keptIds [nq * cap] with nkept[q] ids handed to the aligner, results
 * [nq * cap * (1 + altAlignment)] with nres[q] accepted alignments (backtraces through fshost_search_backtrace until the next call), seconds[4]:
 * host wall time of prepare / prefilter call / coverage pre-filter / align call.  A query whose status is < 0 gets no alignments.  Returns 0 or < 0. */
int fshost_search_kmer_batch(fshost_search *s, const fshost_matrix *mKmer, const fshost_matrix *mUngapped, const fsgpu_kmer_search_params *sp,
                             int kmerThr, int spaced, int nq, const uint8_t *const *qAA, const uint8_t *const *q3di, const int *L,
                             const int64_t *prefIdentity, const int64_t *alnIdentity, fsgpu_kmer_hit *hits, int32_t *nhits, int32_t *status,
                             uint32_t *keptIds, int32_t *nkept, fshost_result *results, int32_t *nres, double *seconds);
/* structurerescorediagonal for nq queries (F/src/strucclustutils/structurerescorediagonal.cpp:50-156,300-370): targetIds / diagonals
 * are the first and third column of the prefilter lines in their order; results[q] has room for n[q] entries; gates, e-value,
 * seq. id, ordering and (with addBacktrace) the all-match backtrace as the module writes them. */
int fshost_search_rescore_diagonal_batch(fshost_search *s, int nq, const uint8_t *const *qAA, const uint8_t *const *q3di, const int *L,
                                         const int64_t *identityId, const uint32_t *const *targetIds, const int16_t *const *diagonals,
                                         const int *n, fshost_result *const *results, int *nres);
/* alignStartPosBacktrace for a sequence query (F/src/commons/StructureSmithWaterman.cpp:540-739, banded_sw :1723-1957, computerBacktrace
 * :746-773): the reverse pass runs on the device (same SW kernels, reversed prefixes), the banded trace-back on the
 * host.  The reference reaches this function for profile queries and in a fall-back that never fires (structurealign.cpp:83-100);
 * it is exported for callers that want the SSW-style start / CIGAR instead of the block aligner's.  (qEnd, dbEnd, score): forward
 * alignScoreEndPos result.  Returns 1 + outputs, 0 if the reverse pass does not reproduce `score` or the trace-back fails, < 0 on error. */
int fshost_search_startpos_backtrace(fshost_search *s, const uint8_t *qAA, const uint8_t *q3di, int L, uint32_t targetId, int qEnd, int dbEnd,
                                     int score, int *qStart, int *dbStart, unsigned int *identicalAA, char *backtrace, size_t btCap);
const char *fshost_search_backtrace(const fshost_search *s, const fshost_result *r);
/* Host wall time (seconds) spent in the stages of the last prefilter/align calls: [0] prefilter profile build,
 * [1] fsgpu_gapless_scan incl. wait, [2] align profiles + e-value net, [3] fsgpu_sw_batch incl. wait, [4] gates,
 * [5] block-aligner backtrace; [6..7] reserved. */
void fshost_search_stats(const fshost_search *s, double *out8);
/* last fshost_search_align_batch (also behind fshost_search_kmer_batch): the accepted hits whose start position and backtrace came from the device
 * block aligner (fsgpu_block_backtrace) and all of them; the difference took the host aligner (blocks beyond 128 rows, FSGPU_DEVICE_BACKTRACE=0) */
void fshost_search_backtrace_counts(const fshost_search *s, int64_t *onDevice, int64_t *all);
/* Host worker pool for the per-hit backtraces (block aligner, ~20 us each): a feeder thread hands the accepted pairs of its
 * batch to the pool and takes part itself.  n = 0: the calling threads do everything themselves.  Default:
 * FSGPU_HOST_WORKERS or min(6, usable cores - 1), "usable" = hardware threads capped by the cgroup CPU quota. */
void fshost_set_host_workers(int n);
int fshost_host_workers(void);
int fshost_usable_cores(void);
/* Raw per-pair device results of the last fshost_search_align (n entries each), for tests. */
void fshost_search_last_sw(const fshost_search *s, const fsgpu_swres **fwd, const fsgpu_swres **rev);

/* Start position + backtrace of one accepted hit on the host (alignStartPosBacktraceBlock,
 * F/src/commons/StructureSmithWaterman.cpp:369-537): qAA/q3Di/cbAA/cbSS describe the forward query (codes + rounded
 * composition biases from fshost_align_profiles), tAA/t3Di the unmasked target, (qEnd, dbEnd, score) the forward
 * alignScoreEndPos result.  Returns 1 and fills qStart/dbStart/identicalAA/backtrace (capacity btCap, NUL terminated)
 * when the block aligner reproduces `score`, 0 when it does not (the reference then reports start -1 and no backtrace). */
int fshost_block_backtrace(const fshost_matrix *mAA, const fshost_matrix *m3Di, const uint8_t *qAA, const uint8_t *q3Di,
                           const int8_t *cbAA, const int8_t *cbSS, int Lq, const uint8_t *tAA, const uint8_t *t3Di, int Lt,
                           int qEnd, int dbEnd, int score, int gapOpen, int gapExtend, int *qStart, int *dbStart,
                           unsigned int *identicalAA, char *backtrace, size_t btCap);

/* banded_sw + computerBacktrace (F/src/commons/StructureSmithWaterman.cpp:1723-1957, 746-773) on the host for a pair whose start AND end
 * cells are known: CIGAR string (M / I / D, start -> end) of the rectangle [qStart, qEnd] x [dbStart, dbEnd], band |dbLen - qLen| + 1 doubled
 * until `score` is reached.  q* / t* / cb* point at position 0 of the sequences.  Returns 1, 0 (trace-back failed) or < 0. */
int fshost_banded_backtrace(const fshost_matrix *mAA, const fshost_matrix *m3Di, const uint8_t *qAA, const uint8_t *q3Di, const int8_t *cbAA,
                            const int8_t *cbSS, const uint8_t *tAA, const uint8_t *t3Di, int qStart, int qEnd, int dbStart, int dbEnd, int score,
                            int gapOpen, int gapExtend, unsigned int *identicalAA, char *backtrace, size_t btCap);

/* text formats: QueryMatcher::prefilterHitToBuffer (QueryMatcher.h:120-132), Matcher::resultToBuffer (Matcher.cpp:282) */
size_t fshost_format_prefilter_hit(char *buf, uint32_t key, int score, int diagonal);
size_t fshost_format_result(char *buf, const fshost_result *r, const char *backtrace, int addBacktrace);

/* ---- module entry points with the reference's sub-command contract (argv without the program / command name) ----
 * They read and write MMseqs2 databases on disk exactly like the reference modules they stand in for:
 *   fsmod_ungappedprefilter  <queryDB_ss> <targetDB_ss> <outPrefDB>       M/src/prefiltering/ungappedprefilter.cpp:484-595
 *   fsmod_prefilter          <queryDB_ss> <targetDB_ss> <outPrefDB>       M/src/prefiltering/Prefiltering.cpp:22-245,755-982 (k-mer prefilter)
 *   fsmod_structurealign     <queryDB> <targetDB> <prefDB> <outAlnDB>     F/src/strucclustutils/structurealign.cpp:141-481
 *   fsmod_makepaddedseqdb    <seqDB> <outPaddedDB>                        M/src/util/makepaddedseqdb.cpp:14-154
 * Return EXIT_SUCCESS or print a message to stderr and return EXIT_FAILURE. */
int fsmod_ungappedprefilter(int argc, const char **argv);
int fsmod_prefilter(int argc, const char **argv);
int fsmod_search(int argc, const char **argv);   /* <queryDB> <targetDB> <outAlnDB> [<outPrefDB>]: prefilter + structurealign fused */
int fsmod_structurealign(int argc, const char **argv);
int fsmod_makepaddedseqdb(int argc, const char **argv);
/* structurerescorediagonal <queryDB> <targetDB> <prefDB> <outAlnDB>   F/src/strucclustutils/structurerescorediagonal.cpp:159-393 */
int fsmod_structurerescorediagonal(int argc, const char **argv);
/* convertalis <queryDB> <targetDB> <alnDB> <outFile>   F/src/strucclustutils/structureconvertalis.cpp:253-1445: the BLAST-tab family
 * (--format-mode 0 / 2 / 4) with every --format-output column that is a function of the alignment record, the sequences and the
 * headers (+ the set columns from <db>.lookup / <db>.source); columns that need C-alpha coordinates, taxonomy or multimer data are refused by name.  Host only (text formatting). */
int fsmod_convertalis(int argc, const char **argv);
/* gpuserver <targetDB_ss[_pad]>: keeps the target DB resident in HBM and serves gapless scans over the reference's
 * shared-memory protocol until SIGINT/SIGTERM (M/src/util/gpuserver.cpp:24-101, M/src/commons/GpuUtil.h:9-49);
 * `ungappedprefilter ... --gpu-server 1` is the client (M/src/prefiltering/ungappedprefilter.cpp:71-122,208-257). */
int fsmod_gpuserver(int argc, const char **argv);
/* indexdb <seqDB> <seqDB> [--index-subset N] [--index-dbsuffix S] ...: writes <seqDB>.idx, the precomputed index of
 * PrefilteringIndexReader::createIndexFile (M/src/prefiltering/PrefilteringIndexReader.cpp:53-307; module M/src/util/indexdb.cpp:43-215): sequence and
 * header databases, sequence lookup and -- unless --index-subset has bit 2 -- the k-mer table and the extended 2-/3-mer matrices, taken from the index
 * this library builds on the device and renumbered to the reference's k-mer order.  Read by the reference's CPU prefilter / structurealign and by the
 * modules above.  createindex <seqDB> <tmpDir> = F/data/structureindex.sh: indexdb on <seqDB> (no k-mer table) and <seqDB>_ss (with it), header links,
 * C-alpha database appended when present. */
int fsmod_indexdb(int argc, const char **argv);
int fsmod_createindex(int argc, const char **argv);
/* protocol constants, exported so that tests (and a reference build) can check them:
 * name of the block for a database (Util::hash of realpath + visible devices + version, GpuUtil.cpp:18-34),
 * its size (GpuUtil.h:34-39) and the header layout: out = {sizeof, offsets of maxSeqLen, maxResListLen, state, serverExit,
 * queryOffset, queryLen, resultsOffset, resultLen, profileOffset, sizeof(Marv::Result)} */
int fshost_gpu_shm_name(const char *db, const char *visibleDevices, const char *version, char *out, size_t cap);
size_t fshost_gpu_shm_bytes(unsigned int maxSeqLen, unsigned int maxResListLen);
void fshost_gpu_shm_layout(unsigned int out[11]);

#ifdef __cplusplus
}
#endif
#endif
