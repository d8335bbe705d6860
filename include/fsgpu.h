/*
 * fsgpu.h -- C ABI of the MI355X-native Foldseek hot path (prefilter + structure alignment).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types, never throws, never exits.
 * Every entry point returns 0 on success or a negative FSGPU_E_* code; fsgpu_last_error() gives the text.
 * One context per GPU and per host thread (the reference's single-caller Marv contract,
 * M/src/prefiltering/ungappedprefilter.cpp:124-158,207).
 *
 * Reference interfaces these replace (M/ = lib/mmseqs of the reference tree, F/ = its root):
 *   fsgpu_create / fsgpu_destroy      Marv::Marv / ~Marv                      M/lib/libmarv/src/marv.h:6-27
 *   fsgpu_db_load                     Marv::loadDb + Marv::setDb              M/lib/libmarv/src/marv.h:37-43,
 *                                                                             call site ungappedprefilter.cpp:150-155
 *   fsgpu_db_adopt_device             (none: the reference has no collective; lets the caller hand over a
 *                                      buffer that an RCCL broadcast already placed in HBM)
 *   fsgpu_gapless_scan                Marv::scan                              M/lib/libmarv/src/marv.h:45-51,
 *                                     + the filter/sort tail of runFilterOnCpu ungappedprefilter.cpp:450-470
 *                                     (score semantics = SmithWaterman::ungapped_alignment,
 *                                      M/src/alignment/StripedSmithWaterman.cpp:1817-1876, i.e. capped at 255-bias)
 *   fsgpu_sw_batch                    StructureSmithWaterman::alignScoreEndPos x2 (forward query, reversed query)
 *                                     F/src/commons/StructureSmithWaterman.cpp:263-362, call sites
 *                                     F/src/strucclustutils/structurealign.cpp:46-47,65-67
 *   fsgpu_kmer_index_build            Prefiltering::getIndexTable -> IndexBuilder::fillDatabase + the extended 3-mer matrix
 *                                     M/src/prefiltering/Prefiltering.cpp:544-583,220-225, IndexBuilder.cpp:56-271
 *   fsgpu_kmer_search                 the per-query body of Prefiltering::runSplit = QueryMatcher::matchQuery
 *                                     M/src/prefiltering/Prefiltering.cpp:847-917, QueryMatcher.cpp:103-376
 *   fsgpu_sw_multi / _multi_dir       the same for a batch of queries, one device launch per register class and direction
 *                                     (forward over all pairs, reversed over the pairs alignStructure still needs:
 *                                     F/src/strucclustutils/structurealign.cpp:50-65)
 *   fsgpu_db_broadcast                replication of the resident DB over the GPUs of a node (RCCL), SURVEY 8e
 *   fshost_*                          host-side pieces of the same path that stay on the CPU, exported so the
 *                                     reference-side adapter and the tests can reach them.
 */
#ifndef FSGPU_H
#define FSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSGPU_ALPHABET 21          /* 20 states + X, order ACDEFGHIKLMNPQRSTVWYX (M/src/commons/DBReader.cpp:353-361) */
#define FSGPU_MAX_SEQ_LEN 65535    /* Parameters::maxSeqLen default, M/src/commons/Parameters.cpp:2496 */

enum {
    FSGPU_OK = 0,
    FSGPU_E_ARG = -1,        /* bad argument */
    FSGPU_E_HIP = -2,        /* a HIP runtime call failed (text in fsgpu_last_error) */
    FSGPU_E_NODB = -3,       /* no database loaded / AA part missing */
    FSGPU_E_UNSUPPORTED = -4,/* parameter combination not implemented on the device path */
    FSGPU_E_NOMEM = -5
};

typedef struct fsgpu_ctx fsgpu_ctx;

/* == Marv::Result without the unused end positions (M/lib/libmarv/src/marv.h:16-28): id = target index in the
 * loaded DB (the caller maps it to a key with DBReader::getDbKey), score = gapless diagonal score. */
typedef struct {
    uint32_t id;
    int32_t score;
} fsgpu_hit;

/* == the fields of StructureSmithWaterman::s_align that alignScoreEndPos fills (StructureSmithWaterman.h:84-109):
 * score1, qEndPos1, dbEndPos1 and `word` (1: int16 pass final, 2: int32 re-run because score1 hit INT16_MAX). */
typedef struct {
    int32_t score;
    int32_t qEnd;
    int32_t dbEnd;
    int32_t word;
} fsgpu_swres;

/* ---- context ---------------------------------------------------------------------------------------------- */
int fsgpu_create(int device, fsgpu_ctx **out);
void fsgpu_destroy(fsgpu_ctx *ctx);
const char *fsgpu_last_error(const fsgpu_ctx *ctx); /* ctx may be NULL: returns the last creation error */
/* A second context on the same GPU with its own HIP stream and scratch buffers that SHARES the resident database of
 * `src` (reference counted): lets several host threads keep the device busy, the way the reference runs one aligner
 * object per OpenMP thread over one shared DBReader (F/src/strucclustutils/structurealign.cpp:284-321). */
int fsgpu_clone(const fsgpu_ctx *src, fsgpu_ctx **out);
/* One node, several GPUs, one process: replicates the resident database of `src` into the n contexts dst[] created on
 * OTHER devices, one broadcast per buffer over RCCL (xGMI; librccl is loaded on demand) or peer copies when RCCL is
 * not available (FSGPU_NO_RCCL=1 forces that).  *usedRccl (may be NULL) reports which.  Queries then shard over the
 * devices with no further communication (the reference's multi-GPU path shards TARGETS per device and
 * merges top-N lists, M/lib/libmarv/src/cudasw4.cuh:1477-1553 -- not needed when the DB fits every GPU's 288 GB). */
int fsgpu_db_broadcast(fsgpu_ctx *src, fsgpu_ctx **dst, int n, int *usedRccl);
/* librccl alone on this context's device: a one-rank communicator broadcasts 1 MiB in place on the context's stream.  FSGPU_OK when the
 * library loaded and the call sequence of fsgpu_db_broadcast's RCCL branch ran; what a one-GPU box can verify of the replication path. */
int fsgpu_rccl_selfcheck(fsgpu_ctx *ctx);
int fsgpu_device(const fsgpu_ctx *ctx);
int fsgpu_device_count(void);      /* visible HIP devices, 0 when there is none */
/* HIP stream all kernels of this context are launched on (a hipStream_t), for callers that time with events */
void *fsgpu_stream(const fsgpu_ctx *ctx);

/* ---- target database -------------------------------------------------------------------------------------- */
/* data3di/dataAA: the *padded GPU database* byte buffers exactly as makepaddedseqdb writes them
 * (M/src/util/makepaddedseqdb.cpp:59-109): numeric codes 0..20, +32 when soft-masked, entries padded with 20.
 * offsets[n+1] in bytes (offsets[n] = end), lengths[n] = residues per entry (index length - 2).
 * dataAA may be NULL (prefilter only).  Pointers are borrowed for the duration of the call; the DB is copied to HBM
 * and re-tiled there (coalesced 8-target stripes for the scan, plain unmasked copies for the aligner). */
int fsgpu_db_load(fsgpu_ctx *ctx, const uint8_t *data3di, const uint8_t *dataAA,
                  const uint64_t *offsets, const int32_t *lengths, uint64_t n, uint64_t bytes);
/* Same, but the four buffers already live in this device's HBM (e.g. filled by an RCCL broadcast over xGMI).
 * The context does not take ownership of them and does not need them after the call returns. */
int fsgpu_db_adopt_device(fsgpu_ctx *ctx, const void *d_data3di, const void *d_dataAA,
                          const void *d_offsets, const void *d_lengths, uint64_t n, uint64_t bytes);
uint64_t fsgpu_db_size(const fsgpu_ctx *ctx);      /* entries */
uint64_t fsgpu_db_residues(const fsgpu_ctx *ctx);  /* sum of lengths */

/* ---- prefilter: exhaustive gapless diagonal scan ----------------------------------------------------------- */
/* pssm: int8 [21][L] row-major, pssm[a*L+i] = subMat[a][q_i] + round(compBias_i) -- what runFilterOnGpu hands to
 *       Marv::scan (ungappedprefilter.cpp:195-203).
 * scoreCap: 255 - bias of the CPU path (StripedSmithWaterman.cpp:1397-1406), 0..255; scores are
 *       min(cap, best diagonal run), which is exactly what the uint8 striped kernel returns.
 * Keeps targets with score > minScore (plus identityId if >= 0), orders them by (score desc, id asc)
 * (hit_t::compareHitsByScoreAndId, QueryMatcher.h:38-48) and returns the first maxRes in out[0..*nout). */
int fsgpu_gapless_scan(fsgpu_ctx *ctx, const int8_t *pssm, int L, int scoreCap, int minScore,
                       int64_t identityId, int maxRes, fsgpu_hit *out, int *nout);
/* The same for nq queries with as few device launches as their lengths allow (queries of equal ceil(L / 16) share one
 * launch; the workgroups of one query move in as those of the previous one drain, so the device stays full across
 * queries and a launch's duration is a per-launch figure worth quoting).  out[q * maxRes ...] / nout[q] per query. */
typedef struct {
    const int8_t *pssm;     /* int8 [21][L], as for fsgpu_gapless_scan */
    int32_t L;
    int32_t scoreCap;
    int64_t identityId;     /* target id that is always kept, or -1 */
} fsgpu_gapless_query;
int fsgpu_gapless_scan_multi(fsgpu_ctx *ctx, const fsgpu_gapless_query *q, int nq, int minScore, int maxRes, fsgpu_hit *out, int *nout);
/* scans / queries of the last fsgpu_gapless_scan_multi: one per launch of the device batch plus one per row-tiled query (> 896 residues, run on
 * its own); fsgpu_last_kernel_ms(ctx, 0) then is the device time of all of them together */
int fsgpu_gapless_last_batch(const fsgpu_ctx *ctx, int *launches, int *queries);
/* raw scores of query `queryIndex` of the last fsgpu_gapless_scan_multi call (queries of <= 896 residues); for tests */
int fsgpu_gapless_scores_multi(fsgpu_ctx *ctx, int queryIndex, uint8_t *scores_out);
/* Raw per-target scores of the last scan (n bytes, already capped); for tests and statistics. */
int fsgpu_gapless_scores(fsgpu_ctx *ctx, uint8_t *scores_out);
/* Asynchronous halves of fsgpu_gapless_scan for callers that pipeline several queries / time the device part:
 * _launch enqueues profile upload + kernels on the context stream, _finish waits and post-processes. */
/* Host-only planning step of the scan (no device call): the work list for stripes of stripeLen[] 16-column chunks when a
 * column segment needs `overlap` warm-up chunks (= ceil(Lq / 16)) and `waves` waves pull from the queue.  items[i] =
 * stripe << 32 | split << 31 | firstChunk << 16 | endChunk, longest first; *cap = the cut length chosen (minimises
 * max(cut, (work + warm-up work) / waves)).  Returns the number of items (may exceed capacity; nothing beyond it is
 * written).  Exported so that the policy can be tested without a GPU. */
int64_t fsgpu_gapless_plan_items(const uint32_t *stripeLen, uint32_t nStripes, int overlap, double waves, uint64_t *items,
                                 uint64_t capacity, uint32_t *cap);
int fsgpu_gapless_launch(fsgpu_ctx *ctx, const int8_t *pssm, int L, int scoreCap, int minScore,
                         int64_t identityId, int maxRes);
int fsgpu_gapless_finish(fsgpu_ctx *ctx, fsgpu_hit *out, int *nout);

/* ---- alignment: dual-profile striped-semantics affine Smith-Waterman, score + end position ----------------- */
/* Profiles are the linear int16 word profiles of StructureSmithWaterman::ssw_init (StructureSmithWaterman.cpp:
 * 1566-1640): pAA[a*L+i] = matAA[a][qAA_i] + cbAA_i, p3Di[a*L+i] = mat3Di[a][q3Di_i] + cbSS_i, a in 0..20,
 * once for the forward query (_fwd) and once built from the reversed query (_rev, structurealign.cpp:345-347).
 * pAA_* may be NULL when the AA matrix is all zero (--alignment-type 0).
 * For every targetIds[k] the call returns alignScoreEndPos(...) of the forward profile in fwd[k] and of the
 * reversed-query profile in rev[k], bit-identical to the AVX2 striped kernels including the int16->int32 re-run.
 * Requires gapOpen > gapExtend >= 0 (Foldseek: 10/1); otherwise FSGPU_E_UNSUPPORTED. */
int fsgpu_sw_batch(fsgpu_ctx *ctx, const int16_t *pAA_fwd, const int16_t *p3Di_fwd,
                   const int16_t *pAA_rev, const int16_t *p3Di_rev, int L,
                   const uint32_t *targetIds, int n, int gapOpen, int gapExtend,
                   fsgpu_swres *fwd, fsgpu_swres *rev);
/* The same for n EXPLICIT target sequences instead of database entries: tAA / t3Di hold unmasked codes 0..20, entry k at
 * offsets[k] with lengths[k] residues (offsets[n] = total bytes).  structurealign --alt-ali re-aligns a target whose
 * previous alignment range was overwritten with X (F/src/strucclustutils/structurealign.cpp:115-138, 415-429); that
 * sequence exists nowhere in the database.  tAA may be NULL when pAA_* are. */
int fsgpu_sw_batch_seqs(fsgpu_ctx *ctx, const int16_t *pAA_fwd, const int16_t *p3Di_fwd, const int16_t *pAA_rev,
                        const int16_t *p3Di_rev, int L, const uint8_t *tAA, const uint8_t *t3Di, const uint64_t *offsets,
                        const int32_t *lengths, int n, int gapOpen, int gapExtend, fsgpu_swres *fwd, fsgpu_swres *rev);
/* Several queries per call (what a host thread of structurealign would do one after the other, structurealign.cpp:322-452):
 * same semantics per query as fsgpu_sw_batch; results are concatenated in query order (sum of n entries).  All queries of
 * at most 512 residues that share a register class run in ONE launch, which fills the device where a single query's
 * ~1000 pairs cannot; longer queries run row-tiled, one launch per tile level for all of them.  Either all or none of the queries carry AA profiles. */
typedef struct {
    const int16_t *pAA_fwd, *p3Di_fwd, *pAA_rev, *p3Di_rev;
    int32_t L;
    int32_t n;
    const uint32_t *targetIds;
} fsgpu_sw_query;
int fsgpu_sw_multi(fsgpu_ctx *ctx, const fsgpu_sw_query *queries, int nq, int gapOpen, int gapExtend,
                   fsgpu_swres *fwd, fsgpu_swres *rev);
/* ONE direction of the same (dir 0: forward-query profiles, 1: reversed-query profiles) for a selection of the pairs:
 * sel[i][0..nsel[i]) are indices into q[i].targetIds (sel = nsel = NULL: every pair).  out has the layout of fwd[] above
 * (all pairs of all queries, concatenated); only the selected entries are written.  structurealign looks at the
 * reversed-query score only for pairs whose forward score passes the coverage and e-value gates
 * (F/src/strucclustutils/structurealign.cpp:55-65): run dir 0 over everything, gate on the host, run dir 1 over the
 * survivors.  Device side: two targets of one query share a wave (int16 halves), half the waves of fsgpu_sw_batch. */
int fsgpu_sw_multi_dir(fsgpu_ctx *ctx, const fsgpu_sw_query *q, int nq, int gapOpen, int gapExtend, int dir,
                       const int32_t *const *sel, const int32_t *nsel, fsgpu_swres *out);
/* The same with COMPACT queries.  A structurealign profile is nothing but a matrix column plus a position bias
 * (pAA[a*L+i] = matAA[a][qAA_i] + cbAA_i, StructureSmithWaterman.cpp:1566-1640), so a query is its codes (0..20) and its bias arrays --
 * for the forward query and for the reversed query (cb*_rev[i] belongs to position i of the REVERSED sequence; NULL = all zero) --
 * and the 21 x 21 int8 matrices are passed once: mat[a * 21 + b], matAA = NULL for --alignment-type 0 (then qAA / cbAA_* are not read).
 * The device builds its profile images itself: 6 L bytes per query cross the bus instead of 4 x 21 x L int16 words.  Results, selection
 * and the int16 -> int32 re-run are those of fsgpu_sw_multi_dir; a reversed call (dir 1) over the queries of the preceding forward call
 * finds the images in place.  Device side: queries of up to 512 residues run four targets per wave (32 lanes per target pair), up to
 * 1024 residues two per wave, longer ones row-tiled through the profile-based path. */
typedef struct {
    const uint8_t *qAA, *q3Di;
    const int8_t *cbAA_fwd, *cb3Di_fwd, *cbAA_rev, *cb3Di_rev;
    int32_t L;
    int32_t n;
    const uint32_t *targetIds;
} fsgpu_sw_cquery;
int fsgpu_sw_multi_dir_c(fsgpu_ctx *ctx, const int8_t *mat3Di, const int8_t *matAA, const fsgpu_sw_cquery *q, int nq, int gapOpen, int gapExtend,
                         int dir, const int32_t *const *sel, const int32_t *nsel, fsgpu_swres *out);
/* Both directions of every pair in ONE submission (one upload, one wait): for batches whose device time is small against a round trip --
 * all-vs-all steps with a handful of hits per query -- computing the reversed-query score of every pair is cheaper than gating on the
 * host between two passes.  fwd / rev as in fsgpu_sw_multi. */
int fsgpu_sw_multi_c(fsgpu_ctx *ctx, const int8_t *mat3Di, const int8_t *matAA, const fsgpu_sw_cquery *q, int nq, int gapOpen, int gapExtend,
                     fsgpu_swres *fwd, fsgpu_swres *rev);
/* Asynchronous halves of fsgpu_sw_batch.  The four profile arrays and targetIds are BORROWED until fsgpu_sw_finish returns
 * (the int32 re-run of saturated pairs reads the profiles again); after a failed _launch nothing is pending. */
int fsgpu_sw_launch(fsgpu_ctx *ctx, const int16_t *pAA_fwd, const int16_t *p3Di_fwd,
                    const int16_t *pAA_rev, const int16_t *p3Di_rev, int L,
                    const uint32_t *targetIds, int n, int gapOpen, int gapExtend);
int fsgpu_sw_finish(fsgpu_ctx *ctx, fsgpu_swres *fwd, fsgpu_swres *rev);

/* ---- single-diagonal rescoring: Foldseek's structurerescorediagonal (F/src/strucclustutils/structurerescorediagonal.cpp:23-102) ----
 * For every (query, target, diagonal) triple -- what linclust's kmermatcher hands on, prefilter-format lines
 * "targetKey score diagonal" -- the best ungapped run of sub3Di[q][t] + subAA[q][t] along that one diagonal for the forward
 * and for the reversed query.  Targets are entries of the resident database (AA half required), queries are given
 * explicitly: codes 0..20, query k at qOffsets[k] with qLengths[k] residues (qOffsets[nq] = bytes).  mat3Di / matAA: the
 * bit-scaled int16 matrices [21*21] (SubstitutionMatrix(3di.out, 2.1, 0) and (blosum62.out, 1.4 or 0.0, 0)). */
typedef struct {
    uint32_t query;         /* index into the query batch */
    uint32_t target;        /* index in the loaded DB */
    int32_t diagonal;       /* i - j as the prefilter prints it (int16 range) */
} fsgpu_diag_pair;
enum {
    FSGPU_DIAG_OK = 0,
    FSGPU_DIAG_NO_OVERLAP = 1,   /* |diagonal| beyond the sequence: the reference keeps LocalAlignment's defaults (-1, -1, 0) */
    FSGPU_DIAG_UNDEFINED = 2,    /* negative diagonal with a target longer than the query: the reference's reverse pass reads
                                    past the query (structurerescorediagonal.cpp:96-99); forward fields are valid, revScore is not */
    FSGPU_DIAG_BAD_ID = 3
};
typedef struct {
    int32_t score;          /* forward run */
    int32_t startPos, endPos;   /* along the diagonal, like DistanceCalculator::LocalAlignment */
    int32_t revScore;       /* reversed-query run (module score = score - revScore) */
    int32_t diagonalLen;
    int32_t identicalAA;    /* equal amino acids inside [startPos, endPos] */
    int32_t status;         /* FSGPU_DIAG_* */
    int32_t reserved;
} fsgpu_diag_res;
int fsgpu_diag_rescore(fsgpu_ctx *ctx, const uint8_t *qAA, const uint8_t *q3Di, const uint64_t *qOffsets, const int32_t *qLengths,
                       int nq, const int16_t *mat3Di, const int16_t *matAA, const fsgpu_diag_pair *pairs, int64_t n, fsgpu_diag_res *out);

/* ---- start position + backtrace of accepted hits on the device (rounds 5-6) ------------------------------------------------
 * StructureSmithWaterman::alignStartPosBacktraceBlock (F/src/commons/StructureSmithWaterman.cpp:369-537) over the block-aligner crate's align_3di
 * (M/lib/block-aligner/src/scan_block.rs:120-630, 1302-1443, 1844-2007) for a batch of hits: the reversed prefixes query[0..qEnd], target[0..dbEnd] are
 * aligned from their ends with starting block sizes 32, 64, 128 until the SW score is reproduced.  tblAA / tbl3Di: the crate's AAMatrix of the two
 * substitution matrices ([27][32] int8, letter-indexed, as block_set_aamatrix fills them: host/block_aligner.cpp, block_aamatrix_scores); letterAA /
 * letter3Di [21]: residue code -> letter - 'A'.  cbAA / cbSS: the query's forward composition bias (int8 per residue, StructureSmithWaterman.cpp:1566-1640).
 * status 1: qStart / dbStart / identicalAA and btLen characters ('M', 'I', 'D', query start to end) at *btBase + btOff (valid until the next call on this
 * context); status 2: the aligner's score differs from the SW score -- the reference leaves such a hit without start position and backtrace
 * (structurealign.cpp:83); status 0: not computed here -- the alignment needs a block of more than 128 rows and the second pass (blocks of up to 512 rows;
 * the crate grows to 4096) did not run for it (it runs when at least 64 such hits per usable host core have come back from the first pass, or with
 * FSGPU_BT_PASS2=1) or could not finish it either: the caller runs its host path (fshost_block_backtrace) -- same answer either way.
 * Needs a database loaded WITH AA sequences. */
typedef struct { const uint8_t *qAA, *q3Di; const int8_t *cbAA, *cbSS; int32_t L; int32_t reserved; } fsgpu_bt_query;
typedef struct { uint32_t query, target; int32_t qEnd, dbEnd, score; } fsgpu_bt_task;
typedef struct { int32_t status, qStart, dbStart, identicalAA, btLen, blockSizes; uint64_t btOff; } fsgpu_bt_res;
int fsgpu_block_backtrace(fsgpu_ctx *ctx, const int8_t *tblAA, const int8_t *tbl3Di, const uint8_t *letterAA, const uint8_t *letter3Di,
                          const fsgpu_bt_query *queries, int nq, const fsgpu_bt_task *tasks, int nt, int gapOpen, int gapExtend,
                          fsgpu_bt_res *res, const char **btBase);
/* Workgroups of the block aligner per compute unit for the following fsgpu_block_backtrace calls of this context (0: the default, 4 -- every CU's LDS
 * and three waves per SIMD for the 10-20 ms of a call).  1 leaves two thirds of a CU's LDS and most issue slots to the kernels of other contexts: what a
 * caller that shares a batch between its host threads and the device asks for (fshost_search_align_batch with FSGPU_DEVICE_BACKTRACE=2). */
int fsgpu_block_backtrace_footprint(fsgpu_ctx *ctx, int workgroupsPerCU);
/* Number of devices that hold at least one live context of this process (a module run with --gpus 8 : 8).  The host side divides the cores the process may
 * use by it when it decides who computes the backtraces (FSGPU_CORES_PER_GPU overrides the quotient: one rank of a multi-process job sets it to its share). */
int fsgpu_live_devices(void);

/* ---- prefilter: k-mer matching with double-diagonal hits + ungapped diagonal scoring ------------------------- */
/* Index parameters == the subset of Prefiltering's members that shape IndexTable / SequenceLookup.  Sequence-
 * sequence searches with k = 6 only (what setupSplit picks below 3.35e9 residues, IndexTable.h:456-458). */
typedef struct {
    int32_t kmerSize;       /* 6 */
    int32_t spaced;         /* 1: pattern 1101010011 (M/src/commons/Sequence.h:25), 0: contiguous */
    int32_t kmerThr;        /* Prefiltering::getKmerThreshold (Prefiltering.cpp:1036-1096); 78 at -s 9.5 */
    int32_t maskLowerCase;  /* --mask-lower-case (Foldseek: 1) */
    int32_t maskNrepeats;   /* --mask-n-repeat   (Foldseek: 6); tantan masking (--mask 1) is not available */
} fsgpu_kmer_index_params;
/* Builds, in HBM and from the resident 3Di database (fsgpu_db_load / fsgpu_db_adopt_device), the masked sequence
 * lookup, the k-mer index (offset table over 20^6 k-mers + one (seqId, first position) entry per distinct k-mer of
 * every target, lists ordered by seqId) and the sorted 8000 x 8000 extended 3-mer matrix of kmerSubMat21x21
 * (the 8-bit "k-mer" matrix of the prefilter: SubstitutionMatrix(3di.out, 8.0, -0.2), int16 row-major 21x21).
 * Limits (FSGPU_E_UNSUPPORTED beyond them): k = 6; fewer than 2^32 residues in the database (the reference switches to k = 7 from 3.35e9 residues
 * on, IndexTable.h:456-458, before that limit is reached); at most 33.5 M targets (512 coarse bins of the hit-stream partition, each inside one block of
 * 65536 target ids; 16.4 M until round 5); no target of 32768 residues or more.  Index entries are 4 bytes (seqId << posBits | position) while the id
 * bits and the position bits of the database fit 32 together, 8 bytes otherwise. */
int fsgpu_kmer_index_build(fsgpu_ctx *ctx, const fsgpu_kmer_index_params *p, const int16_t *kmerSubMat21x21);
uint64_t fsgpu_kmer_index_entries(const fsgpu_ctx *ctx);
/* Inspection (tests): copy the offset table (64e6+1 uint32), the entries (seqId << 16 | position) and/or the masked
 * sequence lookup (padded DB layout) to host memory; any pointer may be NULL.  The table is kept in the DEVICE k-mer
 * order: index = first3mer * 8000 + last3mer (the reference numbers k-mers first3mer + 8000 * last3mer).  Row `row` of the extended 3-mer matrix. */
int fsgpu_kmer_index_copy(fsgpu_ctx *ctx, uint32_t *offsets, uint64_t *entries, uint8_t *masked);
int fsgpu_kmer_row_copy(fsgpu_ctx *ctx, int row, int16_t *score, uint16_t *index);

typedef struct {
    int32_t maxResListLen;      /* --max-seqs */
    int32_t minDiagScoreThr;    /* --min-ungapped-score, >= 0 (0: a query with fewer than maxResListLen scored targets also gets the score-0 elements the reference hands on) */
    int32_t bins;               /* BINSIZE of CacheFriendlyOperations; 0 = derive from l2CacheSize like initDiagonalMatcher */
    int32_t kmerScoreOnly;      /* 1 = --diag-score 0 (QueryMatcher with diagonalScoring == false, QueryMatcher.cpp:215-232): no ungapped diagonal scores,
                                 * the score of a target is its number of double-diagonal k-mer matches (findDuplicates with computeTotalScore,
                                 * CacheFriendlyOperations.cpp:217-241; capped at 255), the diagonal that of its first one; the first step of
                                 * easy-cluster's cascaded prefilter runs like this (-s 1 --diag-score 0 --min-ungapped-score 0) */
    int64_t maxDbMatches;       /* 0 = 2*max(1e6,N) (QueryMatcher.cpp:45) */
    int64_t foundDiagonalsSize; /* 0 = max(1e6,N)   (QueryMatcher.cpp:44) */
    uint64_t l2CacheSize;       /* Util::getL2CacheSize() of the host whose tie order is to be reproduced; 0 = this host */
} fsgpu_kmer_search_params;
/* One query: numeric codes seq[L]; kmerThr[i] for every k-mer start i in [0, L - patternSize] =
 * max(kmerThr - round(sum of the composition bias over the k-mer), 0) (QueryMatcher.cpp:262-270);
 * profile[L][21] = ungappedSubMat[q_i][a] + round(bias_i / 4) (UngappedAlignment::createProfile).
 * fshost_kmer_query_prepare fills both.  identity = target id that is the query itself, or -1. */
typedef struct {
    const uint8_t *seq;
    const int16_t *kmerThr;
    const int8_t *profile;
    int32_t L;
    int32_t reserved;
    int64_t identity;
} fsgpu_kmer_query;
/* == hit_t (QueryMatcher.h:31-48) with the target's DB index instead of its key */
typedef struct {
    uint32_t id;
    int32_t score;
    uint16_t diagonal;
    uint16_t pad;
} fsgpu_kmer_hit;
enum {
    FSGPU_KMER_OK = 0,
    FSGPU_KMER_UNSTABLE = 1,     /* >= foundDiagonalsSize/2 targets carried a diagonal: the reference orders equal scores with an
                                    unstable std::sort there (QueryMatcher.cpp:205-215); replayed with the same call on the same
                                    sequence, identical wherever both builds use libstdc++'s introsort (informational) */
    FSGPU_KMER_E_OUTPUT = -1,    /* the reference would have cut findDuplicates short (output array full): replayed in the diagonal-score mode since round 4,
                                    still a status with kmerScoreOnly (and beyond 256 MB of per-bin counters for the flagged queries of a batch) */
    FSGPU_KMER_E_CHUNKS = -2,    /* more than 255 databaseHits refills */
    FSGPU_KMER_E_REFILL_COUNTS = -3  /* no longer returned (kept for the ABI): kmerScoreOnly queries that refill databaseHits have the reference's merge of the
                                        per-refill counts (mergeScoreDuplicates, CacheFriendlyOperations.cpp:150-180) replayed on the device since round 4 */
};
/* Runs nq queries (batched on the device), writes per query q up to maxResListLen hits to out[q*maxResListLen ..],
 * their number to nout[q] and a FSGPU_KMER_* code to status[q]; hits are bit-identical to QueryMatcher::matchQuery
 * (order included) whenever status[q] == 0.  stats (may be NULL): per query kmersPerPos, dbMatches, overflowed, bins. */
int fsgpu_kmer_search(fsgpu_ctx *ctx, const fsgpu_kmer_search_params *p, const fsgpu_kmer_query *queries, int nq,
                      fsgpu_kmer_hit *out, int32_t *nout, int32_t *status, double *stats);

/* Work counters of the last fsgpu_kmer_search batch: [0] similar k-mers probed in the index table, [1] index hits,
 * [2] double-diagonal candidates, [3] elements handed to the host tail. */
void fsgpu_kmer_last_counts(const fsgpu_ctx *ctx, uint64_t *out4);
/* Queries worth handing to the next fsgpu_kmer_search call of this context (a multiple of 32, 32..1024): as many as one device batch takes at
 * the index hits per query the previous call saw.  A low-sensitivity prefilter (clustering: ~10^4 hits per query) needs hundreds of queries
 * per batch to fill the device, the sensitive search default (~10^7 hits per query at 1M targets) fills it with 32. */
int fsgpu_kmer_batch_hint(const fsgpu_ctx *ctx);
/* Accounting of the last batch's hit-stream partition (round 6: one order-preserving scatter into (query, key) runs, a key = a run of blocks
 * of 1024 target ids): [1] = [4] (query, databaseHits chunk, key) runs walked by the duplicate stage, [3] tiles of the scatter, [5] coarse keys
 * of the level the batch used, [6] target ids of its widest key (16 bits of LDS each in k_kmer_dup_stream); [0] = [2] = 0. */
void fsgpu_kmer_last_segments(const fsgpu_ctx *ctx, uint32_t *out7);
/* The coarse keys of that partition (host-only planning, also used by the index build): consecutive blocks of 1024 target ids joined into keys.
 * blocksPerKey = 1..64: that many blocks per key; 0: about 128 keys of equal residue count, none longer than twice the average number of blocks
 * nor than 64.  key(t) = blkKey[t >> 10]; keyFirst[key] .. keyFirst[key + 1] are its target ids (at most 65536: the low 16 bits of an id are unique
 * inside a key).  Fills blkKey[max(1, ceil(n / 1024))] / keyFirst[keys + 1] (either may be NULL) and returns the number of keys (<= 512 up to
 * 33.5 M targets), or FSGPU_E_ARG when cap is too small. */
int fsgpu_kmer_plan_coarse(const int32_t *lengths, uint64_t n, uint32_t blocksPerKey, uint16_t *blkKey, uint32_t *keyFirst, uint32_t cap);

/* ---- instrumentation -------------------------------------------------------------------------------------- */
/* Device time (ms, HIP events on the context stream) of the dominant kernel of the last _finish()ed call:
 * which = 0 gapless scan kernel, 1 SW kernel, 2 whole device part of the last fsgpu_kmer_search batch,
 * 3..10 its stages (similar-k-mer count, index probes, hit gather, partition into (query, bin) segments, double-diagonal
 * detection per segment, scoring, replay, selection), 11 host tail, 12 the index-probe kernel (k_kmer_lists) alone.
 * Returns < 0 if nothing was recorded. */
double fsgpu_last_kernel_ms(const fsgpu_ctx *ctx, int which);
/* out[2][4], per direction (0 forward, 1 reversed query) of the last fsgpu_sw_multi_dir calls of this context: device ms of that pass's
 * k_sw2 launches (HIP events on the context stream; -1 when the pass did not run), DP cells (query rows x target columns over the
 * single-tile pairs), pairs, and the VALU wave-instructions its waves issue (DP rows + per-step overhead; the issue-rate roofline's unit). */
void fsgpu_sw_last_passes(const fsgpu_ctx *ctx, double *out);

#ifdef __cplusplus
}
#endif
#endif /* FSGPU_H */
