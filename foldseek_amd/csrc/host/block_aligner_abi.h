// block_aligner_abi.h -- the subset of the block-aligner C ABI that Foldseek's structurealign uses
// (reference: M/lib/block-aligner/c/block_aligner.h, generated from M/lib/block-aligner/src/ffi.rs), re-declared here
// because the Rust crate cannot be built in this environment.  block_aligner.cpp implements it as a lane-exact C++
// restatement of the crate's AVX2 code path (L = 16 int16 lanes), so that the reference's own
// StructureSmithWaterman.cpp links against it unchanged in oracle/_ref.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum BaOperation { BA_Sentinel = 0, BA_M = 1, BA_Eq = 2, BA_X = 3, BA_I = 4, BA_D = 5 };

typedef struct AAMatrix AAMatrix;
typedef struct Cigar Cigar;
typedef struct PaddedBytes PaddedBytes;
typedef struct PosBias PosBias;
typedef void *BlockHandle;

typedef struct OpLen {
    uint8_t op;
    uintptr_t len;
} OpLen;
typedef struct Gaps {
    int8_t open;
    int8_t extend;
} Gaps;
typedef struct SizeRange {
    uintptr_t min;
    uintptr_t max;
} SizeRange;
typedef struct AlignResult {
    int32_t score;
    uintptr_t query_idx;
    uintptr_t reference_idx;
} AlignResult;

AAMatrix *block_new_simple_aamatrix(int8_t match_score, int8_t mismatch_score);
void block_set_aamatrix(AAMatrix *matrix, uint8_t a, uint8_t b, int8_t score);
void block_set_aamatrix_num(AAMatrix *matrix, int8_t a, int8_t b, int8_t score);
void block_free_aamatrix(AAMatrix *matrix);

Cigar *block_new_cigar(uintptr_t query_len, uintptr_t reference_len);
OpLen block_get_cigar(const Cigar *cigar, uintptr_t i);
uintptr_t block_len_cigar(const Cigar *cigar);
void block_free_cigar(Cigar *cigar);

PaddedBytes *block_new_padded_aa(uintptr_t len, uintptr_t max_size);
void block_set_bytes_padded_aa(PaddedBytes *padded, const uint8_t *s, uintptr_t len, uintptr_t max_size);
void block_free_padded_aa(PaddedBytes *padded);

PosBias *block_new_pos_bias(uintptr_t len, uintptr_t max_size);
void block_set_pos_bias(PosBias *bias, const int16_t *b, uintptr_t len);
void block_free_pos_bias(PosBias *bias);

BlockHandle block_new_aa_trace_xdrop(uintptr_t query_len, uintptr_t reference_len, uintptr_t max_size);
void block_align_3di_aa_trace_xdrop(BlockHandle b, const PaddedBytes *q, const PaddedBytes *q_3di, const PosBias *q_bias,
                                    const PaddedBytes *r, const PaddedBytes *r_3di, const PosBias *r_bias, const AAMatrix *m,
                                    const AAMatrix *m_3di, Gaps g, SizeRange s, int32_t x);
// single-matrix variant (Block::align); used by the known-answer tests taken from the crate's own unit tests
void block_align_aa_trace_xdrop(BlockHandle b, const PaddedBytes *q, const PaddedBytes *r, const AAMatrix *m, Gaps g, SizeRange s, int32_t x);
AlignResult block_res_aa_trace_xdrop(BlockHandle b);
void block_cigar_aa_trace_xdrop(BlockHandle b, uintptr_t query_idx, uintptr_t reference_idx, Cigar *cigar);
void block_cigar_eq_aa_trace_xdrop(BlockHandle b, const PaddedBytes *q, const PaddedBytes *r, uintptr_t query_idx, uintptr_t reference_idx, Cigar *cigar);
void block_free_aa_trace_xdrop(BlockHandle b);

// global (no X-drop) variants with traceback: Block<true,false>; only needed for the known-answer tests
BlockHandle block_new_aa_trace(uintptr_t query_len, uintptr_t reference_len, uintptr_t max_size);
void block_align_aa_trace(BlockHandle b, const PaddedBytes *q, const PaddedBytes *r, const AAMatrix *m, Gaps g, SizeRange s, int32_t x);
AlignResult block_res_aa_trace(BlockHandle b);
void block_cigar_aa_trace(BlockHandle b, uintptr_t query_idx, uintptr_t reference_idx, Cigar *cigar);
void block_cigar_eq_aa_trace(BlockHandle b, const PaddedBytes *q, const PaddedBytes *r, uintptr_t query_idx, uintptr_t reference_idx, Cigar *cigar);
void block_free_aa_trace(BlockHandle b);

// not in the crate's C header: the crate's Rust API Trace::blocks() (scan_block.rs:2009-2030) for either handle type -- the regions the
// last alignment computed, out[5 k ..] = {row, column, height, width, right}; returns their number (tests/ba_model.py compares trajectories)
uintptr_t block_trace_blocks(BlockHandle b, uint32_t *out, uintptr_t cap);
/* not in the crate's header either: the 27 x 32 int8 score table of an AAMatrix (scores.rs:44-70), what the device aligner (k_btrace.hpp) is handed */
const int8_t *block_aamatrix_scores(const AAMatrix *m);

#ifdef __cplusplus
}
#endif
