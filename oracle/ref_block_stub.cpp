// ref_block_stub.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).
// Placeholder definitions of the block-aligner C ABI (M/lib/block-aligner/c/block_aligner.h) for the parts of the
// reference objects that are linked but not exercised (MMseqs' SmithWaterman constructor allocates through them;
// profile alignment entry points are never called on this path).  The entry points Foldseek's structurealign DOES
// call (block_align_3di_aa_trace_xdrop & co.) come from foldseek_amd/csrc/host/block_aligner.cpp when
// FS_HAVE_BLOCK_ALIGNER is defined; Rust is not available in this image, see DESIGN.md.
#include "block_aligner.h"
#include <cstdlib>
#include <cstdio>

static void die(const char *w) { fprintf(stderr, "oracle/_ref: %s is not available in this build\n", w); abort(); }

extern "C" {
// --- used only by MMseqs' SmithWaterman (StripedSmithWaterman.cpp ctor/dtor, ssw_init) ---
void block_set_bytes_padded_aa_numsequence(struct PaddedBytes *, const uint8_t *, uintptr_t, uintptr_t) { die("block_set_bytes_padded_aa_numsequence"); }
void block_set_all_gap_open_R_aaprofile(struct AAProfile *, int8_t) { die("aaprofile"); }
void block_set_all_gap_open_C_aaprofile(struct AAProfile *, int8_t) { die("aaprofile"); }
void block_set_all_gap_close_C_aaprofile(struct AAProfile *, int8_t) { die("aaprofile"); }
struct AAProfile *block_new_aaprofile(uintptr_t, uintptr_t, int8_t) { return (struct AAProfile *) calloc(64, 1); }
size_t block_get_curr_len_aaprofile(const struct AAProfile *) { die("aaprofile"); return 0; }
void block_free_aaprofile(struct AAProfile *p) { free(p); }
void block_align_profile_aa_trace_xdrop(BlockHandle, const struct PaddedBytes *, const struct AAProfile *, struct SizeRange, int32_t) { die("block_align_profile_aa_trace_xdrop"); }
void block_align_aa_trace_xdrop_posbias(BlockHandle, const struct PaddedBytes *, const struct PosBias *, const struct PaddedBytes *, const struct PosBias *, const struct AAMatrix *, struct Gaps, struct SizeRange, int32_t) { die("block_align_aa_trace_xdrop_posbias"); }
int8_t *aaprofile_pos_aa(struct AAProfile *) { die("aaprofile"); return NULL; }
int16_t *aaprofile_aa_pos(struct AAProfile *) { die("aaprofile"); return NULL; }

// --- used only by MMseqs' BlockAligner (the whole-binary build of oracle/build_ref_full.sh links it; the score-only
//     x-drop aligner is reached by `mmseqs align --alignment-mode 4`-style paths, never by the foldseek modules tested here) ---
BlockHandle block_new_aa_xdrop(uintptr_t, uintptr_t, uintptr_t) { return calloc(64, 1); }
void block_free_aa_xdrop(BlockHandle b) { free(b); }
void block_align_aa_xdrop_posbias(BlockHandle, const struct PaddedBytes *, const struct PosBias *, const struct PaddedBytes *, const struct PosBias *, const struct AAMatrix *, struct Gaps, struct SizeRange, int32_t) { die("block_align_aa_xdrop_posbias"); }
struct AlignResult block_res_aa_xdrop(BlockHandle) { die("block_res_aa_xdrop"); struct AlignResult r = {0, 0, 0}; return r; }

#ifndef FS_HAVE_BLOCK_ALIGNER
void block_set_pos_bias(struct PosBias *, const int16_t *, uintptr_t) {}
struct AlignResult block_res_aa_trace_xdrop(BlockHandle) { die("block_res_aa_trace_xdrop"); struct AlignResult r = {0, 0, 0}; return r; }
struct AAMatrix *block_new_simple_aamatrix(int8_t, int8_t) { return (struct AAMatrix *) calloc(2048, 1); }
struct PosBias *block_new_pos_bias(uintptr_t, uintptr_t) { return (struct PosBias *) calloc(64, 1); }
struct PaddedBytes *block_new_padded_aa(uintptr_t, uintptr_t) { return (struct PaddedBytes *) calloc(64, 1); }
struct Cigar *block_new_cigar(uintptr_t, uintptr_t) { return (struct Cigar *) calloc(64, 1); }
BlockHandle block_new_aa_trace_xdrop(uintptr_t, uintptr_t, uintptr_t) { return calloc(64, 1); }
uintptr_t block_len_cigar(const struct Cigar *) { die("cigar"); return 0; }
struct OpLen block_get_cigar(const struct Cigar *, uintptr_t) { die("cigar"); struct OpLen o = {Sentinel, 0}; return o; }
void block_free_pos_bias(struct PosBias *p) { free(p); }
void block_free_padded_aa(struct PaddedBytes *p) { free(p); }
void block_free_cigar(struct Cigar *p) { free(p); }
void block_free_aamatrix(struct AAMatrix *p) { free(p); }
void block_free_aa_trace_xdrop(BlockHandle b) { free(b); }
void block_cigar_aa_trace_xdrop(BlockHandle, uintptr_t, uintptr_t, struct Cigar *) { die("block_cigar_aa_trace_xdrop"); }
void block_set_bytes_padded_aa(struct PaddedBytes *, const uint8_t *, uintptr_t, uintptr_t) {}
void block_set_aamatrix_num(struct AAMatrix *, int8_t, int8_t, int8_t) {}
void block_set_aamatrix(struct AAMatrix *, uint8_t, uint8_t, int8_t) {}
void block_align_3di_aa_trace_xdrop(BlockHandle, const struct PaddedBytes *, const struct PaddedBytes *, const struct PosBias *, const struct PaddedBytes *, const struct PaddedBytes *, const struct PosBias *, const struct AAMatrix *, const struct AAMatrix *, struct Gaps, struct SizeRange, int32_t) { die("block_align_3di_aa_trace_xdrop"); }
#endif
}
