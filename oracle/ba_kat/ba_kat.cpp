// ba_kat.cpp -- TEST INFRASTRUCTURE ONLY.  Known-answer harness of the block aligner through its C ABI and nothing else
// (M/lib/block-aligner/c/block_aligner.h == foldseek_amd/csrc/host/block_aligner_abi.h for the functions used here).
// The same translation unit links against
//   (a) this repository's C++ restatement (make ours: foldseek_amd/csrc/host/block_aligner.cpp), which is what runs here, and
//   (b) the Rust crate itself (make crate CRATE=<path to lib/mmseqs/lib/block-aligner>: `cargo build --release --features simd_avx2`
//       of its C library) wherever a Rust toolchain exists,
// reads oracle/ba_kat/cases.txt and prints one line per case: name, score, query_idx, reference_idx, CIGAR, block sizes tried.
// tests/test_block_aligner.py compares the two outputs when oracle/ba_kat/crate.txt is present -- the only way to pin which of
// several co-optimal paths the crate's tie rules (OP_LUT, block trajectory) choose, which this image cannot do (no cargo).
//
// case file: blank-separated columns
//   3di  <name> <gapOpen> <gapExtend> <qAA> <q3Di> <qBias: comma list or -> <tAA> <t3Di>      the call sequence of
//        alignStartPosBacktraceBlock (F/src/commons/StructureSmithWaterman.cpp:369-537): min block size 32, 64, ... 4096 until the
//        x-drop alignment of the two (already reversed) strings reaches <target score>; <name> carries the target score after '@'
//   aa   <name> <gapOpen> <gapExtend> <minSize> <maxSize> <xdrop or -1 for the global variant> <q> <r>      Block::align with one matrix
//   nw   <name> <gapOpen> <gapExtend> <match> <mismatch> <minSize> <maxSize> <xdrop or -1> <q> <r>          the same with AAMatrix::new_simple
// matrices: mat_aa.txt / mat_3di.txt next to the case file (first line: letters, then one row of integers per letter).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#ifdef BA_KAT_CRATE_HEADER
#include "block_aligner.h"
#define BA_M_OP M
#else
#include "block_aligner_abi.h"
#endif

static AAMatrix *loadMatrix(const std::string &path) {
    std::ifstream f(path);
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    std::string letters;
    f >> letters;
    AAMatrix *m = block_new_simple_aamatrix(1, -1);
    for (size_t a = 0; a < letters.size(); a++)
        for (size_t b = 0; b < letters.size(); b++) { int v; f >> v; block_set_aamatrix(m, (uint8_t) letters[a], (uint8_t) letters[b], (int8_t) v); }
    return m;
}

static std::string cigarString(const Cigar *c) {
    static const char names[] = "?M=XID";
    std::string s;
    for (size_t i = 0; i < block_len_cigar(c); i++) { const OpLen o = block_get_cigar(c, i); s += std::to_string((size_t) o.len); s += names[o.op <= 5 ? o.op : 0]; }
    return s.empty() ? "-" : s;
}

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    AAMatrix *mAA = loadMatrix(dir + "/mat_aa.txt"), *m3 = loadMatrix(dir + "/mat_3di.txt");
    std::ifstream cases(dir + "/cases.txt");
    if (!cases) { fprintf(stderr, "cannot open %s/cases.txt\n", dir.c_str()); return 2; }
    const size_t MAX_SIZE = 4096;
    std::string line;
    while (std::getline(cases, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream in(line);
        std::string kind, name;
        int go, ge;
        in >> kind >> name >> go >> ge;
        Gaps gaps; gaps.open = (int8_t) -go; gaps.extend = (int8_t) -ge;
        if (kind == "3di") {
            std::string qa, q3, qb, ta, t3;
            in >> qa >> q3 >> qb >> ta >> t3;
            const int target = atoi(name.substr(name.find('@') + 1).c_str());
            std::vector<int16_t> qBias(qa.size(), 0), tBias(ta.size(), 0);
            if (qb != "-") { std::istringstream bs(qb); std::string tok; size_t i = 0; while (std::getline(bs, tok, ',') && i < qBias.size()) qBias[i++] = (int16_t) atoi(tok.c_str()); }
            BlockHandle blk = block_new_aa_trace_xdrop(qa.size() + 64, ta.size() + 64, MAX_SIZE);
            PaddedBytes *pqa = block_new_padded_aa(qa.size(), MAX_SIZE), *pq3 = block_new_padded_aa(qa.size(), MAX_SIZE);
            PaddedBytes *pta = block_new_padded_aa(ta.size(), MAX_SIZE), *pt3 = block_new_padded_aa(ta.size(), MAX_SIZE);
            PosBias *pqb = block_new_pos_bias(qa.size(), MAX_SIZE), *ptb = block_new_pos_bias(ta.size(), MAX_SIZE);
            block_set_bytes_padded_aa(pqa, (const uint8_t *) qa.data(), qa.size(), MAX_SIZE);
            block_set_bytes_padded_aa(pq3, (const uint8_t *) q3.data(), q3.size(), MAX_SIZE);
            block_set_bytes_padded_aa(pta, (const uint8_t *) ta.data(), ta.size(), MAX_SIZE);
            block_set_bytes_padded_aa(pt3, (const uint8_t *) t3.data(), t3.size(), MAX_SIZE);
            block_set_pos_bias(pqb, qBias.data(), qBias.size());
            block_set_pos_bias(ptb, tBias.data(), tBias.size());
            AlignResult res; res.score = -1000000000; res.query_idx = 0; res.reference_idx = 0;
            std::string sizes;
            size_t minSize = 32;
            while (minSize <= MAX_SIZE && res.score < target) {
                SizeRange range; range.min = minSize; range.max = MAX_SIZE;
                const int32_t xDrop = -((int32_t) minSize * gaps.extend + gaps.open);
                block_align_3di_aa_trace_xdrop(blk, pqa, pq3, pqb, pta, pt3, ptb, mAA, m3, gaps, range, xDrop);
                res = block_res_aa_trace_xdrop(blk);
                sizes += (sizes.empty() ? "" : ",") + std::to_string(minSize) + ":" + std::to_string(res.score);
                minSize *= 2;
            }
            Cigar *cg = block_new_cigar(qa.size(), ta.size());
            block_cigar_aa_trace_xdrop(blk, res.query_idx, res.reference_idx, cg);
            printf("%s\t%d\t%zu\t%zu\t%s\t%s\n", name.c_str(), res.score, (size_t) res.query_idx, (size_t) res.reference_idx, cigarString(cg).c_str(), sizes.c_str());
            block_free_cigar(cg);
            block_free_padded_aa(pqa); block_free_padded_aa(pq3); block_free_padded_aa(pta); block_free_padded_aa(pt3);
            block_free_pos_bias(pqb); block_free_pos_bias(ptb);
            block_free_aa_trace_xdrop(blk);
        } else if (kind == "aa" || kind == "nw") {
            size_t minSize, maxSize;
            int xdrop, match = 0, mismatch = 0;
            std::string q, r;
            if (kind == "nw") in >> match >> mismatch;          // Block::align with AAMatrix::new_simple(match, mismatch)
            in >> minSize >> maxSize >> xdrop >> q >> r;
            AAMatrix *mNw = kind == "nw" ? block_new_simple_aamatrix((int8_t) match, (int8_t) mismatch) : nullptr;
            const AAMatrix *mUse = mNw ? mNw : mAA;
            PaddedBytes *pq = block_new_padded_aa(q.size(), maxSize), *pr = block_new_padded_aa(r.size(), maxSize);
            block_set_bytes_padded_aa(pq, (const uint8_t *) q.data(), q.size(), maxSize);
            block_set_bytes_padded_aa(pr, (const uint8_t *) r.data(), r.size(), maxSize);
            SizeRange range; range.min = minSize; range.max = maxSize;
            Cigar *cg = block_new_cigar(q.size(), r.size());
            AlignResult res;
            if (xdrop >= 0) {
                BlockHandle blk = block_new_aa_trace_xdrop(q.size() + 64, r.size() + 64, maxSize);
                block_align_aa_trace_xdrop(blk, pq, pr, mUse, gaps, range, xdrop);
                res = block_res_aa_trace_xdrop(blk);
                block_cigar_aa_trace_xdrop(blk, res.query_idx, res.reference_idx, cg);
                block_free_aa_trace_xdrop(blk);
            } else {
                BlockHandle blk = block_new_aa_trace(q.size() + 64, r.size() + 64, maxSize);
                block_align_aa_trace(blk, pq, pr, mUse, gaps, range, 0);
                res = block_res_aa_trace(blk);
                block_cigar_aa_trace(blk, res.query_idx, res.reference_idx, cg);
                block_free_aa_trace(blk);
            }
            printf("%s\t%d\t%zu\t%zu\t%s\t-\n", name.c_str(), res.score, (size_t) res.query_idx, (size_t) res.reference_idx, cigarString(cg).c_str());
            block_free_cigar(cg);
            block_free_padded_aa(pq); block_free_padded_aa(pr);
            if (mNw) block_free_aamatrix(mNw);
        }
    }
    block_free_aamatrix(mAA); block_free_aamatrix(m3);
    return 0;
}
