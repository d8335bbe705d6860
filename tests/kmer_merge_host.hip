// Test-side wrapper (built by tests/test_kmer_merge_model.py with hipcc, HOST code only -- no kernel is launched, no device is needed):
// runs the per-thread body of k_kmer_merge_heads (foldseek_amd/csrc/k_kmer.hpp), the replay of mergeScoreDuplicates for --diag-score 0 queries
// that refilled databaseHits, over a candidate array on the CPU, thread after thread, the way the device runs it in any order.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>
#include "k_kmer.hpp"

extern "C" int kmer_merge_heads_host(const uint32_t *ckeys, const uint64_t *cvals, uint64_t nCand, int tbits, int shift, uint32_t nChunks, const uint32_t *ec /*[kMaxChunks]*/,
                                     int reverse, int32_t *count /*[nCand] out: final count of run heads, -1 elsewhere*/, uint32_t *rounds /*[kMaxChunks]*/, uint64_t *resultSize) {
    using namespace fs;
    std::vector<KmerChunks> ck(1);
    ck[0].nChunks = nChunks;
    std::vector<uint64_t> scr(nCand + 1);
    std::vector<uint8_t> kept(nCand + 1);
    std::vector<int32_t> score(nCand + 1, -7);
    std::vector<KmerBest> best(nCand + 1);
    for (uint64_t i = 0; i < nCand; i++) best[i].nElems = 0x12345678u;        // every slot must be written by exactly the thread that owns it
    for (int i = 0; i < kMaxChunks; i++) rounds[i] = 0;
    *resultSize = 0;
    for (uint64_t t = 0; t < nCand; t++) {
        const uint64_t j = reverse ? nCand - 1 - t : t;                       // thread order must not matter
        kmerMergeHeadsThread(j, ckeys, cvals, nCand, tbits, shift, ck.data(), ec, scr.data(), kept.data(), score.data(), best.data(),
                             [&](uint32_t, uint32_t c, uint32_t n) { rounds[c] += n; }, [&](uint32_t, uint32_t n) { *resultSize += n; });
    }
    for (uint64_t i = 0; i < nCand; i++) {
        if (best[i].nElems == 0x12345678u) return 1;
        if (best[i].nElems == 0xFFFFFFFFu) count[i] = -1;
        else { if (best[i].cand != i || (best[i].nElems != 0) != (best[i].count != 0)) return 2; count[i] = (int32_t) best[i].count; }
    }
    return 0;
}
