// fs_selmeta.h -- result header of the top-N selection passes (k_select.hpp); included inside namespace fs.
struct SelMeta {
    int32_t T;         // cut score: score > T always taken, score == T taken for the first `mTies` ids
    uint32_t nGt;      // number of hits with score > T
    uint32_t mTies;    // number of ties at T that are taken
    uint32_t nOut;     // nGt + mTies
};
