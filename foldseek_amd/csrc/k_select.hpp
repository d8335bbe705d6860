// k_select.hpp -- device-side top-N selection of prefilter hits with the reference's total order.
//
// Reference behaviour (runFilterOnCpu tail, M/src/prefiltering/ungappedprefilter.cpp:450-470): keep every target
// with score > minDiagScoreThr (or the identity hit), sort by (score desc, key asc) -- hit_t::compareHitsByScoreAndId,
// QueryMatcher.h:38-48 -- and truncate to maxResListLen.  Scores are bytes, so a 256-bin histogram finds the cut
// score T exactly; everything above T is emitted, and of the ties at T exactly the lowest ids (== lowest keys: the
// DB index is key sorted) that still fit.  Three tiny HBM-bound passes over n bytes; the host only sorts <= maxRes hits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fs {

constexpr int kSelChunk = 4096;     // targets per workgroup
constexpr int kSelThreads = 256;    // 16 consecutive targets per thread -> id order is preserved



__device__ __forceinline__ bool selPasses(int score, uint32_t id, int minScore, int64_t identityId) {
    return score > minScore || (int64_t) id == identityId;
}

// pass A: per-chunk histograms
// Batched form of all three passes: blockIdx.y = query of a multi-query scan; per-query arrays follow each other
// (scores: scoreStride bytes apart, histograms / bases: gridDim.x chunks apart, outputs: K apart); identityIds (may be
// nullptr) holds one identity id per query and overrides the scalar.
__global__ __launch_bounds__(kSelThreads) void k_sel_hist(const uint8_t *scores, uint32_t n, int minScore,
                                                          int64_t identityId, uint32_t *chunkHist, const int64_t *identityIds, uint64_t scoreStride) {
    __shared__ uint32_t h[256];
    scores += (size_t) blockIdx.y * scoreStride;
    chunkHist += (size_t) blockIdx.y * gridDim.x * 256;
    if (identityIds) identityId = identityIds[blockIdx.y];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSelChunk + threadIdx.x * 16;
    if (base < n) {
        uint32_t cnt = n - base < 16 ? n - base : 16;
        for (uint32_t k = 0; k < cnt; k++) {
            int s = scores[base + k];
            if (selPasses(s, base + k, minScore, identityId)) atomicAdd(&h[s], 1u);
        }
    }
    __syncthreads();
    chunkHist[(size_t) blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

// pass B (one workgroup): global histogram -> cut score, per-chunk output bases
__global__ __launch_bounds__(256) void k_sel_threshold(const uint32_t *chunkHist, uint32_t nChunks, uint32_t K,
                                                       SelMeta *meta, uint32_t *chunkBaseGt, uint32_t *chunkBaseTie) {
    __shared__ uint32_t hist[256];
    __shared__ SelMeta m;
    chunkHist += (size_t) blockIdx.x * nChunks * 256;          // one workgroup per query
    meta += blockIdx.x; chunkBaseGt += (size_t) blockIdx.x * nChunks; chunkBaseTie += (size_t) blockIdx.x * nChunks;
    uint32_t s = 0;
    for (uint32_t c = 0; c < nChunks; c++) s += chunkHist[(size_t) c * 256 + threadIdx.x];
    hist[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t cum = 0;
        int T = -1;
        uint32_t nGt = 0, mT = 0;
        bool found = false;
        for (int b = 255; b >= 0; b--) {
            if (cum + hist[b] >= K) { T = b; nGt = cum; mT = K - cum; found = true; break; }
            cum += hist[b];
        }
        if (!found) { T = -1; nGt = cum; mT = 0; }
        m.T = T; m.nGt = nGt; m.mTies = mT; m.nOut = nGt + mT;
        *meta = m;
    }
    __syncthreads();
    const int T = m.T;
    // per-chunk counts, then a serial exclusive scan (nChunks is n/4096: a few hundred)
    for (uint32_t c = threadIdx.x; c < nChunks; c += blockDim.x) {
        uint32_t gt = 0;
        for (int b = T + 1; b < 256; b++) gt += chunkHist[(size_t) c * 256 + b];
        chunkBaseGt[c] = gt;
        chunkBaseTie[c] = T >= 0 ? chunkHist[(size_t) c * 256 + T] : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ag = 0, at = 0;
        for (uint32_t c = 0; c < nChunks; c++) {
            uint32_t g = chunkBaseGt[c], t = chunkBaseTie[c];
            chunkBaseGt[c] = ag; chunkBaseTie[c] = at;
            ag += g; at += t;
        }
    }
}

// pass C: ordered emission
__global__ __launch_bounds__(kSelThreads) void k_sel_emit(const uint8_t *scores, uint32_t n, int minScore,
                                                          int64_t identityId, const SelMeta *meta,
                                                          const uint32_t *chunkBaseGt, const uint32_t *chunkBaseTie,
                                                          uint32_t *outId, int32_t *outScore, const int64_t *identityIds, uint64_t scoreStride,
                                                          uint32_t K) {
    __shared__ uint32_t sg[kSelThreads], st[kSelThreads];
    scores += (size_t) blockIdx.y * scoreStride;
    meta += blockIdx.y; chunkBaseGt += (size_t) blockIdx.y * gridDim.x; chunkBaseTie += (size_t) blockIdx.y * gridDim.x;
    outId += (size_t) blockIdx.y * K; outScore += (size_t) blockIdx.y * K;
    if (identityIds) identityId = identityIds[blockIdx.y];
    const SelMeta m = *meta;
    const uint32_t base = blockIdx.x * kSelChunk + threadIdx.x * 16;
    uint32_t cnt = 0;
    if (base < n) cnt = n - base < 16 ? n - base : 16;
    uint32_t lg = 0, lt = 0;
    for (uint32_t k = 0; k < cnt; k++) {
        int s = scores[base + k];
        if (!selPasses(s, base + k, minScore, identityId)) continue;
        lg += (s > m.T);
        lt += (s == m.T);
    }
    sg[threadIdx.x] = lg; st[threadIdx.x] = lt;
    __syncthreads();
    // exclusive scan over 256 threads (Hillis-Steele in LDS)
    for (int d = 1; d < kSelThreads; d <<= 1) {
        uint32_t vg = 0, vt = 0;
        if ((int) threadIdx.x >= d) { vg = sg[threadIdx.x - d]; vt = st[threadIdx.x - d]; }
        __syncthreads();
        sg[threadIdx.x] += vg; st[threadIdx.x] += vt;
        __syncthreads();
    }
    uint32_t pg = chunkBaseGt[blockIdx.x] + sg[threadIdx.x] - lg;
    uint32_t pt = chunkBaseTie[blockIdx.x] + st[threadIdx.x] - lt;
    for (uint32_t k = 0; k < cnt; k++) {
        int s = scores[base + k];
        if (!selPasses(s, base + k, minScore, identityId)) continue;
        if (s > m.T) {
            outId[pg] = base + k; outScore[pg] = s; pg++;
        } else if (s == m.T) {
            if (pt < m.mTies) { outId[m.nGt + pt] = base + k; outScore[m.nGt + pt] = s; }
            pt++;
        }
    }
}

} // namespace fs
