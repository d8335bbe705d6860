// k_kmer.hpp -- gfx950 kernels of the k-mer prefilter:
//   index build   : masking, k-mer extraction, (k-mer, seqId) de-duplication            IndexBuilder.cpp:56-271, IndexTable.h
//   3-mer table   : extended substitution matrix rows, stable-sorted                      ExtendedSubstitutionMatrix.cpp:20-69
//   search        : similar k-mers per query position (KmerGenerator.cpp:108-217), index gather into the hit stream
//                   (QueryMatcher.cpp:243-376), double-diagonal detection (CacheFriendlyOperations.cpp:188-283),
//                   diagonal scoring (UngappedAlignment.cpp:45-57,430-443), per-target replay of the overflow rounds
//                   (mergeElements / keepMaxScoreElementOnly), score histogram + cut (QueryMatcher.h:211-221).
//
// All of it is integer / byte work bound by HBM latency and bandwidth (random 8-byte index-table probes, 8-byte
// entry gathers, 12 B/hit through the radix sort); no MFMA.  The reference's arrival-order rules are kept exact by
// giving every hit its stream position g and sorting the hit stream by (query, target) with a STABLE radix sort,
// after which every order-dependent rule of the reference becomes a per-target, neighbour-only rule.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fs {

constexpr int kKA = 20;                       // seeding alphabet (X removed, Prefiltering.cpp:560-563)
constexpr int kRow3 = 8000;                   // 20^3
constexpr uint32_t kKmerInvalid = 0xFFFFFFFFu;
constexpr uint32_t kMaxKmerResult = 262144u * 32u;   // KmerGenerator.h:45
constexpr int kMaxChunks = 256;               // databaseHits refills per query we replay (8 bits in the hit payload)
constexpr int kKmerBlock = 256;

// hit payload (sort value): [g:40 | diag:16 | chunk:8]
__host__ __device__ inline uint64_t hitPack(uint64_t g, uint32_t diag, uint32_t chunk) { return (g << 24) | ((uint64_t) (diag & 0xffffu) << 8) | (chunk & 0xffu); }
__host__ __device__ inline uint64_t hitG(uint64_t v) { return v >> 24; }
__host__ __device__ inline uint32_t hitDiag(uint64_t v) { return (uint32_t) (v >> 8) & 0xffffu; }
__host__ __device__ inline uint32_t hitD8(uint64_t v) { return (uint32_t) (v >> 8) & 0xffu; }
__host__ __device__ inline uint32_t hitChunk(uint64_t v) { return (uint32_t) v & 0xffu; }

struct KmerPattern { int size; int pos[6]; };

// Device order of the 20^6 k-mer table: first 3-mer (letters 0..2 of the k-mer) MAJOR.  The reference's index is
// first3 + 8000*last3; KmerGenerator walks "for every similar first 3-mer x: all similar last 3-mers y", so with the
// first 3-mer major all probes of one x fall into one 32 KB row of the offset table (and one 1000-byte bitmap row)
// instead of being spread over the whole 256 MB table.
__host__ __device__ inline uint32_t kmerDeviceIndex(uint32_t first3, uint32_t last3) { return first3 * 8000u + last3; }

// per-query constants of one batch
struct KmerQ {
    uint32_t posBase, nPos;        // k-mer start positions [posBase, posBase+nPos) of the batch position arrays
    uint32_t seqOff, L;            // numeric query in the batch sequence buffer
    uint32_t profOff;              // int8 profile [L][21] in the batch profile buffer
    uint32_t pad;
    uint64_t hitBase;              // first hit of this query in the batch hit stream (filled by k_kq_bases)
    uint64_t listBase;             // first list slot of this query
};

struct KmerChunks {                // per query
    uint32_t nChunks;              // >= 1
    uint32_t aborted;              // a single list >= maxDbMatches (QueryMatcher.cpp:330-332)
    uint64_t total;                // hits of the query
    uint64_t abortKmers;           // aborted == 1: similar k-mers of the positions up to and including the aborting one
    uint64_t start[kMaxChunks + 1];// start[c] = first stream position of chunk c; start[nChunks] = total
};

// --------------------------------------------------------------------------------------------------------------
// index build
// --------------------------------------------------------------------------------------------------------------
// Masker::maskRepeats + lower-case masking + finalizeMasking, one thread per target (one-off, sequential rule)
__global__ void k_kmer_mask(const uint8_t *raw, const uint64_t *offsets, const int32_t *lengths, uint64_t n,
                            int maskLower, int maskNrepeats, uint8_t *out) {
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint8_t *src = raw + offsets[t];
    uint8_t *dst = out + offsets[t];
    const int L = lengths[t];
    for (int i = 0; i < L; i++) { uint8_t c = src[i]; c = c >= 32 ? c - 32 : c; dst[i] = c > 20 ? 20 : c; }
    if (maskNrepeats > 0) {
        unsigned repeatCount = 0;
        int startOfRepeat = -1;
        int previous = 0;                      // '\0': a leading run of code 0 is never masked (startOfRepeat stays -1)
        for (int pos = 0; pos < L; ++pos) {
            const int c = dst[pos];
            if (c == previous) {
                repeatCount++;
            } else {
                if (repeatCount > (unsigned) maskNrepeats && startOfRepeat >= 0)
                    for (int i = startOfRepeat; i < pos; ++i) dst[i] = 20;
                repeatCount = 1; startOfRepeat = pos; previous = c;
            }
        }
        if (repeatCount > (unsigned) maskNrepeats && startOfRepeat >= 0)
            for (int i = startOfRepeat; i < L; ++i) dst[i] = 20;
    }
    if (maskLower) for (int i = 0; i < L; i++) if (src[i] >= 32) dst[i] = 20;
    const int padded = (L + 3) & ~3;
    for (int i = L; i < padded; i++) dst[i] = 20;
}

// one (k-mer, seqId<<16|pos) pair per residue start position, in (seqId, pos) order; non-k-mers get kKmerInvalid.
// IndexTable::addKmerCount / addSequence filters: no X, self score >= kmerThr.
__global__ __launch_bounds__(256) void k_kmer_extract(const uint8_t *masked, const uint64_t *offsets, const int32_t *lengths,
                                                      const uint64_t *resOff, uint64_t n, KmerPattern pat, int kmerThr,
                                                      const int8_t *selfScore /*[21]*/, uint32_t *keys, uint64_t *vals) {
    for (uint64_t t = blockIdx.x; t < n; t += gridDim.x) {
        const uint8_t *s = masked + offsets[t];
        const int L = lengths[t];
        const uint64_t base = resOff[t];
        for (int pos = threadIdx.x; pos < L; pos += blockDim.x) {
            uint32_t key = kKmerInvalid;
            if (pos + pat.size <= L) {
                uint32_t half[2] = {0, 0}, pw = 1;
                int score = 0;
                bool x = false;
#pragma unroll
                for (int z = 0; z < 6; z++) {
                    const uint32_t c = s[pos + pat.pos[z]];
                    x |= c >= 20;
                    half[z / 3] += c * pw; pw = (z == 2) ? 1 : pw * kKA;
                    score += selfScore[c > 20 ? 20 : c];
                }
                if (!x && !(kmerThr > 0 && score < kmerThr)) key = kmerDeviceIndex(half[0], half[1]);
            }
            keys[base + pos] = key;
            vals[base + pos] = (t << 16) | (uint32_t) pos;
        }
    }
}

// after the stable sort by k-mer: keep the first (smallest position) entry of every (k-mer, seqId) run
__global__ void k_kmer_unique_flags(const uint32_t *keys, const uint64_t *vals, uint64_t n, uint32_t *flags, uint32_t *counts) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    uint32_t f = 0;
    if (k != kKmerInvalid) {
        f = (i == 0 || keys[i - 1] != k || (vals[i - 1] >> 16) != (vals[i] >> 16)) ? 1u : 0u;
        if (f) atomicAdd(&counts[k], 1u);
    }
    flags[i] = f;
}
__global__ void k_kmer_compact_entries(const uint64_t *vals, const uint32_t *flags, const uint32_t *scan, uint64_t n, uint64_t *entries) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) entries[scan[i]] = vals[i];
}

// one bit per k-mer: list non-empty (8 MB instead of 256 MB: most probes of the search never reach the offset table)
__global__ void k_kmer_bitmap(const uint32_t *offsets, uint32_t nWords, uint32_t *bitmap) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nWords) return;
    uint32_t bits = 0, prev = offsets[(size_t) w * 32];
#pragma unroll 8
    for (int b = 0; b < 32; b++) { const uint32_t nx = offsets[(size_t) w * 32 + b + 1]; bits |= (nx != prev ? 1u : 0u) << b; prev = nx; }
    bitmap[w] = bits;
}

// --------------------------------------------------------------------------------------------------------------
// extended 3-mer matrix: row a = all 8000 3-mers b sorted by score(a,b) descending, ties in the reference's
// permutation order (first position most significant).  Key = (hi - score) << 13 | permRank is unique, so an LDS
// bitonic sort of 8192 keys reproduces std::stable_sort exactly.  One workgroup per row.
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_kmer_rows3(const int16_t *sub /*21x21*/, int16_t *score, uint16_t *index) {
    __shared__ uint32_t key[8192];
    __shared__ int16_t sm[21 * 21];
    for (int i = threadIdx.x; i < 441; i += blockDim.x) sm[i] = sub[i];
    __syncthreads();
    const int a = blockIdx.x;                 // index = a0 + 20 a1 + 400 a2
    const int a0 = a % 20, a1 = (a / 20) % 20, a2 = a / 400;
    for (int r = threadIdx.x; r < 8192; r += blockDim.x) {
        uint32_t k = 0xFFFFFFFFu;
        if (r < kRow3) {                      // permutation rank r = b0*400 + b1*20 + b2
            const int b0 = r / 400, b1 = (r / 20) % 20, b2 = r % 20;
            const int s = sm[a0 * 21 + b0] + sm[a1 * 21 + b1] + sm[a2 * 21 + b2];
            k = ((uint32_t) (4096 - s) << 13) | (uint32_t) r;
        }
        key[r] = k;
    }
    __syncthreads();
    for (int k = 2; k <= 8192; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < 8192; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint32_t x = key[i], y = key[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { key[i] = y; key[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    for (int z = threadIdx.x; z < kRow3; z += blockDim.x) {
        const uint32_t k = key[z];
        const int r = (int) (k & 8191u);
        const int b0 = r / 400, b1 = (r / 20) % 20, b2 = r % 20;
        score[(size_t) a * kRow3 + z] = (int16_t) (4096 - (int) (k >> 13));
        index[(size_t) a * kRow3 + z] = (uint16_t) (b0 + 20 * b1 + 400 * b2);
    }
}

// --------------------------------------------------------------------------------------------------------------
// search, stage 1: similar k-mers per query position
// --------------------------------------------------------------------------------------------------------------
// number of leading elements of a descending row that are >= cut
__device__ inline int countGE(const int16_t *row, int n, int cut) {
    int lo = 0, hi = n;                       // first index with row[i] < cut
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (row[mid] >= cut) lo = mid + 1; else hi = mid; }
    return lo;
}

// Similar k-mers of one query position (KmerGenerator::generateKmerList for k = 6 = 3 + 3): with S1 / S2 the sorted
// score rows of the first / last 3-mer, the list is  { (x, y) : S1[x] >= thr - S2[0],  S2[y] >= thr - S1[x] }  in
// x-major order.  The inner bound depends on x only through the VALUE S1[x], and a row holds a few dozen distinct values
// among its passing entries, so the staircase is described by one RUN per distinct value v (from S1[0] downwards):
//   xs(v) = #{S1 > v},  len(v) = #{S1 == v},  c(v) = #{S2 >= thr - v}.
// One thread per value builds the runs with three binary searches; the r-th k-mer of the position is then found by a
// search over at most a few hundred run offsets in LDS (3 KB per workgroup instead of staged row prefixes).
constexpr int kMaxRuns = 1024;                // distinct score values between cutoff and row maximum (< 3 * 256)

struct KmerPosInfo { int a, b, thr, vmax, nV; bool skip; };

__device__ inline KmerPosInfo kmerPosInfo(const KmerQ &q, const uint8_t *seqs, const int16_t *thrs, uint32_t p, KmerPattern pat, const int16_t *s3) {
    KmerPosInfo r;
    const uint32_t i = p - q.posBase;
    const uint8_t *s = seqs + q.seqOff + i;
    uint32_t c[6];
    bool x = false;
#pragma unroll
    for (int z = 0; z < 6; z++) { c[z] = s[pat.pos[z]]; x |= c[z] >= 20; }
    r.skip = x;
    r.a = (int) (c[0] + 20 * c[1] + 400 * c[2]);
    r.b = (int) (c[3] + 20 * c[4] + 400 * c[5]);
    r.thr = thrs[p];
    r.vmax = 0; r.nV = 0;
    if (!x) {
        const int cutoff1 = (int) (int16_t) (r.thr - s3[(size_t) r.b * kRow3]);      // short cutoff1 = threshold - possibleRest[0]
        r.vmax = s3[(size_t) r.a * kRow3];
        r.nV = r.vmax >= cutoff1 ? min(r.vmax - cutoff1 + 1, kMaxRuns) : 0;
    }
    return r;
}

// run of value v = vmax - t: first x, number of x, inner list length c
__device__ inline void kmerRun(const int16_t *S1, const int16_t *S2, int thr, int v, int &xs, int &len, int &c) {
    xs = countGE(S1, kRow3, v + 1);
    len = countGE(S1, kRow3, v) - xs;
    c = len ? countGE(S2, kRow3, (int) (int16_t) (thr - v)) : 0;                       // short cutoff2 = threshold - score_i - possibleRest
}

// pass 1: K_p = number of similar k-mers of position p (capped like calculateArrayProduct)
__global__ __launch_bounds__(kKmerBlock) void k_kmer_count(const KmerQ *qs, const uint16_t *posQuery, const uint8_t *seqs, const int16_t *thrs,
                                                           uint32_t nPos, KmerPattern pat, const int16_t *s3, uint32_t *K) {
    const uint32_t p = blockIdx.x;
    if (p >= nPos) return;
    const KmerQ q = qs[posQuery[p]];
    __shared__ KmerPosInfo info;
    __shared__ unsigned long long total;
    if (threadIdx.x == 0) { info = kmerPosInfo(q, seqs, thrs, p, pat, s3); total = 0; }
    __syncthreads();
    if (info.skip || info.nV == 0) { if (threadIdx.x == 0) K[p] = 0; return; }
    const int16_t *S1 = s3 + (size_t) info.a * kRow3, *S2 = s3 + (size_t) info.b * kRow3;
    unsigned long long mine = 0;
    for (int t = threadIdx.x; t < info.nV; t += blockDim.x) {
        int xs, len, c;
        kmerRun(S1, S2, info.thr, info.vmax - t, xs, len, c);
        mine += (unsigned long long) len * (unsigned) c;
    }
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) K[p] = (uint32_t) (total < (unsigned long long) (kMaxKmerResult - 1) ? total : (kMaxKmerResult - 1));
}

// pass 2: enumerate the similar k-mers of position p in the reference's order, probe bitmap + offset table and write one
// (entry start, size, position) triple per k-mer into the list arrays at Kbase[p].  HBM traffic = one sector per
// non-empty list + the list arrays; everything else is served by LDS / L1 / L2.
__global__ __launch_bounds__(kKmerBlock) void k_kmer_lists(const KmerQ *qs, const uint16_t *posQuery, const uint8_t *seqs, const int16_t *thrs,
                                                           uint32_t nPos, KmerPattern pat, const int16_t *s3, const uint16_t *i3,
                                                           const uint32_t *Kcount, const uint64_t *Kbase, const uint32_t *offsets, const uint32_t *bitmap,
                                                           uint32_t *listStart, uint32_t *listSize, uint32_t *listPos) {
    const uint32_t p = blockIdx.x;
    if (p >= nPos) return;
    const uint32_t Kp = Kcount[p];
    if (Kp == 0) return;
    const KmerQ q = qs[posQuery[p]];
    __shared__ KmerPosInfo info;
    __shared__ uint32_t runOx[kMaxRuns + 1];  // exclusive prefix of len * c over the runs
    __shared__ uint16_t runXs[kMaxRuns], runC[kMaxRuns];
    __shared__ uint32_t wsum[kKmerBlock / 64];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) { info = kmerPosInfo(q, seqs, thrs, p, pat, s3); carry = 0; }
    __syncthreads();
    const int16_t *S1 = s3 + (size_t) info.a * kRow3, *S2 = s3 + (size_t) info.b * kRow3;
    const uint16_t *I1 = i3 + (size_t) info.a * kRow3, *I2 = i3 + (size_t) info.b * kRow3;
    const int nV = info.nV;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t0 = 0; t0 < nV; t0 += kKmerBlock) {
        const int t = t0 + threadIdx.x;
        int xs = 0, len = 0, c = 0;
        if (t < nV) kmerRun(S1, S2, info.thr, info.vmax - t, xs, len, c);
        const uint32_t prod = (uint32_t) len * (uint32_t) c;
        // block exclusive scan of prod in thread order
        uint32_t incl = prod;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (t < nV) { runOx[t] = before + incl - prod; runXs[t] = (uint16_t) xs; runC[t] = (uint16_t) c; }
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t sum = carry; for (int w = 0; w < kKmerBlock / 64; w++) sum += wsum[w]; carry = sum; }
        __syncthreads();
    }
    if (threadIdx.x == 0) runOx[nV] = carry;
    __syncthreads();
    const uint64_t base = Kbase[p];
    for (uint32_t r0 = threadIdx.x; r0 < Kp; r0 += 4 * kKmerBlock) {
        uint32_t kmer[4], st[4], en[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t r = r0 + u * kKmerBlock;
            kmer[u] = 0;
            if (r < Kp) {
                int lo = 0, hi = nV;          // last run with runOx <= r (empty runs share their offset with the successor)
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (runOx[mid] <= r) lo = mid; else hi = mid; }
                const uint32_t d = r - runOx[lo], c = runC[lo];
                const uint32_t dx = d / c;
                kmer[u] = kmerDeviceIndex(I1[runXs[lo] + dx], I2[d - dx * c]);
            }
        }
        uint32_t bm[4];
#pragma unroll
        for (int u = 0; u < 4; u++) bm[u] = bitmap[kmer[u] >> 5];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            st[u] = 0; en[u] = 0;
            if ((bm[u] >> (kmer[u] & 31)) & 1u) { st[u] = offsets[kmer[u]]; en[u] = offsets[kmer[u] + 1]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t r = r0 + u * kKmerBlock;
            if (r < Kp) { listStart[base + r] = st[u]; listSize[base + r] = en[u] - st[u]; listPos[base + r] = p; }
        }
    }
}

// per-query bases: list slots and hit-stream start (Kbase / listP are exclusive scans with the total appended)
__global__ void k_kmer_qbases(KmerQ *qs, int nq, const uint64_t *Kbase, const uint64_t *listP) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint64_t lb = Kbase[qs[q].posBase];
    qs[q].listBase = lb;
    qs[q].hitBase = listP[lb];
}

// databaseHits refills: chunk c+1 starts at the first list l with (hits of chunk c so far) + size(l) >= maxDbMatches
// (QueryMatcher.cpp:302-333).  One thread per query; every step is a binary search over the list prefix array.
__global__ void k_kmer_chunks(const KmerQ *qs, int nq, const uint64_t *Kbase, const uint64_t *listP, const uint32_t *listPos, uint64_t maxDbMatches, KmerChunks *out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint64_t lb = Kbase[qs[q].posBase], le = Kbase[qs[q].posBase + qs[q].nPos];
    const uint64_t base = listP[lb];
    KmerChunks &c = out[q];
    c.total = listP[le] - base;
    c.aborted = 0;
    c.abortKmers = 0;
    c.start[0] = 0;
    uint32_t nc = 1;
    uint64_t cur = lb, G = 0;
    while (true) {
        // first l >= cur with listP[l+1] - base >= G + maxDbMatches
        const uint64_t want = base + G + maxDbMatches;
        uint64_t lo = cur, hi = le;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (listP[mid + 1] >= want) hi = mid; else lo = mid + 1; }
        if (lo >= le) break;
        const uint64_t G2 = listP[lo] - base;
        if (listP[lo + 1] - listP[lo] >= maxDbMatches) { c.aborted = 1; c.abortKmers = Kbase[listPos[lo] + 1] - lb; }
        if (nc >= (uint32_t) kMaxChunks) { c.aborted = 2; break; }
        c.start[nc++] = G2;
        if (c.aborted) break;
        G = G2; cur = lo;
    }
    c.nChunks = nc;
    c.start[nc] = c.total;
}

// --------------------------------------------------------------------------------------------------------------
// search, stage 2: the hit stream.  Output-balanced gather: one thread per hit, list found by binary search.
// --------------------------------------------------------------------------------------------------------------
constexpr int kEmitTile = 2048;               // outputs per workgroup
constexpr int kEmitStage = 6144;              // list prefixes staged in LDS (24 KB -> 6 workgroups per CU)
__global__ __launch_bounds__(256) void k_kmer_emit(const KmerQ *qs, const KmerChunks *chunks, const uint16_t *posQuery, uint64_t nLists, const uint64_t *listP,
                                                   const uint32_t *listStart, const uint32_t *listPos, const uint64_t *entries,
                                                   uint64_t nHits, int tbits, uint32_t *keys, uint64_t *vals) {
    __shared__ uint64_t range[2];
    __shared__ uint32_t rel[kEmitStage + 1];  // list prefix relative to the block's first list
    const uint64_t o0 = (uint64_t) blockIdx.x * kEmitTile;
    if (o0 >= nHits) return;
    const uint64_t o1 = min(nHits, o0 + kEmitTile);
    if (threadIdx.x < 2) {
        const uint64_t o = threadIdx.x == 0 ? o0 : o1 - 1;
        uint64_t lo = 0, hi = nLists;         // last l with listP[l] <= o (that list is non-empty and contains o)
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (listP[mid] <= o) lo = mid; else hi = mid; }
        range[threadIdx.x] = lo;
    }
    __syncthreads();
    const uint64_t l0 = range[0], l1 = range[1];
    const uint64_t p0 = listP[l0];
    const int nl = (int) min<uint64_t>(l1 - l0 + 1, (uint64_t) kEmitStage + 1);
    const bool staged = l1 - l0 + 1 <= (uint64_t) kEmitStage;
    if (staged) {
        for (int i = threadIdx.x; i < nl; i += 256) rel[i] = (uint32_t) (listP[l0 + i] - p0);
        __syncthreads();
    }
    // 8 outputs per thread, handled phase by phase so that the dependent loads of all 8 are in flight together
    constexpr int U = kEmitTile / 256;
    uint64_t l[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint64_t o = o0 + threadIdx.x + 256 * u;
        l[u] = l0;
        if (o < o1) {
            if (staged) {
                const uint32_t ro = (uint32_t) (o - p0);
                int lo = 0, hi = nl;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rel[mid] <= ro) lo = mid; else hi = mid; }
                l[u] = l0 + lo;
            } else {
                uint64_t lo = l0, hi = l1 + 1;
                while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (listP[mid] <= o) lo = mid; else hi = mid; }
                l[u] = lo;
            }
        }
    }
    uint32_t p[U], st[U];
    uint64_t lp[U];
#pragma unroll
    for (int u = 0; u < U; u++) { p[u] = listPos[l[u]]; st[u] = listStart[l[u]]; lp[u] = listP[l[u]]; }
    uint64_t e[U];
    uint32_t qi[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint64_t o = o0 + threadIdx.x + 256 * u;
        e[u] = o < o1 ? entries[(uint64_t) st[u] + (o - lp[u])] : 0;
        qi[u] = posQuery[p[u]];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint64_t o = o0 + threadIdx.x + 256 * u;
        if (o >= o1) continue;
        const KmerQ &q = qs[qi[u]];
        const uint32_t seqId = (uint32_t) (e[u] >> 16), posj = (uint32_t) e[u] & 0xffffu;
        const uint32_t i = p[u] - q.posBase;
        const uint64_t g = o - q.hitBase;
        const KmerChunks &ck = chunks[qi[u]];
        uint32_t c = 0;
        while (c + 1 < ck.nChunks && ck.start[c + 1] <= g) c++;
        keys[o] = (qi[u] << tbits) | seqId;
        vals[o] = hitPack(g, (i - posj) & 0xffffu, c);
    }
}

// --------------------------------------------------------------------------------------------------------------
// search, stage 3 (after the stable sort by (query, target)): double-diagonal detection
// --------------------------------------------------------------------------------------------------------------
// findDuplicates pass 1: a hit is a candidate iff its 8-bit diagonal equals that of the previous hit of the same
// target in the same chunk (the byte array starts at 0, so a first hit on diagonal 0 also counts).  Predicate of the
// ordered candidate compaction below.
struct KmerDupPred {
    const uint32_t *keys;
    const uint64_t *vals;
    __device__ bool operator()(uint32_t i) const {
        const uint32_t k = keys[i];
        const uint64_t v = vals[i];
        uint32_t prev = 0;
        if (i > 0 && keys[i - 1] == k) { const uint64_t pv = vals[i - 1]; if (hitChunk(pv) == hitChunk(v)) prev = hitD8(pv); }
        return hitD8(v) == prev;
    }
};
// Ordered compaction of the candidates in two coalesced passes over the sorted hit stream (24 B per hit in total):
// pass 1 counts the candidates of every 2048-hit tile, a scan over the tile counts gives the tile bases, pass 2
// re-evaluates the predicate, ranks the candidates inside the tile with wave ballots and writes them in order.
constexpr int kDupTile = 2048;
__global__ __launch_bounds__(256) void k_kmer_dupcount(KmerDupPred pred, uint64_t n, uint32_t *tileCount) {
    __shared__ uint32_t wsum[4];
    const uint64_t base = (uint64_t) blockIdx.x * kDupTile;
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < kDupTile / 256; u++) {
        const uint64_t i = base + u * 256 + threadIdx.x;
        const bool f = i < n && pred((uint32_t) i);
        c += (uint32_t) __popcll(__ballot(f));
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tileCount[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void k_kmer_dupscatter(KmerDupPred pred, uint64_t n, const uint32_t *tileBase, int tbits,
                                                         uint32_t *ckeys, uint64_t *cvals, uint32_t *ecCount /*[nq][kMaxChunks]*/) {
    constexpr int U = kDupTile / 256;
    __shared__ uint32_t wcnt[U][4];
    __shared__ uint32_t h[kMaxChunks];        // per-chunk candidate counts of the block's first query
    __shared__ uint32_t q0;
    const uint64_t base = (uint64_t) blockIdx.x * kDupTile;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    h[threadIdx.x] = 0;
    if (threadIdx.x == 0) q0 = pred.keys[base] >> tbits;
    uint32_t mine = 0, rank[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint64_t i = base + u * 256 + threadIdx.x;
        const bool f = i < n && pred((uint32_t) i);
        const unsigned long long m = __ballot(f);
        rank[u] = (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
        if (f) mine |= 1u << u;
        if (lane == 0) wcnt[u][wave] = (uint32_t) __popcll(m);
    }
    __syncthreads();
    uint32_t run = tileBase[blockIdx.x];
#pragma unroll
    for (int u = 0; u < U; u++) {
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) before += w < wave ? wcnt[u][w] : 0;
        if ((mine >> u) & 1u) {
            const uint64_t i = base + u * 256 + threadIdx.x;
            const uint32_t j = run + before + rank[u];
            const uint32_t k = pred.keys[i];
            const uint64_t v = pred.vals[i];
            ckeys[j] = k; cvals[j] = v;
            if ((k >> tbits) == q0) atomicAdd(&h[hitChunk(v)], 1u);
            else atomicAdd(&ecCount[(size_t) (k >> tbits) * kMaxChunks + hitChunk(v)], 1u);
        }
        run += wcnt[u][0] + wcnt[u][1] + wcnt[u][2] + wcnt[u][3];
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&ecCount[(size_t) q0 * kMaxChunks + threadIdx.x], h[threadIdx.x]);
}

// findDuplicates pass 2 (collapse runs of equal 8-bit diagonals among the candidates of one target and chunk) fused
// with UngappedAlignment scoring of the survivors: score32 = uncapped best ungapped run on the 16-bit diagonal.
// kept[j] = 0 dropped, 1 kept.  One thread per candidate; the candidates are sorted by query, so a workgroup stages the
// int8 profile of its first query in LDS (other queries of a straddling workgroup read theirs from global memory) and
// every lane walks its diagonal with aligned 8-byte target loads: 8 cells per memory round trip.
template <bool LDS>
__device__ inline int kmerDiagScore(const int8_t *prof, const uint8_t *db, int len) {
    int mx = 0, s = 0, pos = 0;
    // head: up to the first 8-byte boundary of the target
    const int head = min(len, (int) ((8 - ((uintptr_t) db & 7)) & 7));
    for (; pos < head; pos++) { s += prof[pos * 21 + db[pos]]; s = s < 0 ? 0 : s; mx = s > mx ? s : mx; }
    for (; pos + 8 <= len; pos += 8) {
        const uint64_t w = *reinterpret_cast<const uint64_t *>(db + pos);
        const int8_t *p = prof + pos * 21;
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = p[k * 21 + (int) ((w >> (8 * k)) & 0xff)];
#pragma unroll
        for (int k = 0; k < 8; k++) { s += v[k]; s = s < 0 ? 0 : s; mx = s > mx ? s : mx; }
    }
    for (; pos < len; pos++) { s += prof[pos * 21 + db[pos]]; s = s < 0 ? 0 : s; mx = s > mx ? s : mx; }
    return mx;
}

__global__ __launch_bounds__(256) void k_kmer_score(const uint32_t *ckeys, const uint64_t *cvals, const uint32_t *nCandPtr, int tbits, const KmerQ *qs,
                                                    const int8_t *profiles, const uint8_t *masked, const uint64_t *offsets, const int32_t *lengths,
                                                    int ldsBytes, uint8_t *kept, int32_t *score) {
    extern __shared__ int8_t sprof[];
    __shared__ uint32_t q0s;
    const uint64_t nCand = *nCandPtr;
    const uint64_t j0 = (uint64_t) blockIdx.x * 256, j = j0 + threadIdx.x;
    if (j0 >= nCand) return;
    if (threadIdx.x == 0) q0s = ckeys[j0] >> tbits;
    __syncthreads();
    const uint32_t q0 = q0s;
    const int L0 = (int) qs[q0].L;
    const bool staged = L0 * 21 <= ldsBytes;
    if (staged) {
        const int8_t *src = profiles + qs[q0].profOff;       // profOff is a multiple of 1 byte only: copy bytewise in 4-byte lanes when aligned
        const int nb = L0 * 21;
        for (int i = threadIdx.x; i < nb; i += 256) sprof[i] = src[i];
    }
    __syncthreads();
    if (j >= nCand) return;
    const uint32_t k = ckeys[j];
    const uint64_t v = cvals[j];
    bool keep = true;
    if (j > 0 && ckeys[j - 1] == k) { const uint64_t pv = cvals[j - 1]; if (hitChunk(pv) == hitChunk(v) && hitD8(pv) == hitD8(v)) keep = false; }
    kept[j] = keep ? 1 : 0;
    if (!keep) { score[j] = 0; return; }
    const uint32_t t = k & ((1u << tbits) - 1u);
    const uint32_t qi = k >> tbits;
    const KmerQ &q = qs[qi];
    const uint8_t *db = masked + offsets[t];
    const int dbLen = lengths[t], qLen = (int) q.L;
    const uint32_t d = hitDiag(v);
    const int diagonal = (int) (int16_t) (uint16_t) d;
    const int minDist = (int) min((0u - d) & 0xffffu, d);
    int len = 0, poff = 0;
    if (diagonal >= 0 && minDist < qLen) { len = min(dbLen, qLen - minDist); poff = minDist * 21; }
    else if (diagonal < 0 && minDist < dbLen) { len = min(dbLen - minDist, qLen); db += minDist; }
    int mx;
    if (staged && qi == q0) mx = kmerDiagScore<true>(sprof + poff, db, len);
    else mx = kmerDiagScore<false>(profiles + q.profOff + poff, db, len);
    score[j] = mx;
}

// --------------------------------------------------------------------------------------------------------------
// search, stage 4: per-target replay of QueryMatcher::match's overflow rounds + final keepMaxScoreElementOnly.
// One thread per (query, target) segment of the candidate array.  List elements are (candidate index << 8 | count).
// --------------------------------------------------------------------------------------------------------------
struct KmerBest {          // written at the segment head; nElems == 0xFFFFFFFF marks "not a head"
    uint32_t nElems;       // elements this target contributes to resultSize (first max + later zero-score ones)
    uint32_t cand;         // candidate index of the kept element
    uint32_t count;        // its 8-bit score
    uint32_t pad;
};

__global__ __launch_bounds__(128) void k_kmer_walk(const uint32_t *ckeys, const uint64_t *cvals, const uint8_t *kept, const int32_t *score, const uint32_t *nCandPtr,
                            int tbits, const KmerChunks *chunks, uint64_t *scrA, uint64_t *scrB, KmerBest *best,
                            uint32_t *roundCount /*[nq][kMaxChunks]*/, unsigned long long *resultSize /*[nq]*/) {
    __shared__ uint32_t rc[kMaxChunks];       // per-round element counts / result size of the block's first query
    __shared__ unsigned long long rs;
    __shared__ uint32_t q0;
    const uint64_t s = (uint64_t) blockIdx.x * 128 + threadIdx.x;
    const uint64_t nCand = *nCandPtr;
    rc[threadIdx.x] = 0; rc[threadIdx.x + 128] = 0;
    if (threadIdx.x == 0) { rs = 0; q0 = (uint64_t) blockIdx.x * 128 < nCand ? ckeys[(uint64_t) blockIdx.x * 128] >> tbits : 0; }
    __syncthreads();
    bool head = s < nCand;
    uint32_t key = 0;
    if (head) { key = ckeys[s]; if (s > 0 && ckeys[s - 1] == key) { best[s].nElems = 0xFFFFFFFFu; head = false; } }
    if (head) {
    const uint32_t qi = key >> tbits;
    const bool mine = qi == q0;
#define ROUND_ADD(j, v) do { if (v) { if (mine) atomicAdd(&rc[j], (v)); else atomicAdd(&roundCount[(size_t) qi * kMaxChunks + (j)], (v)); } } while (0)
    const KmerChunks &ck = chunks[qi];
    const uint32_t C = ck.nChunks - 1;                       // number of overflow rounds
    const bool lastEmpty = ck.start[ck.nChunks] == ck.start[ck.nChunks - 1];
    uint64_t *A = scrA + s, *B = scrB + s;
    uint32_t nE = 0;
    uint64_t pos = s;
#define EL_CNT(e) ((uint32_t) (e) & 0xffu)
#define EL_D8(e) hitD8(cvals[(e) >> 8])
    for (uint32_t j = 1; j <= C; j++) {
        uint32_t nS = nE;
        while (pos < nCand && ckeys[pos] == key && hitChunk(cvals[pos]) == j - 1) { if (kept[pos]) A[nS++] = pos << 8; pos++; }
        if (j == 1) { nE = nS; ROUND_ADD(j, nE); continue; }
        if (nS == 0) { nE = 0; continue; }
        // mergeDiagonalKeepScoredHitsDuplicates: reverse walk, scored elements always survive
        uint32_t arr = (EL_D8(A[nS - 1]) + 1) & 0xffu, nB = 0;
        for (uint32_t n = nS; n-- > 0;) {
            const uint64_t e = A[n];
            const uint32_t d8 = EL_D8(e);
            if (EL_CNT(e) != 0 || arr != d8) B[nB++] = e;
            arr = d8;
        }
        // UngappedAlignment::align (only unscored elements) + keepMaxScoreElementOnly
        uint32_t mx = 0;
        for (uint32_t n = 0; n < nB; n++) {
            uint64_t e = B[n];
            if (EL_CNT(e) == 0) { const int sc = score[e >> 8]; e |= (uint32_t) (sc > 255 ? 255 : sc); B[n] = e; }
            mx = max(mx, EL_CNT(e));
        }
        arr = mx; nE = 0;
        for (uint32_t n = 0; n < nB; n++) {
            const uint64_t e = B[n];
            if (arr == EL_CNT(e)) { A[nE++] = e; arr = 0; }
        }
        ROUND_ADD(j, nE);
    }
    // last chunk
    uint32_t nS = nE;
    while (pos < nCand && ckeys[pos] == key) { if (kept[pos] && hitChunk(cvals[pos]) == C) A[nS++] = pos << 8; pos++; }
    uint64_t *Lst = A;
    uint32_t nL = nS;
    if (C >= 1) {
        if (lastEmpty) {
            nL = 0;                                          // numMatches == 0 after the last refill: match() returns 0 hits
        } else if (nS > 0) {
            // mergeDiagonalDuplicates: forward walk, counts ignored
            uint32_t arr = (EL_D8(A[0]) + 1) & 0xffu, nB = 0;
            for (uint32_t n = 0; n < nS; n++) {
                const uint64_t e = A[n];
                const uint32_t d8 = EL_D8(e);
                if (arr != d8) B[nB++] = e;
                arr = d8;
            }
            Lst = B; nL = nB;
        }
    }
    // matchQuery: align + keepMaxScoreElementOnly
    uint32_t mx = 0;
    for (uint32_t n = 0; n < nL; n++) {
        uint64_t e = Lst[n];
        if (EL_CNT(e) == 0) { const int sc = score[e >> 8]; e |= (uint32_t) (sc > 255 ? 255 : sc); Lst[n] = e; }
        mx = max(mx, EL_CNT(e));
    }
    uint32_t arr = mx, nF = 0;
    uint64_t first = 0;
    for (uint32_t n = 0; n < nL; n++) {
        const uint64_t e = Lst[n];
        if (arr == EL_CNT(e)) { if (nF == 0) first = e; nF++; arr = 0; }
    }
#undef EL_CNT
#undef EL_D8
    KmerBest b;
    b.nElems = nF; b.cand = (uint32_t) (first >> 8); b.count = (uint32_t) first & 0xffu; b.pad = 0;
    best[s] = b;
    if (nF) { if (mine) atomicAdd(&rs, (unsigned long long) nF); else atomicAdd(&resultSize[qi], (unsigned long long) nF); }
#undef ROUND_ADD
    }
    __syncthreads();
    for (int j = threadIdx.x; j < kMaxChunks; j += 128) if (rc[j]) atomicAdd(&roundCount[(size_t) q0 * kMaxChunks + j], rc[j]);
    if (threadIdx.x == 0 && rs) atomicAdd(&resultSize[q0], rs);
}

// --------------------------------------------------------------------------------------------------------------
// search, stage 5: score histogram, cut (computeScoreThreshold) and hand-over of everything at or above the cut
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kmer_hist(const uint32_t *ckeys, const KmerBest *best, const uint32_t *nCandPtr, int tbits, uint32_t *hist /*[nq][256]*/) {
    __shared__ uint32_t h[256];
    __shared__ uint32_t q0;
    const uint64_t s = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    const uint64_t nCand = *nCandPtr;
    h[threadIdx.x] = 0;
    if (threadIdx.x == 0) q0 = (uint64_t) blockIdx.x * 256 < nCand ? ckeys[(uint64_t) blockIdx.x * 256] >> tbits : 0;
    __syncthreads();
    if (s < nCand) {
        const KmerBest b = best[s];
        if (b.nElems != 0xFFFFFFFFu && b.nElems != 0) {
            const uint32_t qi = ckeys[s] >> tbits;
            if (qi == q0) {
                atomicAdd(&h[b.count], 1u);
                if (b.nElems > 1) atomicAdd(&h[0], b.nElems - 1);          // the extra elements all carry score 0
            } else {
                atomicAdd(&hist[(size_t) qi * 256 + b.count], 1u);
                if (b.nElems > 1) atomicAdd(&hist[(size_t) qi * 256], b.nElems - 1);
            }
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[(size_t) q0 * 256 + threadIdx.x], h[threadIdx.x]);
}
__global__ void k_kmer_cut(const uint32_t *hist, int nq, uint32_t maxHits, uint32_t minDiag, uint32_t *thr) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t *h = hist + (size_t) q * 256;
    uint64_t found = 0;
    uint32_t t;
    for (t = 255; t > 0; t--) { found += h[t]; if (found >= maxHits) break; }
    thr[q] = t < minDiag ? minDiag : t;
}
struct KmerOut { uint32_t id; uint32_t count; uint32_t diag; int32_t score; uint64_t g; };
__global__ __launch_bounds__(256) void k_kmer_out(const uint32_t *ckeys, const uint64_t *cvals, const int32_t *score, const KmerBest *best, const uint32_t *nCandPtr, int tbits,
                                                  const uint32_t *thr, uint32_t cap, uint32_t *outCount /*[nq]*/, KmerOut *out /*[nq][cap]*/) {
    const uint64_t s = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    bool take = false;
    uint32_t qi = 0;
    KmerBest b{};
    if (s < *nCandPtr) {
        b = best[s];
        if (b.nElems != 0xFFFFFFFFu && b.nElems != 0) {
            qi = ckeys[s] >> tbits;
            take = b.count >= thr[qi] && b.count != 0;
        }
    }
    // one atomic per (wave, query) instead of one per element: same-address atomics serialise in L2
    uint32_t slot = 0;
    const int lane = (int) (threadIdx.x & 63);
    unsigned long long pending = __ballot(take);
    while (pending) {
        const int leader = __ffsll((long long) pending) - 1;
        const uint32_t q = (uint32_t) __shfl((int) qi, leader);
        const unsigned long long grp = __ballot(take && qi == q);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&outCount[q], (uint32_t) __popcll(grp));
        base = (uint32_t) __shfl((int) base, leader);
        if (take && qi == q) slot = base + (uint32_t) __popcll(grp & ((1ull << lane) - 1ull));
        pending &= ~grp;
    }
    if (!take || slot >= cap) return;
    const uint64_t v = cvals[b.cand];
    KmerOut o;
    o.id = ckeys[s] & ((1u << tbits) - 1u); o.count = b.count; o.diag = hitDiag(v); o.score = score[b.cand]; o.g = hitG(v);
    out[(size_t) qi * cap + slot] = o;
}

} // namespace fs
