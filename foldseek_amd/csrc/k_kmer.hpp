// k_kmer.hpp -- gfx950 kernels of the k-mer prefilter:
//   index build   : masking, k-mer extraction, (k-mer, seqId) de-duplication            IndexBuilder.cpp:56-271, IndexTable.h
//   3-mer table   : extended substitution matrix rows, stable-sorted                      ExtendedSubstitutionMatrix.cpp:20-69
//   search        : similar k-mers per query position (KmerGenerator.cpp:108-217), index gather into the hit stream
//                   (QueryMatcher.cpp:243-376), double-diagonal detection (CacheFriendlyOperations.cpp:188-283),
//                   diagonal scoring (UngappedAlignment.cpp:45-57,430-443), per-target replay of the overflow rounds
//                   (mergeElements / keepMaxScoreElementOnly), score histogram + cut (QueryMatcher.h:211-221).
//
// All of it is integer / byte work bound by HBM latency and bandwidth (random index-table probes, 4-byte entry gathers,
// 6-byte hit records through one order-preserving scatter); no MFMA.  The reference's arrival-order rules are kept exact:
// the hit stream is partitioned STABLY into (query, target range) runs, the double-diagonal rule walks every run in arrival
// order over the reference's own byte-per-target array (in LDS), and the 2-3 % of the hits it flags get their stream
// position back, after which every later order-dependent rule is a per-target, neighbour-only rule.
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
// the same as 4-byte entries seqId << posBits | position, for databases whose id and position bits fit 32 together (IndexTable.h:25-41 stores
// {u32 seqId, u16 position}: six bytes; the search gathers whole sectors, so the entry width is what the gather of a short list costs)
__global__ void k_kmer_compact_entries32(const uint64_t *vals, const uint32_t *flags, const uint32_t *scan, uint64_t n, int posBits, uint32_t *entries) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) { const uint64_t v = vals[i]; entries[scan[i]] = ((uint32_t) (v >> 16) << posBits) | ((uint32_t) v & 0xffffu); }
}
__global__ void k_kmer_widen_entries(const uint32_t *e32, uint64_t n, int posBits, uint64_t *e64) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = e32[i];
    e64[i] = ((uint64_t) (e >> posBits) << 16) | (e & ((1u << posBits) - 1u));
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
// Work items are SLICES of kListSlice similar k-mers of one position (round 5): a position has between none and 8 M similar k-mers (mean 7 000, 1 % of
// the positions beyond 40 000 at -s 9.5), and with one workgroup per position the launch lasted as long as its heaviest positions.  Slice s of position p
// sits at grid slot p + floor(Kbase[p] / kListSlice) + s -- slots of different positions never collide (floor(a + b) - floor(a) >= floor(b)), a slot without a
// slice exits -- so the grid is nPos + nLists / kListSlice + 1 workgroups and needs no item table.  Every slice rebuilds its position's run table.
constexpr uint32_t kListSlice = 8192;
__global__ __launch_bounds__(kKmerBlock) void k_kmer_lists(const KmerQ *qs, const uint16_t *posQuery, const uint8_t *seqs, const int16_t *thrs,
                                                           uint32_t nPos, KmerPattern pat, const int16_t *s3, const uint16_t *i3,
                                                           const uint32_t *Kcount, const uint64_t *Kbase, const uint32_t *offsets, const uint32_t *bitmap,
                                                           uint32_t *listStart, uint32_t *listSize, uint32_t *listPos) {
    __shared__ uint32_t pSh;
    if (threadIdx.x == 0) {
        const uint64_t slot = blockIdx.x;
        uint32_t lo = 0, hi = nPos;                // last position whose first slot is <= this slot
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t) mid + Kbase[mid] / kListSlice <= slot) lo = mid; else hi = mid; }
        pSh = lo;
    }
    __syncthreads();
    const uint32_t p = pSh;
    const uint32_t Kp = Kcount[p];
    const uint32_t slice = (uint32_t) ((uint64_t) blockIdx.x - ((uint64_t) p + Kbase[p] / kListSlice));
    if ((uint64_t) slice * kListSlice >= Kp) return;            // (covers Kp == 0)
    const uint32_t rBeg = slice * kListSlice, rEnd = min(Kp, rBeg + kListSlice);
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
    const uint32_t qpos = ((uint32_t) posQuery[p] << 16) | (p - q.posBase);     // what k_kmer_emit needs of a list: its query and the k-mer's position in it
    for (uint32_t r0 = rBeg + threadIdx.x; r0 < rEnd; r0 += 4 * kKmerBlock) {
        uint32_t kmer[4], st[4], en[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t r = r0 + u * kKmerBlock;
            kmer[u] = 0;
            if (r < rEnd) {
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
            if (r < rEnd) { listStart[base + r] = st[u]; listSize[base + r] = en[u] - st[u]; listPos[base + r] = qpos; }
        }
    }
}

// ---- the same two passes with one WAVE per query position (four positions per workgroup) ----------------------------------------
// At low sensitivity (-s 4.5 and below: a few dozen similar k-mers per position, the cluster workflow's prefilter steps) a 256-thread
// workgroup per position leaves most lanes idle and its run time is a chain of ~40 dependent probes into the two score rows (three
// binary searches per run).  Here a wave first copies the PASSING PREFIX of both rows (scores and 3-mer indices) into LDS with one or two
// coalesced loads -- a row is sorted by descending score, only entries >= (cutoff of the row) can take part -- and searches there; a
// row whose passing prefix is longer than kWaveRowCap entries is searched in global memory like before.  Runs are handled 256 at a time
// (the r-th k-mer of a position is found by a search over the run offsets of the current batch of runs).  Same K, same lists, same order.
constexpr int kWaveRowCap = 512;
constexpr int kWaveRuns = 256;

// leading entries of the descending row S (kRow3 entries) that are >= cut -> dst[0..n); false when there are more than CAP
template <int CAP = kWaveRowCap>
__device__ inline bool kmerStagePrefix(const int16_t *S, const uint16_t *I, int cut, int16_t *dst, uint16_t *idst, int lane, int &n) {
    n = 0;
    for (int base = 0; base < CAP; base += 64) {
        const int16_t v = S[base + lane];
        dst[base + lane] = v;
        if (I) idst[base + lane] = I[base + lane];
        const int cnt = __popcll(__ballot((int) v >= cut));
        n += cnt;
        if (cnt < 64) return true;
    }
    return (int) S[CAP] < cut;
}

// CAP / RUNS = 512 / 256: 6.1 KB per wave, six workgroups per CU; 128 / 64 (round 6, batches with fewer than 128 similar k-mers per position: all-vs-all at -s 4.5 has
// 30): 1.5 KB per wave -- the waves per SIMD are then bounded by the hardware's eight, and the kernel is a chain of dependent probes that only more waves hide
template <int CAP, int RUNS>
struct KmerWaveLdsT {
    int16_t s1[CAP], s2[CAP];
    uint16_t i1[CAP], i2[CAP];
    uint32_t runOx[RUNS + 1];
    uint16_t runXs[RUNS], runC[RUNS];
};

__global__ __launch_bounds__(256) void k_kmer_count_w(const KmerQ *qs, const uint16_t *posQuery, const uint8_t *seqs, const int16_t *thrs,
                                                      uint32_t nPos, KmerPattern pat, const int16_t *s3, uint32_t *K) {
    __shared__ int16_t pre[4][2][kWaveRowCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t p = blockIdx.x * 4 + wave;
    if (p >= nPos) return;
    const KmerQ q = qs[posQuery[p]];
    const KmerPosInfo info = kmerPosInfo(q, seqs, thrs, p, pat, s3);
    if (info.skip || info.nV == 0) { if (lane == 0) K[p] = 0; return; }
    const int16_t *S1 = s3 + (size_t) info.a * kRow3, *S2 = s3 + (size_t) info.b * kRow3;
    int n1 = kRow3, n2 = kRow3;
    {
        int m1, m2;
        const bool f1 = kmerStagePrefix(S1, nullptr, info.vmax - info.nV + 1, pre[wave][0], nullptr, lane, m1);
        const bool f2 = kmerStagePrefix(S2, nullptr, (int) (int16_t) (info.thr - info.vmax), pre[wave][1], nullptr, lane, m2);
        if (f1) { S1 = pre[wave][0]; n1 = m1; }
        if (f2) { S2 = pre[wave][1]; n2 = m2; }
    }
    unsigned long long mine = 0;
    for (int t = lane; t < info.nV; t += 64) {
        const int v = info.vmax - t;
        const int xs = countGE(S1, n1, v + 1), len = countGE(S1, n1, v) - xs;
        const int c = len ? countGE(S2, n2, (int) (int16_t) (info.thr - v)) : 0;
        mine += (unsigned long long) len * (unsigned) c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if (lane == 0) K[p] = (uint32_t) (mine < (unsigned long long) (kMaxKmerResult - 1) ? mine : (kMaxKmerResult - 1));
}

template <int CAP, int RUNS>
__global__ __launch_bounds__(256) void k_kmer_lists_w(const KmerQ *qs, const uint16_t *posQuery, const uint8_t *seqs, const int16_t *thrs,
                                                      uint32_t nPos, KmerPattern pat, const int16_t *s3, const uint16_t *i3,
                                                      const uint32_t *Kcount, const uint64_t *Kbase, const uint32_t *offsets, const uint32_t *bitmap,
                                                      uint32_t *listStart, uint32_t *listSize, uint32_t *listPos) {
    using KmerWaveLds = KmerWaveLdsT<CAP, RUNS>;
    constexpr int kWaveRuns = RUNS;
    __shared__ KmerWaveLds lds[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t p = blockIdx.x * 4 + wave;
    if (p >= nPos) return;
    const uint32_t Kp = Kcount[p];
    if (Kp == 0) return;
    KmerWaveLds &W = lds[wave];
    const KmerQ q = qs[posQuery[p]];
    const KmerPosInfo info = kmerPosInfo(q, seqs, thrs, p, pat, s3);
    const int16_t *S1 = s3 + (size_t) info.a * kRow3, *S2 = s3 + (size_t) info.b * kRow3;
    const uint16_t *I1 = i3 + (size_t) info.a * kRow3, *I2 = i3 + (size_t) info.b * kRow3;
    int n1 = kRow3, n2 = kRow3;
    {
        int m1, m2;
        const bool f1 = kmerStagePrefix<CAP>(S1, I1, info.vmax - info.nV + 1, W.s1, W.i1, lane, m1);
        const bool f2 = kmerStagePrefix<CAP>(S2, I2, (int) (int16_t) (info.thr - info.vmax), W.s2, W.i2, lane, m2);
        if (f1) { S1 = W.s1; I1 = W.i1; n1 = m1; }
        if (f2) { S2 = W.s2; I2 = W.i2; n2 = m2; }
    }
    const uint64_t base = Kbase[p];
    const uint32_t qpos = ((uint32_t) posQuery[p] << 16) | (p - q.posBase);
    uint32_t carry = 0;                        // similar k-mers of the runs before the current batch of runs
    for (int t0 = 0; t0 < info.nV && carry < Kp; t0 += kWaveRuns) {
        const int nR = min(kWaveRuns, info.nV - t0);
        uint32_t batchTotal = 0;
        for (int tb = 0; tb < nR; tb += 64) {
            const int t = t0 + tb + lane;
            int xs = 0, len = 0, c = 0;
            if (tb + lane < nR) {
                const int v = info.vmax - t;
                xs = countGE(S1, n1, v + 1); len = countGE(S1, n1, v) - xs;
                c = len ? countGE(S2, n2, (int) (int16_t) (info.thr - v)) : 0;
            }
            const uint32_t prod = (uint32_t) len * (uint32_t) c;
            uint32_t incl = prod;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            if (tb + lane < nR) { W.runOx[tb + lane] = batchTotal + incl - prod; W.runXs[tb + lane] = (uint16_t) xs; W.runC[tb + lane] = (uint16_t) c; }
            batchTotal += __shfl(incl, 63);
        }
        if (lane == 0) W.runOx[nR] = batchTotal;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t end = min(Kp, carry + batchTotal);          // r in [carry, end) lies in this batch of runs
        for (uint32_t r0 = carry + lane; r0 < end; r0 += 4 * 64) {
            uint32_t kmer[4], st[4], en[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t r = r0 + u * 64;
                kmer[u] = 0;
                if (r < end) {
                    const uint32_t rr = r - carry;
                    int lo = 0, hi = nR;          // last run with runOx <= rr (empty runs share their offset with the successor)
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (W.runOx[mid] <= rr) lo = mid; else hi = mid; }
                    const uint32_t d = rr - W.runOx[lo], c = W.runC[lo];
                    const uint32_t dx = d / c;
                    kmer[u] = kmerDeviceIndex(I1[W.runXs[lo] + dx], I2[d - dx * c]);
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
                const uint32_t r = r0 + u * 64;
                if (r < end) { listStart[base + r] = st[u]; listSize[base + r] = en[u] - st[u]; listPos[base + r] = qpos; }
            }
        }
        carry += batchTotal;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
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
        if (listP[lo + 1] - listP[lo] >= maxDbMatches) { c.aborted = 1; c.abortKmers = Kbase[qs[q].posBase + (listPos[lo] & 0xffffu) + 1] - lb; }
        if (nc >= (uint32_t) kMaxChunks) { c.aborted = 2; break; }
        c.start[nc++] = G2;
        if (c.aborted) break;
        G = G2; cur = lo;
    }
    c.nChunks = nc;
    c.start[nc] = c.total;
}

// --------------------------------------------------------------------------------------------------------------
// search, stage 2: the hit stream -> double-diagonal candidates (round 6: a STABLE partition of narrow records).
//
// The reference decides "is this hit a double-diagonal hit" with a byte per target that remembers the 8-bit diagonal of
// the PREVIOUS hit of that target in arrival order (findDuplicates, CacheFriendlyOperations.cpp:188-283); to make that
// byte array fit its L2 it first scatters the hits into bins by target id (hashElements, :285-311), and it zeroes the
// array whenever databaseHits is flushed (QueryMatcher.cpp:311-346).  The same here, one level up the memory hierarchy:
//   k_kmer_emit            output-balanced gather of the index entries.  One 4-byte record (target id & 0xffff, 16-bit
//                          diagonal) + its 2-byte COARSE KEY per hit at the hit's stream position, and the number of
//                          hits per (tile, key) -- a tile = up to 16384 consecutive hits of ONE databaseHits chunk of
//                          one query.  A coarse key is a run of blocks of 1024 target ids (at most 64: the low 16 bits
//                          of an id are unique inside it), about 128 keys per database.
//   k_kmer_col_sums / _offsets   per (query, key) column over the query's tiles: where the run of (tile, key) starts
//   k_kmer_scatter_stable  records -> (query, key) segments, ORDER PRESERVING: inside a segment the hits stand in
//                          arrival order, so no record has to carry its stream position (rounds 3-5 moved 8-byte
//                          records twice through two unordered scatters and ranked every target's hits by their
//                          32-bit stream positions afterwards).  What identifies a hit later travels beside it as
//                          a 16-bit position inside its tile.
//   k_kmer_dup_stream      one wave per (query, chunk, key): the reference's byte array itself, in LDS (a byte per target
//                          of the key), walked in arrival order 64 hits at a time; hits of one step that share a
//                          target take their turn in lane order.  Flagged hits (2-3 %) get their stream position back
//                          (tile of their run + the 16-bit position) and leave as (query | target | stream position,
//                          diagonal | chunk).
//   k_kmer_cand_gather + a stable radix sort by (query | target)   the candidates in (query, target, arrival)
//                          order: the arrays the scoring and replay stages read.
// Bytes per hit: 6 written by emit, 6 read + 6 written by the scatter, 4 (+ 2 for the flagged ones) read by the
// duplicate stage.  Nothing here depends on the order in which atomics are served.
// --------------------------------------------------------------------------------------------------------------
constexpr int kEmitTile = 2048;               // hits per workgroup of k_kmer_emit
constexpr int kEmitStage = 6144;              // list prefixes of an emit tile staged in LDS
static_assert(kEmitStage < 8192, "k_kmer_emit searches the staged prefixes with 13 halving steps");
constexpr int kTileA = 16384;                 // hits per counting tile (a position inside it fits 16 bits)
constexpr int kSubTiles = kTileA / kEmitTile; // emit workgroups per tile
constexpr int kMaxCoarse = 512;               // coarse keys per database (LDS counters of emit and of the scatter)
constexpr int kCoarseBlocks = 64;             // blocks of 1024 ids per key at most
constexpr int kScStage = 4096;                // records per LDS pass of the stable scatter
constexpr int kScThreads = 512;
constexpr int kColGroups = 16;                // tile groups of the column kernels

struct KmerCoarse {
    const uint16_t *blkKey;       // [nBlk] key of every block of 1024 target ids
    const uint32_t *keyFirst;     // [nKeys + 1] first target id of a key
    uint32_t nBlk, nKeys, keyBits, maxIds;
};
struct KmerTiles {
    const uint32_t *tileStart;    // [nT + 1] first stream position of a tile (tiles follow the stream; tileStart[nT] = hits of the batch)
    const uint32_t *qTile0;       // [nq + 1] first tile of a query
    const uint32_t *vqTile0;      // [nVq + 1] first tile of a (query, chunk) pair
    const uint32_t *qVq0;         // [nq + 1] first (query, chunk) pair of a query
    const uint16_t *vqQ;          // [nVq]
    uint32_t nT, nVq;
};

// list that holds stream position o: the last l with listP[l] <= o (empty lists share their prefix with the successor, so that list is non-empty)
__device__ inline uint64_t kmerListOf(const uint64_t *listP, uint64_t nLists, uint64_t o) {
    uint64_t lo = 0, hi = nLists;
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (listP[mid] <= o) lo = mid; else hi = mid; }
    return lo;
}
// first and last list of every emit workgroup (sub-tile s of tile T covers kEmitTile stream positions from tileStart[T] + s * kEmitTile on), one thread per
// bound.  (Until round 5 two threads of every emit workgroup ran these searches themselves: 27 dependent global loads before the other 254 could start.)
__global__ void k_kmer_tile_lists(const uint64_t *listP, uint64_t nLists, const uint32_t *tileStart, uint32_t nT, uint32_t *tileL /*[2 * nT * kSubTiles]*/) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2u * nT * kSubTiles) return;
    const uint32_t b = i >> 1, T = b / kSubTiles, s = b % kSubTiles;
    const uint64_t o0 = (uint64_t) tileStart[T] + (uint64_t) s * kEmitTile, tEnd = tileStart[T + 1];
    if (o0 >= tEnd) { tileL[i] = 0; return; }
    const uint64_t o = (i & 1u) ? min(tEnd, o0 + kEmitTile) - 1 : o0;
    tileL[i] = (uint32_t) kmerListOf(listP, nLists, o);          // list slots of a batch fit 32 bits (listStart / listSize / listPos are indexed by them)
}

// Output-balanced gather: one thread per hit, list found by binary search over the staged list prefixes.
// E = uint64_t: entries seqId << 16 | position; E = uint32_t: seqId << posBits | position (k_kmer_compact_entries32)
template <class E>
__global__ __launch_bounds__(256) void k_kmer_emit(uint64_t nLists, const uint64_t *listP, const uint32_t *listStart, const uint32_t *listPos /* query << 16 | position */,
                                                   const uint32_t *tileL, const E *entries, int posBits, const uint32_t *tileStart, KmerCoarse co, int blkInLds,
                                                   uint32_t *cntA /*[nT][nKeys]*/, uint32_t *rec, uint16_t *recKey) {
    // list prefixes relative to the tile's first stream position, clamped below at 0 (only the tile's first list can start before the tile): < kEmitTile,
    // 16 bits each -- 12 KB instead of 24, so the workgroups per CU are bounded by their waves, not by LDS
    __shared__ uint16_t rel[kEmitStage + 1];
    __shared__ __attribute__((aligned(16))) uint16_t owner[kEmitTile];       // list (relative to the tile's first) of every stream position of the tile
    __shared__ uint32_t hist[kMaxCoarse], wmax[4];
    extern __shared__ uint16_t sblk[];        // [nBlk] when blkInLds
    const uint32_t T = blockIdx.x / kSubTiles, sub = blockIdx.x % kSubTiles;
    const uint64_t o0 = (uint64_t) tileStart[T] + (uint64_t) sub * kEmitTile, tEnd = tileStart[T + 1];
    if (o0 >= tEnd) return;
    const uint64_t o1 = min(tEnd, o0 + kEmitTile);
    const uint64_t l0 = tileL[2 * blockIdx.x], l1 = tileL[2 * blockIdx.x + 1];
    const uint64_t p0 = listP[l0];
    const int nl = (int) min<uint64_t>(l1 - l0 + 1, (uint64_t) kEmitStage + 1);
    const bool staged = l1 - l0 + 1 <= (uint64_t) kEmitStage;
    for (uint32_t i = threadIdx.x; i < co.nKeys; i += 256) hist[i] = 0;
    if (blkInLds) for (uint32_t i = threadIdx.x; i < co.nBlk; i += 256) sblk[i] = co.blkKey[i];
    if (staged) for (int i = threadIdx.x; i < nl; i += 256) { const uint64_t lp = listP[l0 + i]; rel[i] = (uint16_t) (lp > o0 ? lp - o0 : 0); }
    constexpr int U = kEmitTile / 256;
    static_assert(U == 8, "k_kmer_emit reads a thread's eight list owners as one 16-byte word");
    reinterpret_cast<uint4 *>(owner)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // 8 outputs per thread, handled phase by phase so that the dependent loads of all 8 are in flight together
    uint64_t l[U];
    if (staged) {
        // The list of every stream position of the tile, WITHOUT a search per position (until round 6 every thread ran eight 13-step binary searches over the
        // staged prefixes: 80 of the kernel's ~170 wave instructions per 64 hits, and the kernel is as close to its issue limit as to the HBM's): every
        // non-empty list marks its first position with its index, a running maximum over the positions spreads it.  Thread t scans positions 8 t .. 8 t + 7, a
        // wave scan and the four wave totals carry the maximum across; the result goes back through LDS because the gathers and the stores below keep the
        // interleaved mapping (position = thread + 256 u: consecutive lanes, consecutive addresses).
        const uint32_t tileLen = (uint32_t) (o1 - o0);
        for (int i = threadIdx.x; i < nl; i += 256) {
            const uint32_t sPos = rel[i], ePos = i + 1 < nl ? (uint32_t) rel[i + 1] : tileLen;
            if (ePos > sPos) owner[sPos] = (uint16_t) i;
        }
        __syncthreads();
        const uint4 ow = reinterpret_cast<const uint4 *>(owner)[threadIdx.x];
        uint32_t mx[U];
        mx[0] = ow.x & 0xffffu; mx[1] = max(mx[0], ow.x >> 16);
        mx[2] = max(mx[1], ow.y & 0xffffu); mx[3] = max(mx[2], ow.y >> 16);
        mx[4] = max(mx[3], ow.z & 0xffffu); mx[5] = max(mx[4], ow.z >> 16);
        mx[6] = max(mx[5], ow.w & 0xffffu); mx[7] = max(mx[6], ow.w >> 16);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t incl = mx[7];
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t v = (uint32_t) __shfl_up((int) incl, dd); if (lane >= dd) incl = max(incl, v); }
        uint32_t before = (uint32_t) __shfl_up((int) incl, 1);
        if (lane == 0) before = 0;
        if (lane == 63) wmax[wave] = incl;
        __syncthreads();
        for (int w = 0; w < wave; w++) before = max(before, wmax[w]);
#pragma unroll
        for (int u = 0; u < U; u++) mx[u] = max(mx[u], before);
        reinterpret_cast<uint4 *>(owner)[threadIdx.x] = make_uint4(mx[0] | (mx[1] << 16), mx[2] | (mx[3] << 16), mx[4] | (mx[5] << 16), mx[6] | (mx[7] << 16));
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; u++) l[u] = l0 + owner[min((uint32_t) (threadIdx.x + 256 * u), tileLen - 1u)];
    } else {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t o = min<uint64_t>(o0 + threadIdx.x + 256 * u, o1 - 1);
            uint64_t lo = l0, hi = l1 + 1;
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (listP[mid] <= o) lo = mid; else hi = mid; }
            l[u] = lo;
        }
    }
    // two dependent global round trips per hit: (list record) -> (index entry); the list's first stream position comes from the staged prefixes
    uint32_t p[U], st[U];
    uint64_t lp[U];
#pragma unroll
    for (int u = 0; u < U; u++) { p[u] = listPos[l[u]]; st[u] = listStart[l[u]]; lp[u] = !staged ? listP[l[u]] : l[u] == l0 ? p0 : o0 + rel[(uint32_t) (l[u] - l0)]; }
    E e[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint64_t o = o0 + threadIdx.x + 256 * u;
        e[u] = o < o1 ? entries[(uint64_t) st[u] + (o - lp[u])] : (E) 0;
    }
    const int pb = sizeof(E) == 8 ? 16 : posBits;
    const uint32_t pmask = (1u << pb) - 1u;
    uint32_t key[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const uint32_t b = (uint32_t) (e[u] >> pb) >> 10; key[u] = blkInLds ? (uint32_t) sblk[b] : (uint32_t) co.blkKey[b]; }      // (entry 0 -> block 0: harmless)
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint64_t o = o0 + threadIdx.x + 256 * u;
        if (o >= o1) continue;
        const uint32_t posj = (uint32_t) e[u] & pmask, t = (uint32_t) (e[u] >> pb);
        rec[o] = (t << 16) | (((p[u] & 0xffffu) - posj) & 0xffffu);
        recKey[o] = (uint16_t) key[u];
        atomicAdd(&hist[key[u]], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < co.nKeys; i += 256) { const uint32_t c = hist[i]; if (c) atomicAdd(&cntA[(size_t) T * co.nKeys + i], c); }
}

// block-wide in-place scan (NT threads) over an array in LDS or global memory: every thread owns a contiguous run (one block scan over the
// run sums instead of one per NT elements)
template <class T, bool INCLUSIVE, int NT>
__device__ inline uint32_t kmerBlockScan(T *v, uint32_t n, uint32_t *wsum /*[NT / 64]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per = (n + NT - 1) / NT, i0 = threadIdx.x * per, i1 = min(n, i0 + per);
    uint32_t x = 0;
    for (uint32_t i = i0; i < i1; i++) x += (uint32_t) v[i];
    uint32_t incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = incl - x, total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) { const uint32_t ws = wsum[w]; if (w < wave) before += ws; total += ws; }
    for (uint32_t i = i0; i < i1; i++) { const uint32_t y = (uint32_t) v[i]; before += INCLUSIVE ? y : 0u; v[i] = (T) before; before += INCLUSIVE ? 0u : y; }
    __syncthreads();
    return total;
}

// Column kernels: cnt[T][key] -> where the run of (T, key) starts in the output of the stable scatter.  The output is laid out (query, key, tile), so
// the prefix runs DOWN a column of the query's rows.  A workgroup = one query x 64 keys x kColGroups groups of consecutive tiles: pass 1 sums every
// group (-> segment sizes: one scan over the (query, key) pairs of the batch in between), pass 2 walks the groups again and leaves the starts in
// place of the counts, plus a transposed copy (column-contiguous) in which k_kmer_dup_stream looks a position's tile up.
__global__ __launch_bounds__(64 * kColGroups) void k_kmer_col_sums(const uint32_t *cntA, const uint32_t *qTile0, uint32_t nKeys, uint32_t *grpSum /*[nq][kColGroups][nKeys]*/,
                                                                    uint32_t *segCnt /*[nq * nKeys]*/) {
    __shared__ uint32_t sh[kColGroups][64];
    const uint32_t q = blockIdx.x, kx = threadIdx.x & 63u, g = threadIdx.x >> 6, key = blockIdx.y * 64u + kx;
    const uint32_t T0 = qTile0[q], T1 = qTile0[q + 1], per = (T1 - T0 + kColGroups - 1) / kColGroups;
    const uint32_t a = min(T1, T0 + g * per), b = min(T1, a + per);
    uint32_t s = 0;
    if (key < nKeys) for (uint32_t T = a; T < b; T++) s += cntA[(size_t) T * nKeys + key];
    sh[g][kx] = s;
    __syncthreads();
    if (key >= nKeys) return;
    grpSum[((size_t) q * kColGroups + g) * nKeys + key] = s;
    if (g == 0) { uint32_t tot = 0; for (int i = 0; i < kColGroups; i++) tot += sh[i][kx]; segCnt[(size_t) q * nKeys + key] = tot; }
}
__global__ __launch_bounds__(64 * kColGroups) void k_kmer_col_offsets(uint32_t *cntA /* in: counts, out: run starts */, const uint32_t *qTile0, uint32_t nKeys, const uint32_t *grpSum,
                                                                       const uint32_t *segStart /*[nq * nKeys + 1]*/, uint32_t *colA /*[qTile0[q] * nKeys + key * tiles(q) + tile]*/) {
    const uint32_t q = blockIdx.x, kx = threadIdx.x & 63u, g = threadIdx.x >> 6, key = blockIdx.y * 64u + kx;
    if (key >= nKeys) return;
    const uint32_t T0 = qTile0[q], T1 = qTile0[q + 1], nTq = T1 - T0, per = (nTq + kColGroups - 1) / kColGroups;
    const uint32_t a = min(T1, T0 + g * per), b = min(T1, a + per);
    uint32_t base = segStart[(size_t) q * nKeys + key];
    for (uint32_t i = 0; i < g; i++) base += grpSum[((size_t) q * kColGroups + i) * nKeys + key];
    uint32_t *col = colA + (size_t) T0 * nKeys + (size_t) key * nTq;
    for (uint32_t T = a; T < b; T++) {
        const uint32_t c = cntA[(size_t) T * nKeys + key];
        cntA[(size_t) T * nKeys + key] = base;
        col[T - T0] = base;
        base += c;
    }
}

// Stable scatter: one workgroup per tile, kScStage records per pass through LDS.  A wave owns a contiguous eighth of the pass and ranks its
// records in lane order: the lanes of a wave instruction that share a key find each other with one ballot per key bit (KB of them, a template
// parameter: the loop must unroll -- as a loop it cost 16 instructions per bit and made the kernel issue-bound at 209 wave instructions per 64
// records), the lowest of them adds the group's size to the wave's private counter of that key (one lane per key and instruction, and a wave's
// LDS accesses execute in order), everybody reads the counter back: rank = counter - group size + position among the group's lower lanes.
// A cross-wave prefix per key and a scan over the keys turn the ranks into LDS positions; the runs leave LDS key by key as consecutive 4-byte
// stores into the (tile, key) runs of the output, whose starts the column kernels computed.
template <int KB>
__global__ __launch_bounds__(kScThreads) void k_kmer_scatter_stable(const uint32_t *rec, const uint16_t *recKey, const uint32_t *tileStart, const uint32_t *offA, uint32_t nKeys,
                                                                     uint32_t *out, uint16_t *ordOut) {
    constexpr int NW = kScThreads / 64, U = kScStage / kScThreads, PW = kScStage / NW;
    __shared__ uint32_t srec[kScStage];
    __shared__ uint16_t skey[kScStage], sord[kScStage];
    __shared__ uint16_t whist[NW][kMaxCoarse];
    __shared__ uint32_t lbase[kMaxCoarse + 1], gcur[kMaxCoarse];
    __shared__ uint32_t wsum[NW];
    const uint32_t T = blockIdx.x;
    const uint32_t a = tileStart[T], n = tileStart[T + 1] - a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t k = threadIdx.x; k < nKeys; k += kScThreads) {
        gcur[k] = offA[(size_t) T * nKeys + k];
#pragma unroll
        for (int w = 0; w < NW; w++) whist[w][k] = 0;
    }
    __syncthreads();
    const uint32_t ltLo = lane < 32 ? (1u << lane) - 1u : 0xffffffffu, ltHi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
    uint16_t *myHist = whist[wave];
    uint32_t r[U], k[U], rn[U], kn[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t e = (uint32_t) (wave * PW + u * 64 + lane);
        r[u] = e < n ? rec[a + e] : 0u;
        k[u] = e < n ? (uint32_t) recKey[a + e] : 0u;
    }
    for (uint32_t sp0 = 0; sp0 < n; sp0 += kScStage) {
        const uint32_t m = min((uint32_t) kScStage, n - sp0);
        uint32_t rk[U];
#pragma unroll
        for (int u = 0; u < U; u++) {          // the next pass's records are in flight while this pass goes through LDS
            const uint32_t e = sp0 + kScStage + (uint32_t) (wave * PW + u * 64 + lane);
            rn[u] = e < n ? rec[a + e] : 0u;
            kn[u] = e < n ? (uint32_t) recKey[a + e] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t e = (uint32_t) (wave * PW + u * 64 + lane);
            const bool valid = e < m;
            const unsigned long long vm = __ballot(valid);
            uint32_t lo = (uint32_t) vm, hi = (uint32_t) (vm >> 32);            // lanes with this lane's key (an invalid lane's own set is not used)
#pragma unroll
            for (int bit = 0; bit < KB; bit++) {
                const bool s = (k[u] & (1u << bit)) != 0u;
                const unsigned long long mb = __ballot(s);
                const uint32_t same = s ? 0u : 0xffffffffu;                       // set -> the lanes of mb, clear -> the others
                lo &= (uint32_t) mb ^ same; hi &= (uint32_t) (mb >> 32) ^ same;
            }
            const uint32_t below = (uint32_t) __popc(lo & ltLo) + (uint32_t) __popc(hi & ltHi), cnt = (uint32_t) __popc(lo) + (uint32_t) __popc(hi);
            if (valid && below == 0u) myHist[k[u]] = (uint16_t) (myHist[k[u]] + cnt);
            asm volatile("" ::: "memory");     // the counter is read back by the other lanes of the group
            rk[u] = valid ? (uint32_t) myHist[k[u]] - cnt + below : 0u;
        }
        __syncthreads();
        for (uint32_t kk = threadIdx.x; kk < nKeys; kk += kScThreads) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) { const uint32_t c = whist[w][kk]; whist[w][kk] = (uint16_t) run; run += c; }
            lbase[kk] = run;
        }
        if (threadIdx.x == 0) lbase[nKeys] = 0;
        __syncthreads();
        kmerBlockScan<uint32_t, false, kScThreads>(lbase, nKeys + 1, wsum);          // lbase[key] = first LDS position of the key, lbase[nKeys] = m
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t e = (uint32_t) (wave * PW + u * 64 + lane);
            if (e < m) { const uint32_t p = lbase[k[u]] + myHist[k[u]] + rk[u]; srec[p] = r[u]; skey[p] = (uint16_t) k[u]; sord[p] = (uint16_t) e; }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < m; i += kScThreads) {
            const uint32_t kk = skey[i];
            const uint32_t dst = gcur[kk] + (i - lbase[kk]);
            out[dst] = srec[i];
            ordOut[dst] = (uint16_t) (sp0 + sord[i]);
        }
        __syncthreads();
        for (uint32_t kk = threadIdx.x; kk < nKeys; kk += kScThreads) {
            gcur[kk] += lbase[kk + 1] - lbase[kk];
#pragma unroll
            for (int w = 0; w < NW; w++) whist[w][kk] = 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) { r[u] = rn[u]; k[u] = kn[u]; }
        __syncthreads();
    }
}

// The double-diagonal rule on a (query, chunk, key) run of the scatter's output: `tab` IS the reference's duplicateBitArray for the targets of the
// key -- one byte per target, the 8-bit diagonal of its last hit, zero at the start of a chunk -- and a hit is flagged when its diagonal equals
// the byte it finds (findDuplicates: currDiagonal == prevDiagonal).  ONE WAVE walks a run in arrival order, 64 hits a step, with no barrier
// anywhere (a wave's LDS accesses execute in order):
//   old = tab[t];  tab[t] = lane;  back = tab[t];
// -- the table itself tells whether two hits of the step share a target: then one of them reads the other's lane back.  No such lane (most
// steps): flag = old == d, tab[t] = d.  Otherwise the lanes that read a foreign lane and the lanes they read are set aside, everybody else
// proceeds as before (their targets are theirs alone), the set-aside lanes put the old byte back and then take their turn one after the other
// in lane = arrival order: read the byte, compare, replace -- the reference's loop, literally, for the two to four hits concerned.
// About 20 wave instructions per 64 hits on the common path.  What bounds the kernel is the chain of four LDS round trips per step times the
// steps of a run, hidden by the other waves of the CU: the number of concurrent runs is what LDS holds tables for (a byte per target id of the
// key + 1.5 KB per wave).
// Flagged hits (2-3 %) are noted as 4-byte positions in the run's own range of the emit records' buffer (dead since the scatter; a run has
// room for every one of its hits), the run's count goes to runCand.  No atomics: a reservation in a batch-wide candidate array per 64 flagged
// hits was 67 k atomic adds on one address per batch, which is what the kernel then waited for (1.08 ms against 0.41 ms without them); one per
// run still cost a third of the kernel.  k_kmer_cand_gather turns the notes into candidates at the scanned offsets.
struct KmerDupStream {
    const uint32_t *recA; const uint16_t *ordA;
    uint32_t *note;               // [hits of the batch] positions of the flagged hits of a run, at the run's own range
    const uint32_t *offA, *colA, *segStart;
    KmerTiles tl;
    const KmerQ *qs;
    const uint32_t *keyFirst;
    uint32_t nKeys;
    int tbits;
    uint32_t *runCand;            // [runs] flagged hits of every run (zeroed by the caller: an empty run returns early)
    const uint32_t *runBase;      // [runs + 1] their exclusive prefix: k_kmer_cand_gather
    uint32_t *candKey; uint64_t *candVal;
    uint32_t *ecCount;            // [nq][kMaxChunks] candidates per (query, chunk)
};
// the run of workgroup / wave `unit`: false when it is empty.  Runs are numbered (query, key, chunk): the flagged hits of all runs, one run after
// the other, then stand in (query, key) blocks with the stream positions ascending inside a block -- a STABLE sort by (query, target) is all that
// separates them from the order the scoring and replay stages need
struct KmerRun { uint32_t q, chunk, A, T0, nTc, p0, p1, first, lowFirst; const uint32_t *col; };
__device__ inline bool kmerRunOf(const KmerDupStream &a, uint32_t unit, KmerRun &r) {
    r.q = a.tl.vqQ[unit / a.nKeys];           // the runs of a query are numbered from (its first chunk pair) x keys on, whatever their order inside
    const uint32_t vq0 = a.tl.qVq0[r.q], nCh = a.tl.qVq0[r.q + 1] - vq0, local = unit - vq0 * a.nKeys;
    r.A = local / nCh; r.chunk = local - r.A * nCh;
    const uint32_t vq = vq0 + r.chunk;
    const uint32_t T0 = a.tl.vqTile0[vq], T1 = a.tl.vqTile0[vq + 1], qT0 = a.tl.qTile0[r.q], qT1 = a.tl.qTile0[r.q + 1];
    if (T0 == T1) return false;
    r.T0 = T0; r.nTc = T1 - T0;
    r.p0 = a.offA[(size_t) T0 * a.nKeys + r.A];
    r.p1 = T1 < qT1 ? a.offA[(size_t) T1 * a.nKeys + r.A] : a.segStart[(size_t) r.q * a.nKeys + r.A + 1];
    r.first = a.keyFirst[r.A]; r.lowFirst = r.first & 0xffffu;
    r.col = a.colA + (size_t) qT0 * a.nKeys + (size_t) r.A * (qT1 - qT0) + (T0 - qT0);          // run starts of the chunk's tiles in this key
    return r.p0 < r.p1;
}
constexpr int kDupColStage = 128;             // tiles of a chunk whose column slice is staged in LDS (a chunk of 2 x 10^6 hits has 123)
constexpr int kDupAhead = 8;                  // steps whose records are in flight
__global__ __launch_bounds__(64) void k_kmer_dup_stream(KmerDupStream a) {
    extern __shared__ uint32_t tabWords[];    // [maxIds / 4 + 1]: one byte per target id of the key
    uint8_t *tab = reinterpret_cast<uint8_t *>(tabWords);
    // what a lane reads back from a byte it has just written is what ANOTHER lane may have written there: the compiler must not forward the stored
    // value to the load (a volatile pointer would do that too, but turns the accesses into flat_* instructions)
#define FS_LDS_REREAD() asm volatile("" ::: "memory")
    KmerRun run;
    if (!kmerRunOf(a, blockIdx.x, run)) return;
    const uint32_t p0 = run.p0, p1 = run.p1, lowFirst = run.lowFirst;
    const uint32_t ids = a.keyFirst[run.A + 1] - run.first;
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < ids / 4 + 1; i += 64) tabWords[i] = 0;
    uint32_t nSt = 0;                          // flagged hits so far (wave-uniform)
    constexpr int R = kDupAhead;
    uint32_t cur[R], nxt[R];
#pragma unroll
    for (int k = 0; k < R; k++) { const uint32_t pos = p0 + k * 64 + lane; cur[k] = pos < p1 ? a.recA[pos] : 0u; }
    for (uint32_t base = p0; base < p1; base += R * 64) {
#pragma unroll
        for (int k = 0; k < R; k++) { const uint32_t pos = base + (R + k) * 64 + lane; nxt[k] = pos < p1 ? a.recA[pos] : 0u; }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const uint32_t sbase = base + k * 64;
            if (sbase >= p1) break;            // uniform
            const uint32_t pos = sbase + lane;
            const bool valid = pos < p1;
            const uint32_t t = ((cur[k] >> 16) - lowFirst) & 0xffffu, d = cur[k] & 0xffu;
            uint32_t old = 0, back = lane;
            if (valid) { old = tab[t]; tab[t] = (uint8_t) lane; FS_LDS_REREAD(); back = tab[t]; }
            const unsigned long long losers = __ballot(back != lane);
            bool flag = false;
            if (losers == 0ull) {
                if (valid) { flag = old == d; tab[t] = (uint8_t) d; }
            } else {
                unsigned long long inv = losers, m = losers;
                while (m) { const int j = __ffsll((long long) m) - 1; m &= m - 1ull; inv |= 1ull << (uint32_t) __builtin_amdgcn_readlane((int) back, j); }
                const bool involved = (inv >> lane) & 1ull;
                if (valid && !involved) { flag = old == d; tab[t] = (uint8_t) d; }
                if (involved) tab[t] = (uint8_t) old;
                m = inv;
                while (m) {
                    const int j = __ffsll((long long) m) - 1; m &= m - 1ull;
                    FS_LDS_REREAD();
                    if ((int) lane == j) { const uint32_t p = tab[t]; flag = p == d; tab[t] = (uint8_t) d; }
                }
                FS_LDS_REREAD();
            }
            const unsigned long long fm = __ballot(flag);
            if (fm) {
                if (flag) a.note[p0 + nSt + (uint32_t) __popcll(fm & ((1ull << lane) - 1ull))] = pos;
                nSt += (uint32_t) __popcll(fm);
            }
        }
#pragma unroll
        for (int k = 0; k < R; k++) cur[k] = nxt[k];
    }
    if (lane == 0 && nSt) { a.runCand[blockIdx.x] = nSt; atomicAdd(&a.ecCount[(size_t) run.q * kMaxChunks + run.chunk], nSt); }
#undef FS_LDS_REREAD
}

// the flagged hits of all runs -> (query | target) keys and (stream position, diagonal, chunk) values at the runs' places in the candidate array (exclusive
// scan of runCand).  One wave per run; the stream position of a hit = first position of the tile whose run holds it (search over the chunk's slice of the key's
// column, staged in LDS) + its 16-bit position inside the tile.
__global__ __launch_bounds__(256) void k_kmer_cand_gather(KmerDupStream a, uint32_t nRuns) {
    __shared__ uint32_t colAll[4][kDupColStage];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, unit = blockIdx.x * 4u + wv;
    if (unit >= nRuns) return;
    const uint32_t n = a.runCand[unit];
    if (n == 0) return;
    KmerRun run;
    if (!kmerRunOf(a, unit, run)) return;
    uint32_t *colS = colAll[wv];
    const bool colStaged = run.nTc <= (uint32_t) kDupColStage;
    if (colStaged) for (uint32_t i = lane; i < run.nTc; i += 64) colS[i] = run.col[i];           // (read by the same wave only: its LDS accesses execute in order)
    const uint32_t hb = (uint32_t) a.qs[run.q].hitBase, fb = a.runBase[unit];
    const uint32_t keyHi = (run.q << a.tbits) | run.first;
    for (uint32_t k = lane; k < n; k += 64) {
        const uint32_t pos = a.note[run.p0 + k];
        const uint32_t rr = a.recA[pos];
        uint32_t lo = 0, hi = run.nTc;          // last tile of the chunk whose run starts at or before pos (empty runs share their start with the next one)
        if (colStaged) { while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (colS[mid] <= pos) lo = mid; else hi = mid; } }
        else { while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (run.col[mid] <= pos) lo = mid; else hi = mid; } }
        const uint32_t g = a.tl.tileStart[run.T0 + lo] + (uint32_t) a.ordA[pos] - hb;
        a.candKey[fb + k] = keyHi + (((rr >> 16) - run.lowFirst) & 0xffffu);
        a.candVal[fb + k] = hitPack(g, rr & 0xffffu, run.chunk);
    }
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
    uint64_t wNext = pos + 8 <= len ? *reinterpret_cast<const uint64_t *>(db + pos) : 0;
    for (; pos + 8 <= len; pos += 8) {
        const uint64_t w = wNext;
        if (pos + 16 <= len) wNext = *reinterpret_cast<const uint64_t *>(db + pos + 8);      // in flight while these 8 cells are scored
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

// The same scores with EIGHT lanes per candidate (k_kmer_score gives every candidate one lane: 64 lanes of a wave then stream 64 different
// targets eight bytes at a time -- every 8-byte load opens its own 64-byte sector, the lines of 64 targets per wave and 32 waves per CU do
// not survive in L2 until their next 8 bytes are wanted, and the pass fetched 2.7x the diagonals' residues).  Here the diagonal is cut into
// eight contiguous parts of whole 8-byte words; lane g runs the reference's recurrence (s = max(0, s + x), best = max(best, s):
// UngappedAlignment.cpp:36-120, scalar form) over part g and keeps FOUR numbers that describe the part as a function of the run value s it
// is entered with:  A = sum,  S0 = end value from s = 0,  P = best from s = 0,  Q = best prefix sum  (end(s) = max(S0, s + A), best(s) =
// max(P, s + Q)).  Parts compose associatively, in order:  (L . R):  A = AL + AR,  S0 = max(S0R, S0L + AR),  P = max(PL, PR, S0L + QR),
// Q = max(QL, AL + QR) -- three butterfly steps over the eight lanes give the whole diagonal's P, exactly the sequential result.
struct KmerSum { int A, S0, P, Q; };
__device__ __forceinline__ KmerSum kmerSumJoin(const KmerSum &l, const KmerSum &r) {
    KmerSum o;
    o.A = l.A + r.A;
    o.S0 = max(r.S0, l.S0 + r.A);
    o.P = max(max(l.P, r.P), l.S0 + r.Q);
    o.Q = max(l.Q, l.A + r.Q);
    return o;
}

// part g (of eight) of one diagonal: see k_kmer_score8.  Inlined once per address space of `prof`.
__device__ __forceinline__ KmerSum kmerScorePart(const int8_t *prof, const uint8_t *db, int len, int g) {
    // positions [0, head) up to the first 8-byte boundary of the target go to lane 0 byte by byte (they are the leftmost part);
    // the aligned words behind them are dealt out in eight contiguous runs.  No address below `prof` / `db` is ever formed: a flat
    // access whose base register lies under the LDS aperture faults even when base + offset does not.
    const int head = min(len, (int) ((8 - ((uintptr_t) db & 7)) & 7));
    int A = 0, S0 = 0, P = 0, Q = 0;
    if (g == 0)
        for (int pos = 0; pos < head; pos++) { const int x = prof[pos * 21 + db[pos]]; A += x; Q = max(Q, A); S0 = max(0, S0 + x); P = max(P, S0); }
    const uint64_t *words = reinterpret_cast<const uint64_t *>(db + head);
    const int rest = len - head, nW = (rest + 7) >> 3, per = (nW + 7) >> 3;
    const int w0 = g * per, w1 = min(nW, w0 + per);
    uint64_t wNext = w0 < w1 ? words[w0] : 0;
    for (int w = w0; w < w1; w++) {
        const uint64_t cur = wNext;
        if (w + 1 < w1) wNext = words[w + 1];
        const int p0 = head + 8 * w;                       // first position of this word, >= 0
        const int8_t *pp = prof + p0 * 21;
        const uint32_t lo = (uint32_t) cur, hi = (uint32_t) (cur >> 32);
        int x[8];
        if (p0 + 8 <= len) {
#pragma unroll
            for (int b = 0; b < 8; b++) x[b] = pp[b * 21 + (int) (((b < 4 ? lo : hi) >> (8 * (b & 3))) & 0xffu)];
        } else {
#pragma unroll
            for (int b = 0; b < 8; b++) x[b] = (p0 + b < len) ? (int) pp[b * 21 + (int) (((b < 4 ? lo : hi) >> (8 * (b & 3))) & 0xffu)] : 0;
        }
#pragma unroll
        for (int b = 0; b < 8; b++) { A += x[b]; Q = max(Q, A); const int t = S0 + x[b]; P = max(P, t); S0 = max(0, t); }
    }
    return KmerSum{A, S0, P, Q};
}

__global__ __launch_bounds__(256) void k_kmer_score8(const uint32_t *ckeys, const uint64_t *cvals, const uint32_t *nCandPtr, int tbits, const KmerQ *qs,
                                                     const int8_t *profiles, const uint8_t *masked, const uint64_t *offsets, const int32_t *lengths,
                                                     int ldsBytes, uint8_t *kept, int32_t *score) {
    extern __shared__ int8_t sprof[];
    __shared__ uint32_t q0s;
    const uint64_t nCand = *nCandPtr;
    const uint64_t j0 = (uint64_t) blockIdx.x * 32;
    if (j0 >= nCand) return;
    if (threadIdx.x == 0) q0s = ckeys[j0] >> tbits;
    __syncthreads();
    const uint32_t q0 = q0s;
    const int L0 = (int) qs[q0].L;
    const bool staged = L0 * 21 <= ldsBytes;
    if (staged) {
        const int8_t *src = profiles + qs[q0].profOff;
        const int nb = L0 * 21;
        for (int i = threadIdx.x; i < nb; i += 256) sprof[i] = src[i];
    }
    __syncthreads();
    const int g = threadIdx.x & 7;
    const uint64_t j = j0 + (threadIdx.x >> 3);
    const bool live = j < nCand;
    bool keep = false;
    KmerSum sum{0, 0, 0, 0};
    if (live) {
        const uint32_t k = ckeys[j];
        const uint64_t v = cvals[j];
        keep = true;
        if (j > 0 && ckeys[j - 1] == k) { const uint64_t pv = cvals[j - 1]; if (hitChunk(pv) == hitChunk(v) && hitD8(pv) == hitD8(v)) keep = false; }
        if (keep) {
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
            // two copies of the loop, one per address space of the profile: a pointer that may be LDS or global memory makes every profile byte a
            // flat_load with a 64-bit address (four address instructions per cell, and flat accesses to LDS go through the vector memory pipe);
            // the staged copy reads ds_read_i8 at a 32-bit address + immediate
            if (len > 0) {
                if (staged && qi == q0) sum = kmerScorePart(sprof + poff, db, len, g);
                else sum = kmerScorePart(profiles + q.profOff + poff, db, len, g);
            }
        }
    }
    // ordered reduction over the eight lanes of a candidate (lane g holds part g; at distance d the lower lane's block is the left operand)
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        KmerSum o;
        o.A = __shfl_xor(sum.A, d); o.S0 = __shfl_xor(sum.S0, d); o.P = __shfl_xor(sum.P, d); o.Q = __shfl_xor(sum.Q, d);
        sum = (g & d) ? kmerSumJoin(o, sum) : kmerSumJoin(sum, o);
    }
    if (live && g == 0) { kept[j] = keep ? 1 : 0; score[j] = keep ? sum.P : 0; }
}

// --------------------------------------------------------------------------------------------------------------
// search, stage 4: per-target replay of QueryMatcher::match's overflow rounds + final keepMaxScoreElementOnly.
// One thread per (query, target) segment of the candidate array.  List elements are (candidate index << 8 | count).
// --------------------------------------------------------------------------------------------------------------
struct KmerBest {          // written at the segment head; nElems == 0xFFFFFFFF marks "not a head"
    uint32_t nElems;       // elements this target contributes to resultSize (first max + later zero-score ones)
    uint32_t cand;         // candidate index of the kept element
    uint32_t count;        // its 8-bit score
    uint32_t pad;          // k_kmer_walk: length of the target's final list, bit 31 = it lives in the second scratch array (0 elsewhere)
};

// FF: rounds in which a target takes no new candidates are skipped (the per-round element counts of the skipped rounds go through a difference array)
template <bool FF>
__global__ __launch_bounds__(128) void k_kmer_walk(const uint32_t *ckeys, const uint64_t *cvals, const uint8_t *kept, const int32_t *score, const uint32_t *nCandPtr,
                            int tbits, const KmerChunks *chunks, uint64_t *scrA, uint64_t *scrB, KmerBest *best,
                            uint32_t *roundCount /*[nq][kMaxChunks]*/, unsigned long long *resultSize /*[nq]*/) {
    __shared__ uint32_t rc[kMaxChunks];       // per-round element counts / result size of the block's first query
    __shared__ uint32_t rcd[kMaxChunks + 2];  // ... and their difference array: +1 at the first, -1 behind the last of a run of skipped rounds
    __shared__ uint32_t rcw[2];
    __shared__ unsigned long long rs;
    __shared__ uint32_t q0;
    const uint64_t s = (uint64_t) blockIdx.x * 128 + threadIdx.x;
    const uint64_t nCand = *nCandPtr;
    rc[threadIdx.x] = 0; rc[threadIdx.x + 128] = 0;
    rcd[threadIdx.x] = 0; rcd[threadIdx.x + 128] = 0; if (threadIdx.x < 2) rcd[256 + threadIdx.x] = 0;
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
    bool settled = false;                                    // the list went through a merge round since it last took new candidates
    for (uint32_t j = 1; j <= C; j++) {
        if (FF) {
            // Rounds without new candidates (a target's candidates sit in one or two of the query's chunks): with no elements carried they do
            // nothing; a single element that has been through a merge round comes out of every further one unchanged (scored elements survive
            // mergeDiagonalKeepScoredHitsDuplicates, and keepMaxScoreElementOnly keeps the only element) and counts once per round.
            const uint32_t cn = (pos < nCand && ckeys[pos] == key) ? hitChunk(cvals[pos]) : 0xFFFFFFFFu;      // chunks ascend inside a target's run
            if (cn >= j) {
                if (nE == 0) { if (cn >= C) break; j = cn; settled = false; continue; }                      // next round to do work: cn + 1
                if (nE == 1 && settled) {
                    const uint32_t jend = min(cn, C);                                                        // rounds j .. jend are no-ops
                    if (mine) { atomicAdd(&rcd[j], 1u); atomicAdd(&rcd[jend + 1], 0xFFFFFFFFu); }
                    else for (uint32_t jj = j; jj <= jend; jj++) atomicAdd(&roundCount[(size_t) qi * kMaxChunks + jj], 1u);
                    j = jend; continue;
                }
            }
        }
        uint32_t nS = nE;
        while (pos < nCand && ckeys[pos] == key && hitChunk(cvals[pos]) == j - 1) { if (kept[pos]) A[nS++] = pos << 8; pos++; }
        settled = j > 1;
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
    b.nElems = nF; b.cand = (uint32_t) (first >> 8); b.count = (uint32_t) first & 0xffu;
    b.pad = nL | (Lst == B ? 0x80000000u : 0u);             // where the final list lives: k_kmer_out walks it again when the cut is 0
    best[s] = b;
    if (nF) { if (mine) atomicAdd(&rs, (unsigned long long) nF); else atomicAdd(&resultSize[qi], (unsigned long long) nF); }
#undef ROUND_ADD
    }
    __syncthreads();
    if (FF) {       // prefix sum of the difference array (two entries per thread) onto the direct counts
        const uint32_t d0 = rcd[2 * threadIdx.x], d1 = rcd[2 * threadIdx.x + 1], sum = d0 + d1;
        uint32_t incl = sum;
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (lane == 63) rcw[threadIdx.x >> 6] = incl;
        __syncthreads();
        const uint32_t before = incl - sum + (threadIdx.x >= 64 ? rcw[0] : 0u);
        rc[2 * threadIdx.x] += before + d0; rc[2 * threadIdx.x + 1] += before + sum;
        __syncthreads();
    }
    for (int j = threadIdx.x; j < kMaxChunks; j += 128) if (rc[j]) atomicAdd(&roundCount[(size_t) q0 * kMaxChunks + j], rc[j]);
    if (threadIdx.x == 0 && rs) atomicAdd(&resultSize[q0], rs);
}

// --diag-score 0 (QueryMatcher with diagonalScoring == false): no diagonal scores and no replay -- findDuplicates runs with computeTotalScore
// (CacheFriendlyOperations.cpp:217-241): a target's double-diagonal candidates are COUNTED (capped at 255) and it is handed on once, with the
// diagonal of its first candidate.  One thread per candidate; the head of a (query, target) run counts the run (candidates are in (target,
// stream position) order, so the head is the first arrival).  Queries that refilled databaseHits are answered with a status by the host.
__global__ __launch_bounds__(256) void k_kmer_count_heads(const uint32_t *ckeys, const uint32_t *nCandPtr, int tbits, uint8_t *kept, int32_t *score, KmerBest *best,
                                                          unsigned long long *resultSize /*[nq]*/) {
    const uint64_t nCand = *nCandPtr;
    const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nCand) return;
    const uint32_t k = ckeys[j];
    KmerBest b;
    b.nElems = 0xFFFFFFFFu; b.cand = 0; b.count = 0; b.pad = 0;
    kept[j] = 1;
    if (j == 0 || ckeys[j - 1] != k) {
        uint32_t c = 1;
        while (c < 255 && j + c < nCand && ckeys[j + c] == k) c++;
        b.nElems = 1; b.cand = (uint32_t) j; b.count = c;
        atomicAdd(&resultSize[k >> tbits], 1ull);
    }
    score[j] = (int32_t) b.count;
    best[j] = b;
}

// --diag-score 0 and the query refilled databaseHits (QueryMatcher.cpp:311-346): every refill's findDuplicates output -- one element per target
// with candidates in that chunk: (count, diagonal of the first candidate) -- is appended to the elements of the earlier rounds and the whole list
// goes through mergeScoreDuplicates (CacheFriendlyOperations.cpp:150-180).  As that function executes: per bin the counts of a target's elements
// are summed (saturating at 255) INTO the target's byte of duplicateBitArray, every element is handed on with the byte's current value (dropped
// when that is 0), and the byte is then left at the low byte of that element's diagonal -- so a target present in two rounds comes out twice
// (the sum, then the first element's diagonal byte as "count"), and the byte one bin leaves behind is the starting value of the target of the
// next bin that shares id >> shift (the B consecutive ids of one byte; bins are walked in ascending order, i.e. ascending id inside the group).
// Elements keep their identity through the merges (target, round of origin, first candidate), only their counts move: the state of the replay
// is one count per (target, chunk) run head, kept in cnt[head candidate].  The merge of round m only runs when an earlier round left elements
// (first = first chunk of the QUERY with candidates: QueryMatcher.cpp:328 tests overflowHitCount != 0), with or without new ones.
// heads: candidate indices of the group's run heads in (target, chunk) order.  roundAdd(c, n): n elements of this group are in foundDiagonals
// when chunk c's findDuplicates starts (the host's output-capacity test).  Returns the group's elements after the last round.
template <class RoundAdd>
__host__ __device__ inline uint32_t kmerMergeScoreGroup(const uint64_t *heads, uint32_t nH, const uint32_t *ckeys, const uint64_t *cvals, int32_t *cnt,
                                                        uint32_t first, uint32_t C, RoundAdd roundAdd) {
    uint32_t alive = 0;
    for (uint32_t k = 0; k < nH; k++) alive += hitChunk(cvals[heads[k]]) <= first ? 1u : 0u;
    if (first < C) roundAdd(first + 1, alive);
    for (uint32_t m = first + 1; m <= C; m++) {
        uint32_t d = 0;
        alive = 0;
        for (uint32_t k = 0; k < nH;) {
            const uint32_t key = ckeys[heads[k]];
            uint32_t e = k;
            for (; e < nH && ckeys[heads[e]] == key; e++) {
                const uint64_t j = heads[e];
                if (hitChunk(cvals[j]) <= m && cnt[j] > 0) { d += (uint32_t) cnt[j]; d = d > 255u ? 255u : d; }
            }
            for (uint32_t x = k; x < e; x++) {
                const uint64_t j = heads[x];
                if (hitChunk(cvals[j]) <= m && cnt[j] > 0) { cnt[j] = (int32_t) d; alive += d != 0 ? 1u : 0u; d = hitD8(cvals[j]); }
            }
            k = e;
        }
        if (m < C) roundAdd(m + 1, alive);
    }
    return alive;
}

// One thread per candidate; the first candidate of a (query, id >> shift) group replays the group (pass 1: run heads and run lengths, then
// kmerMergeScoreGroup), every other thread only marks its own slot when it is not a run head.  Queries without refills come out as
// k_kmer_count_heads leaves them.  scr: 8 bytes per candidate (the group's head list lives at its first candidate's slot).
// The body is host-callable so that tests/test_kmer_merge_model.py can hold it to the reference's functions without a device.
template <class RoundAdd, class ResultAdd>
__host__ __device__ inline void kmerMergeHeadsThread(uint64_t j, const uint32_t *ckeys, const uint64_t *cvals, uint64_t nCand, int tbits, int shift, const KmerChunks *chunks,
                                                     const uint32_t *ecCount, uint64_t *scr, uint8_t *kept, int32_t *score, KmerBest *best, RoundAdd roundAdd, ResultAdd resultAdd) {
    const uint32_t key = ckeys[j], tmask = (1u << tbits) - 1u;
    const uint32_t qi = key >> tbits, grp = (key & tmask) >> shift;
    kept[j] = 1;
    bool runHead = true, groupHead = true;
    if (j > 0) {
        const uint32_t pk = ckeys[j - 1];
        runHead = pk != key || hitChunk(cvals[j - 1]) != hitChunk(cvals[j]);
        groupHead = (pk >> tbits) != qi || ((pk & tmask) >> shift) != grp;
    }
    if (!runHead) { KmerBest b; b.nElems = 0xFFFFFFFFu; b.cand = 0; b.count = 0; b.pad = 0; best[j] = b; }
    if (!groupHead) return;
    // pass 1: the group's run heads with their run lengths (findDuplicates with computeTotalScore: one count per candidate, capped at 255)
    uint64_t *heads = scr + j;
    uint32_t nH = 0;
    for (uint64_t p = j; p < nCand; p++) {
        const uint32_t k = ckeys[p];
        if ((k >> tbits) != qi || ((k & tmask) >> shift) != grp) break;
        if (p == j || ckeys[p - 1] != k || hitChunk(cvals[p - 1]) != hitChunk(cvals[p])) { heads[nH++] = p; score[p] = 1; }
        else { const uint64_t h = heads[nH - 1]; if (score[h] < 255) score[h]++; }
    }
    const uint32_t C = chunks[qi].nChunks - 1;
    uint32_t first = 0;
    while (first < C && ecCount[(size_t) qi * kMaxChunks + first] == 0) first++;
    const uint32_t alive = kmerMergeScoreGroup(heads, nH, ckeys, cvals, score, first, C, [&](uint32_t c, uint32_t n) { if (n) roundAdd(qi, c, n); });
    for (uint32_t k = 0; k < nH; k++) {
        const uint64_t h = heads[k];
        KmerBest b;
        b.nElems = score[h] > 0 ? 1u : 0u; b.cand = (uint32_t) h; b.count = (uint32_t) score[h]; b.pad = 0;
        best[h] = b;
    }
    if (alive) resultAdd(qi, alive);
}
__global__ __launch_bounds__(128) void k_kmer_merge_heads(const uint32_t *ckeys, const uint64_t *cvals, const uint32_t *nCandPtr, int tbits, int shift, const KmerChunks *chunks,
                                                          const uint32_t *ecCount /*[nq][kMaxChunks]*/, uint64_t *scr, uint8_t *kept, int32_t *score, KmerBest *best,
                                                          uint32_t *roundCount /*[nq][kMaxChunks]*/, unsigned long long *resultSize /*[nq]*/) {
    const uint64_t nCand = *nCandPtr;
    const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nCand) return;
    kmerMergeHeadsThread(j, ckeys, cvals, nCand, tbits, shift, chunks, ecCount, scr, kept, score, best,
                         [&](uint32_t q, uint32_t c, uint32_t n) { atomicAdd(&roundCount[(size_t) q * kMaxChunks + c], n); },
                         [&](uint32_t q, uint32_t n) { atomicAdd(&resultSize[q], (unsigned long long) n); });
}

// findDuplicates cut short (CacheFriendlyOperations.cpp:188-283): the reference walks its bins (id & (B - 1)) in order and RETURNS at the first bin whose
// candidates would not fit behind what the earlier bins handed on -- doubleElementCount + elementCount >= outputSize, outputSize = foundDiagonalsSize minus
// the elements the earlier rounds left (QueryMatcher.cpp:311-346) -- so a chunk loses the candidates of that bin and of every later one.  The host decides
// per (query, chunk) which bin that is (fsgpu_kmer.hip: replayOutputTruncation) from the two counts per (query, chunk, bin) this kernel collects for the
// queries its caller flagged: candidates, and candidates that survive the collapse of equal consecutive diagonals (= elements handed on).
__global__ __launch_bounds__(256) void k_kmer_trunc_hist(const uint32_t *ckeys, const uint64_t *cvals, const uint8_t *kept, const uint32_t *nCandPtr, int tbits, uint32_t B,
                                                         const int32_t *qSlot /*[nq]: slot of a flagged query, -1 otherwise*/, uint32_t *hist /*[slot][kMaxChunks][B][2]*/) {
    const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= *nCandPtr) return;
    const uint32_t key = ckeys[j];
    const int32_t slot = qSlot[key >> tbits];
    if (slot < 0) return;
    const uint32_t bin = (key & ((1u << tbits) - 1u)) & (B - 1u);
    uint32_t *h = hist + (((size_t) slot * kMaxChunks + hitChunk(cvals[j])) * B + bin) * 2;
    atomicAdd(&h[0], 1u);
    if (kept[j]) atomicAdd(&h[1], 1u);
}
// kept[j] = the scoring pass's flag, cleared for the candidates of bins at or beyond their (query, chunk)'s truncation bin (0xFFFFFFFF = none)
__global__ __launch_bounds__(256) void k_kmer_apply_trunc(const uint32_t *ckeys, const uint64_t *cvals, const uint8_t *kept0, const uint32_t *nCandPtr, int tbits, uint32_t B,
                                                          const uint32_t *trunc /*[nq][kMaxChunks]*/, uint8_t *kept) {
    const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= *nCandPtr) return;
    const uint32_t key = ckeys[j];
    const uint32_t bin = (key & ((1u << tbits) - 1u)) & (B - 1u);
    kept[j] = kept0[j] && bin < trunc[(size_t) (key >> tbits) * kMaxChunks + hitChunk(cvals[j])] ? 1 : 0;
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
struct KmerOut { uint32_t id; uint32_t count /* 8-bit score | query << 8 */; uint32_t diag; int32_t score; uint64_t g; };
// elements at or above their query's cut go to ONE output array shared by the batch (a place per wave-group from a single counter; the
// host tail sorts by query anyway), per-query counts alongside.
// A cut of 0 (--min-ungapped-score 0 and fewer than max-seqs scored targets; diagonal-score mode) also takes the elements keepMaxScoreElementOnly
// hands on with score 0 (CacheFriendlyOperations.cpp:112-148: after a target's best element every later zero-score element of it matches the zeroed
// byte; a target whose best is 0 keeps all of them): the head walks its final list (KmerBest.pad, scrA / scrB of k_kmer_walk) once more and emits
// one element per match, each with its own diagonal and arrival position.
// Slots come from ONE counter of the batch: a workgroup of 1024 threads sums its elements in LDS and takes its range with a single global atomic
// (until round 5 every wave with an element went to that counter and to its query's: ~130 k same-address atomics per batch, 0.45 ms for a pass that
// reads 170 MB); the per-query counts of the workgroup's first query are collected in LDS too.
__global__ __launch_bounds__(1024) void k_kmer_out(const uint32_t *ckeys, const uint64_t *cvals, const int32_t *score, const KmerBest *best, const uint32_t *nCandPtr, int tbits,
                                                   const uint32_t *thr, uint32_t outCap, uint32_t *outCount /*[nq]*/, uint32_t *outTotal, KmerOut *out,
                                                   const uint64_t *scrA, const uint64_t *scrB) {
    __shared__ uint32_t blockTotal, blockBase, q0s, q0Count;
    const uint64_t s = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nCand = *nCandPtr;
    if (threadIdx.x == 0) { blockTotal = 0; q0Count = 0; q0s = (uint64_t) blockIdx.x * blockDim.x < nCand ? ckeys[(uint64_t) blockIdx.x * blockDim.x] >> tbits : 0; }
    __syncthreads();
    uint32_t cnt = 0;
    uint32_t qi = 0;
    KmerBest b{};
    if (s < nCand) {
        b = best[s];
        if (b.nElems != 0xFFFFFFFFu && b.nElems != 0) {
            qi = ckeys[s] >> tbits;
            const uint32_t t = thr[qi];
            cnt = t == 0 ? b.nElems : (b.count >= t ? 1u : 0u);
        }
    }
    const bool take = cnt != 0;
    const int lane = (int) (threadIdx.x & 63);
    const unsigned long long all = __ballot(take);
    uint32_t incl = cnt, waveBase = 0;                         // inclusive wave scan of the element counts (a ballot count when nobody has more than one)
    if (all) {
        if (__ballot(cnt > 1) == 0ull) incl = (uint32_t) __popcll(all & ((2ull << lane) - 1ull));
        else for (int d = 1; d < 64; d <<= 1) { const uint32_t v = (uint32_t) __shfl_up((int) incl, d); if (lane >= d) incl += v; }
        const uint32_t waveTotal = (uint32_t) __shfl((int) incl, 63);
        if (lane == 0) waveBase = atomicAdd(&blockTotal, waveTotal);
        waveBase = (uint32_t) __shfl((int) waveBase, 0);
        // one atomic per (wave, query): the workgroup's first query into LDS, the others (a workgroup that runs into the next query) to memory
        const uint32_t q0 = q0s;
        unsigned long long pending = all;
        while (pending) {
            const int leader = __ffsll((long long) pending) - 1;
            const uint32_t q = (uint32_t) __shfl((int) qi, leader);
            const unsigned long long grp = __ballot(take && qi == q);
            uint32_t sum = take && qi == q ? cnt : 0u;
            for (int d = 32; d >= 1; d >>= 1) sum += (uint32_t) __shfl_xor((int) sum, d);
            if (lane == leader) { if (q == q0) atomicAdd(&q0Count, sum); else atomicAdd(&outCount[q], sum); }
            pending &= ~grp;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        blockBase = blockTotal ? atomicAdd(outTotal, blockTotal) : 0u;
        if (q0Count) atomicAdd(&outCount[q0s], q0Count);
    }
    __syncthreads();
    const uint32_t slot = blockBase + waveBase + incl - cnt;
    if (!take) return;
    const uint32_t id = ckeys[s] & ((1u << tbits) - 1u);
    if (cnt == 1 && b.count != 0) {
        if (slot >= outCap) return;
        const uint64_t v = cvals[b.cand];
        KmerOut o;
        o.id = id; o.count = b.count | (qi << 8); o.diag = hitDiag(v); o.score = score[b.cand]; o.g = hitG(v);
        out[slot] = o;
        return;
    }
    // cut 0: the target's final list once more, as keepMaxScoreElementOnly reads it
    const uint64_t *L = ((b.pad & 0x80000000u) ? scrB : scrA) + s;
    const uint32_t nL = b.pad & 0x7fffffffu;
    uint32_t arr = b.count, k = 0;
    for (uint32_t n = 0; n < nL && k < cnt; n++) {
        const uint64_t e = L[n];
        if (arr != ((uint32_t) e & 0xffu)) continue;
        arr = 0;
        const uint32_t c = (uint32_t) (e >> 8);
        if (slot + k < outCap) {
            const uint64_t v = cvals[c];
            KmerOut o;
            o.id = id; o.count = ((uint32_t) e & 0xffu) | (qi << 8); o.diag = hitDiag(v); o.score = score[c]; o.g = hitG(v);
            out[slot + k] = o;
        }
        k++;
    }
}

} // namespace fs
