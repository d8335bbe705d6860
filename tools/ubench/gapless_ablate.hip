// gapless_ablate.hip -- where does k_gapless spend its cycles?  (tools, not product)
// Runs the product kernel on a synthetic uniform DB next to variants with one resource removed:
//   MODE 1: no LDS profile reads (operands stay in registers)      -> VALU-only time
//   MODE 2: LDS reads only (no DP arithmetic)                       -> LDS-only time
//   MODE 3: no lane hand-off (dpp/cndmask/perm dropped)
//   MODE 4: running maximum as a tree (no dependent max3 chain)
//   MODE 5: like 0 but 256-thread workgroups
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../foldseek_amd/csrc -o gapless_ablate gapless_ablate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "k_gapless.hpp"

using namespace fs;
struct AblArgs : GaplessArgs { const uint32_t *order; uint32_t nStripes; };   // variants: one whole stripe per queue entry
static constexpr int oldRowBytes(int R) { return (R / 4) * 256; }   // row-major image of the first kernel version

template <int R, int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_var(AblArgs a) {
    constexpr int ROWB = oldRowBytes(R);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int L = a.L;
        constexpr int nDw = (kAlphabet + 1) * (R / 4) * 2 * 8 * 4;
        for (int idx = threadIdx.x; idx < nDw; idx += blockDim.x) {
            int w = idx & 3, g = (idx >> 2) & 7, copy = (idx >> 5) & 1;
            int k = (idx >> 6) % (R / 4), row = (idx >> 6) / (R / 4);
            int r = 4 * k + w;
            int qlo = g * 2 * R + r, qhi = qlo + R;
            uint32_t v;
            if (row == kDeadCode) v = kDead2;
            else {
                int lo = qlo < L ? (int) a.pssm[row * L + qlo] : 0;
                int hi = qhi < L ? (int) a.pssm[row * L + qhi] : 0;
                v = f16ScaledBits(lo) | (f16ScaledBits(hi) << 16);
            }
            *(uint32_t *) (smem + row * ROWB + k * 256 + copy * 128 + g * 16 + w * 4) = v;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = lane >> 3, g = lane & 7;
    const uint32_t laneOff = (uint32_t) (((j >> 1) & 1) * 128 + g * 16);
    const uint32_t sel = 0x01000706u;
    for (;;) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(a.queue, 1u);
        w = __builtin_amdgcn_readfirstlane(w);
        if (w >= a.nStripes) break;
        const uint32_t stripe = a.order[w];
        const uint32_t len16 = a.stripeLen[stripe];
        const uint64_t soff = a.stripeOff[stripe];
        const uint4 *src = a.scan + soff + j;
        uint32_t S[R], P[R];
        uint32_t M = 0, M2 = 0;
#pragma unroll
        for (int r = 0; r < R; r++) { S[r] = 0; P[r] = (uint32_t) (lane * 7 + r) & 0x03ff03ffu; }
        uint4 nxt = src[0];
        for (uint32_t c = 0; c < len16; c++) {
            const uint4 cur = nxt;
            if (c + 1 < len16) nxt = src[(size_t) (c + 1) * 8];
            const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const uint32_t code = (words[b >> 2] >> ((b & 3) * 8)) & 0xffu;
                const unsigned char *rowp = smem + code * ROWB + laneOff;
                if constexpr (MODE != 1) {
#pragma unroll
                    for (int k = 0; k < R / 4; k++) {
                        const uint4 v = *(const uint4 *) (rowp + k * 256);
                        P[4 * k + 0] = v.x; P[4 * k + 1] = v.y; P[4 * k + 2] = v.z; P[4 * k + 3] = v.w;
                    }
                } else {
                    asm volatile("" : "+v"(P[0]) : "v"(code));
                }
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int r = 0; r < R; r++) asm volatile("" ::"v"(P[r]));
                    continue;
                }
                uint32_t in;
                if constexpr (MODE == 3) {
                    in = S[R - 1];
                } else {
                    uint32_t prev = __builtin_amdgcn_update_dpp(0u, S[R - 1], 0x111, 0xf, 0xf, false);
                    prev = (g == 0) ? 0u : prev;
                    in = __builtin_amdgcn_perm(prev, S[R - 1], sel);
                }
#pragma unroll
                for (int r = R - 1; r >= 1; r--) S[r] = pk_addc_f16(S[r - 1], P[r]);
                S[0] = pk_addc_f16(in, P[0]);
                if constexpr (MODE == 4) {
#pragma unroll
                    for (int r = 0; r < R; r += 4) {
                        M = pk_max3_f16(M, S[r], S[r + 1]);
                        M2 = pk_max3_f16(M2, S[r + 2], S[r + 3]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; r += 2) M = pk_max3_f16(M, S[r], S[r + 1]);
                }
            }
        }
        if constexpr (MODE == 4) M = pk_max3_f16(M, M2, M2);
        int m = max((int) (M & 0xffff), (int) (M >> 16));
        m = max(m, __shfl_xor(m, 1));
        m = max(m, __shfl_xor(m, 2));
        m = max(m, __shfl_xor(m, 4));
        const uint32_t tid = stripe * kStripeTargets + j;
        if (g == 0 && tid < a.nTargets) {
            int sc = (int) (__half2float(__ushort_as_half((unsigned short) m)) * 2048.0f + 0.5f);
            sc = sc < a.cap ? sc : a.cap;
            a.scores[tid] = (uint8_t) sc;
        }
    }
}


// ---- candidate kernel: per-lane perm selector (no cndmask, no dpp old value), interleaved two-accumulator maximum,
//      optional 16-bit pre-scaled row offsets (U16) so that the LDS address is one SDWA add ----
template <int R, bool U16, int BLOCK, int ILV>
__global__ __launch_bounds__(BLOCK) void k_v2(AblArgs a, const uint4 *scan16) {
    constexpr int ROWB = U16 ? 2048 : oldRowBytes(R);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int L = a.L;
        constexpr int nDw = (kAlphabet + 1) * (R / 4) * 2 * 8 * 4;
        for (int idx = threadIdx.x; idx < nDw; idx += blockDim.x) {
            int w = idx & 3, g = (idx >> 2) & 7, copy = (idx >> 5) & 1;
            int k = (idx >> 6) % (R / 4), row = (idx >> 6) / (R / 4);
            int r = 4 * k + w;
            int qlo = g * 2 * R + r, qhi = qlo + R;
            uint32_t v;
            if (row == kDeadCode) v = kDead2;
            else {
                int lo = qlo < L ? (int) a.pssm[row * L + qlo] : 0;
                int hi = qhi < L ? (int) a.pssm[row * L + qhi] : 0;
                v = f16ScaledBits(lo) | (f16ScaledBits(hi) << 16);
            }
            *(uint32_t *) (smem + row * ROWB + k * 256 + copy * 128 + g * 16 + w * 4) = v;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = lane >> 3, g = lane & 7;
    const uint32_t laneOff = (uint32_t) (((j >> 1) & 1) * 128 + g * 16);
    // lane g == 0: low half <- constant zero (selector byte 0x0c), else <- hi half of the previous lane
    const uint32_t sel = (g == 0) ? 0x01000c0cu : 0x01000706u;
    const uint32_t ldsBase = (uint32_t) (uintptr_t) smem + laneOff;   // LDS byte address
    for (;;) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(a.queue, 1u);
        w = __builtin_amdgcn_readfirstlane(w);
        if (w >= a.nStripes) break;
        const uint32_t stripe = a.order[w];
        const uint32_t len16 = a.stripeLen[stripe];
        const uint64_t soff = a.stripeOff[stripe];
        const uint4 *src = U16 ? scan16 + soff * 2 + j * 2 : a.scan + soff + j;
        uint32_t S[R];
        uint32_t M = 0, M2 = 0;
#pragma unroll
        for (int r = 0; r < R; r++) S[r] = 0;
        uint4 nxt = src[0], nxt2 = U16 ? src[1] : make_uint4(0, 0, 0, 0);
        for (uint32_t c = 0; c < len16; c++) {
            const uint4 cur = nxt, cur2 = nxt2;
            if (c + 1 < len16) {
                if (U16) { nxt = src[(size_t) (c + 1) * 16]; nxt2 = src[(size_t) (c + 1) * 16 + 1]; }
                else nxt = src[(size_t) (c + 1) * 8];
            }
            const uint32_t words[8] = {cur.x, cur.y, cur.z, cur.w, cur2.x, cur2.y, cur2.z, cur2.w};
#pragma unroll
            for (int b = 0; b < 16; b++) {
                uint32_t P[R];
                if constexpr (U16) {
                    uint32_t addr;
                    if (b & 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(addr) : "v"(ldsBase), "v"(words[b >> 1]));
                    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(addr) : "v"(ldsBase), "v"(words[b >> 1]));
                    const unsigned char __attribute__((address_space(3))) *rowp = (const unsigned char __attribute__((address_space(3))) *) (uintptr_t) addr;
#pragma unroll
                    for (int k = 0; k < R / 4; k++) {
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 v = *(const u32x4 __attribute__((address_space(3))) *) (rowp + k * 256);
                        P[4 * k + 0] = v.x; P[4 * k + 1] = v.y; P[4 * k + 2] = v.z; P[4 * k + 3] = v.w;
                    }
                } else {
                    const uint32_t code = (words[b >> 2] >> ((b & 3) * 8)) & 0xffu;
                    const unsigned char *rowp = smem + code * ROWB + laneOff;
#pragma unroll
                    for (int k = 0; k < R / 4; k++) {
                        const uint4 v = *(const uint4 *) (rowp + k * 256);
                        P[4 * k + 0] = v.x; P[4 * k + 1] = v.y; P[4 * k + 2] = v.z; P[4 * k + 3] = v.w;
                    }
                }
                const uint32_t prev = __builtin_amdgcn_mov_dpp(S[R - 1], 0x111, 0xf, 0xf, true);
                const uint32_t in = __builtin_amdgcn_perm(prev, S[R - 1], sel);
                if constexpr (ILV == 0) {
#pragma unroll
                    for (int r = R - 1; r >= 1; r--) S[r] = pk_addc_f16(S[r - 1], P[r]);
                    S[0] = pk_addc_f16(in, P[0]);
#pragma unroll
                    for (int r = 0; r < R; r += 4) {
                        M = pk_max3_f16(M, S[r], S[r + 1]);
                        M2 = pk_max3_f16(M2, S[r + 2], S[r + 3]);
                    }
                } else {
                    // adds in descending register order; the maximum of a register pair is issued one pair late
#pragma unroll
                    for (int r = R - 1; r >= 1; r -= 2) {
                        S[r] = pk_addc_f16(S[r - 1], P[r]);
                        S[r - 1] = pk_addc_f16(r >= 2 ? S[r - 2] : in, P[r - 1]);
                        if (r + 2 < R) {
                            if (((r + 1) >> 1) & 1) M = pk_max3_f16(M, S[r + 1], S[r + 2]);
                            else M2 = pk_max3_f16(M2, S[r + 1], S[r + 2]);
                        }
                        asm volatile("" ::: "memory");
                    }
                    M2 = pk_max3_f16(M2, S[0], S[1]);
                }
            }
        }
        M = pk_max3_f16(M, M2, M2);
        int m = max((int) (M & 0xffff), (int) (M >> 16));
        m = max(m, __shfl_xor(m, 1));
        m = max(m, __shfl_xor(m, 2));
        m = max(m, __shfl_xor(m, 4));
        const uint32_t tid = stripe * kStripeTargets + j;
        if (g == 0 && tid < a.nTargets) {
            int sc = (int) (__half2float(__ushort_as_half((unsigned short) m)) * 2048.0f + 0.5f);
            sc = sc < a.cap ? sc : a.cap;
            a.scores[tid] = (uint8_t) sc;
        }
    }
}


// ---- candidate 3: v2 + explicit software prefetch of the next column's profile row (double-buffered P) ----
template <int R, bool U16, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_v3(AblArgs a, const uint4 *scan16) {
    constexpr int ROWB = U16 ? 2048 : oldRowBytes(R);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int L = a.L;
        constexpr int nDw = (kAlphabet + 1) * (R / 4) * 2 * 8 * 4;
        for (int idx = threadIdx.x; idx < nDw; idx += blockDim.x) {
            int w = idx & 3, g = (idx >> 2) & 7, copy = (idx >> 5) & 1;
            int k = (idx >> 6) % (R / 4), row = (idx >> 6) / (R / 4);
            int r = 4 * k + w;
            int qlo = g * 2 * R + r, qhi = qlo + R;
            uint32_t v;
            if (row == kDeadCode) v = kDead2;
            else {
                int lo = qlo < L ? (int) a.pssm[row * L + qlo] : 0;
                int hi = qhi < L ? (int) a.pssm[row * L + qhi] : 0;
                v = f16ScaledBits(lo) | (f16ScaledBits(hi) << 16);
            }
            *(uint32_t *) (smem + row * ROWB + k * 256 + copy * 128 + g * 16 + w * 4) = v;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = lane >> 3, g = lane & 7;
    const uint32_t laneOff = (uint32_t) (((j >> 1) & 1) * 128 + g * 16);
    const uint32_t sel = (g == 0) ? 0x01000c0cu : 0x01000706u;
    const uint32_t ldsBase = (uint32_t) (uintptr_t) smem + laneOff;
    auto rowAddr = [&](const uint32_t (&words)[8], int b) -> uint32_t {
        uint32_t addr;
        if constexpr (U16) {
            if (b & 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(addr) : "v"(ldsBase), "v"(words[b >> 1]));
            else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(addr) : "v"(ldsBase), "v"(words[b >> 1]));
        } else {
            const uint32_t code = (words[b >> 2] >> ((b & 3) * 8)) & 0xffu;
            addr = code * ROWB + ldsBase;
        }
        return addr;
    };
    auto loadRow = [&](uint32_t (&P)[R], uint32_t addr) {
        const unsigned char __attribute__((address_space(3))) *rowp = (const unsigned char __attribute__((address_space(3))) *) (uintptr_t) addr;
#pragma unroll
        for (int k = 0; k < R / 4; k++) {
            const u32x4 v = *(const u32x4 __attribute__((address_space(3))) *) (rowp + k * 256);
            P[4 * k + 0] = v.x; P[4 * k + 1] = v.y; P[4 * k + 2] = v.z; P[4 * k + 3] = v.w;
        }
    };
    for (;;) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(a.queue, 1u);
        w = __builtin_amdgcn_readfirstlane(w);
        if (w >= a.nStripes) break;
        const uint32_t stripe = a.order[w];
        const uint32_t len16 = a.stripeLen[stripe];
        const uint64_t soff = a.stripeOff[stripe];
        const uint4 *src = U16 ? scan16 + soff * 2 + j * 2 : a.scan + soff + j;
        uint32_t S[R], PA[R], PB[R];
        uint32_t M = 0, M2 = 0;
#pragma unroll
        for (int r = 0; r < R; r++) S[r] = 0;
        uint4 nxt = src[0], nxt2 = U16 ? src[1] : make_uint4(0, 0, 0, 0);
        {
            const uint32_t w0[8] = {nxt.x, nxt.y, nxt.z, nxt.w, nxt2.x, nxt2.y, nxt2.z, nxt2.w};
            loadRow(PA, rowAddr(w0, 0));
        }
        for (uint32_t c = 0; c < len16; c++) {
            const uint4 cur = nxt, cur2 = nxt2;
            if (c + 1 < len16) {
                if (U16) { nxt = src[(size_t) (c + 1) * 16]; nxt2 = src[(size_t) (c + 1) * 16 + 1]; }
                else nxt = src[(size_t) (c + 1) * 8];
            }
            const uint32_t words[8] = {cur.x, cur.y, cur.z, cur.w, cur2.x, cur2.y, cur2.z, cur2.w};
#pragma unroll
            for (int b = 0; b < 16; b++) {
                uint32_t (&P)[R] = (b & 1) ? PB : PA;
                uint32_t (&Pn)[R] = (b & 1) ? PA : PB;
                if (b < 15) {
                    loadRow(Pn, rowAddr(words, b + 1));
                } else {
                    const uint32_t wn[8] = {nxt.x, nxt.y, nxt.z, nxt.w, nxt2.x, nxt2.y, nxt2.z, nxt2.w};
                    loadRow(Pn, rowAddr(wn, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t prev = __builtin_amdgcn_mov_dpp(S[R - 1], 0x111, 0xf, 0xf, true);
                const uint32_t in = __builtin_amdgcn_perm(prev, S[R - 1], sel);
#pragma unroll
                for (int r = R - 1; r >= 1; r--) S[r] = pk_addc_f16(S[r - 1], P[r]);
                S[0] = pk_addc_f16(in, P[0]);
#pragma unroll
                for (int r = 0; r < R; r += 4) {
                    M = pk_max3_f16(M, S[r], S[r + 1]);
                    M2 = pk_max3_f16(M2, S[r + 2], S[r + 3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        M = pk_max3_f16(M, M2, M2);
        int m = max((int) (M & 0xffff), (int) (M >> 16));
        m = max(m, __shfl_xor(m, 1));
        m = max(m, __shfl_xor(m, 2));
        m = max(m, __shfl_xor(m, 4));
        const uint32_t tid = stripe * kStripeTargets + j;
        if (g == 0 && tid < a.nTargets) {
            int sc = (int) (__half2float(__ushort_as_half((unsigned short) m)) * 2048.0f + 0.5f);
            sc = sc < a.cap ? sc : a.cap;
            a.scores[tid] = (uint8_t) sc;
        }
    }
}


// ---- candidate 4: u8 codes, LDS image chunk-major (chunk k at k*22*256, row stride 256 B) so that the row address
//      (code << 8) | laneOff is ONE v_perm_b32; per-lane hand-off selector; two-accumulator maximum ----
template <int R, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_v4(AblArgs a, const uint4 *) {
    constexpr int CHB = (kAlphabet + 1) * 256;       // bytes per 4-register chunk plane
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int L = a.L;
        constexpr int nDw = (kAlphabet + 1) * (R / 4) * 2 * 8 * 4;
        for (int idx = threadIdx.x; idx < nDw; idx += blockDim.x) {
            int w = idx & 3, g = (idx >> 2) & 7, copy = (idx >> 5) & 1;
            int k = (idx >> 6) % (R / 4), row = (idx >> 6) / (R / 4);
            int r = 4 * k + w;
            int qlo = g * 2 * R + r, qhi = qlo + R;
            uint32_t v;
            if (row == kDeadCode) v = kDead2;
            else {
                int lo = qlo < L ? (int) a.pssm[row * L + qlo] : 0;
                int hi = qhi < L ? (int) a.pssm[row * L + qhi] : 0;
                v = f16ScaledBits(lo) | (f16ScaledBits(hi) << 16);
            }
            *(uint32_t *) (smem + k * CHB + row * 256 + copy * 128 + g * 16 + w * 4) = v;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = lane >> 3, g = lane & 7;
    const uint32_t laneOff = (uint32_t) (((j >> 1) & 1) * 128 + g * 16);      // < 256: one byte
    const uint32_t sel = (g == 0) ? 0x01000c0cu : 0x01000706u;
    for (;;) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(a.queue, 1u);
        w = __builtin_amdgcn_readfirstlane(w);
        if (w >= a.nStripes) break;
        const uint32_t stripe = a.order[w];
        const uint32_t len16 = a.stripeLen[stripe];
        const uint64_t soff = a.stripeOff[stripe];
        const uint4 *src = a.scan + soff + j;
        uint32_t S[R];
        uint32_t M = 0, M2 = 0;
#pragma unroll
        for (int r = 0; r < R; r++) S[r] = 0;
        uint4 nxt = src[0];
        for (uint32_t c = 0; c < len16; c++) {
            const uint4 cur = nxt;
            if (c + 1 < len16) nxt = src[(size_t) (c + 1) * 8];
            const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int b = 0; b < 16; b++) {
                // address bytes {0, 0, code, laneOff}: S0 = word (selector bytes 4..7), S1 = laneOff (bytes 0..3)
                const uint32_t addr = __builtin_amdgcn_perm(words[b >> 2], laneOff, 0x0c0c0000u | ((4u + (b & 3)) << 8));
                const unsigned char __attribute__((address_space(3))) *rowp = (const unsigned char __attribute__((address_space(3))) *) (uintptr_t) addr;
                uint32_t P[R];
#pragma unroll
                for (int k = 0; k < R / 4; k++) {
                    const u32x4 v = *(const u32x4 __attribute__((address_space(3))) *) (rowp + k * CHB);
                    P[4 * k + 0] = v.x; P[4 * k + 1] = v.y; P[4 * k + 2] = v.z; P[4 * k + 3] = v.w;
                }
                const uint32_t prev = __builtin_amdgcn_mov_dpp(S[R - 1], 0x111, 0xf, 0xf, true);
                const uint32_t in = __builtin_amdgcn_perm(prev, S[R - 1], sel);
#pragma unroll
                for (int r = R - 1; r >= 1; r--) S[r] = pk_addc_f16(S[r - 1], P[r]);
                S[0] = pk_addc_f16(in, P[0]);
#pragma unroll
                for (int r = 0; r < R; r += 4) {
                    M = pk_max3_f16(M, S[r], S[r + 1]);
                    M2 = pk_max3_f16(M2, S[r + 2], S[r + 3]);
                }
            }
        }
        M = pk_max3_f16(M, M2, M2);
        int m = max((int) (M & 0xffff), (int) (M >> 16));
        m = max(m, __shfl_xor(m, 1));
        m = max(m, __shfl_xor(m, 2));
        m = max(m, __shfl_xor(m, 4));
        const uint32_t tid = stripe * kStripeTargets + j;
        if (g == 0 && tid < a.nTargets) {
            int sc = (int) (__half2float(__ushort_as_half((unsigned short) m)) * 2048.0f + 0.5f);
            sc = sc < a.cap ? sc : a.cap;
            a.scores[tid] = (uint8_t) sc;
        }
    }
}

// ---- candidate 4: u8 codes, LDS image chunk-major (chunk k at k*22*256, row stride 256 B) so that the row address
//      (code << 8) | laneOff is ONE v_perm_b32; per-lane hand-off selector; two-accumulator maximum ----
template <int R, int BLOCK>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_v4w(AblArgs a, const uint4 *) {
    constexpr int CHB = (kAlphabet + 1) * 256;       // bytes per 4-register chunk plane
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int L = a.L;
        constexpr int nDw = (kAlphabet + 1) * (R / 4) * 2 * 8 * 4;
        for (int idx = threadIdx.x; idx < nDw; idx += blockDim.x) {
            int w = idx & 3, g = (idx >> 2) & 7, copy = (idx >> 5) & 1;
            int k = (idx >> 6) % (R / 4), row = (idx >> 6) / (R / 4);
            int r = 4 * k + w;
            int qlo = g * 2 * R + r, qhi = qlo + R;
            uint32_t v;
            if (row == kDeadCode) v = kDead2;
            else {
                int lo = qlo < L ? (int) a.pssm[row * L + qlo] : 0;
                int hi = qhi < L ? (int) a.pssm[row * L + qhi] : 0;
                v = f16ScaledBits(lo) | (f16ScaledBits(hi) << 16);
            }
            *(uint32_t *) (smem + k * CHB + row * 256 + copy * 128 + g * 16 + w * 4) = v;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = lane >> 3, g = lane & 7;
    const uint32_t laneOff = (uint32_t) (((j >> 1) & 1) * 128 + g * 16);      // < 256: one byte
    const uint32_t sel = (g == 0) ? 0x01000c0cu : 0x01000706u;
    for (;;) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(a.queue, 1u);
        w = __builtin_amdgcn_readfirstlane(w);
        if (w >= a.nStripes) break;
        const uint32_t stripe = a.order[w];
        const uint32_t len16 = a.stripeLen[stripe];
        const uint64_t soff = a.stripeOff[stripe];
        const uint4 *src = a.scan + soff + j;
        uint32_t S[R];
        uint32_t M = 0, M2 = 0;
#pragma unroll
        for (int r = 0; r < R; r++) S[r] = 0;
        uint4 nxt = src[0];
        for (uint32_t c = 0; c < len16; c++) {
            const uint4 cur = nxt;
            if (c + 1 < len16) nxt = src[(size_t) (c + 1) * 8];
            const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int b = 0; b < 16; b++) {
                // address bytes {0, 0, code, laneOff}: S0 = word (selector bytes 4..7), S1 = laneOff (bytes 0..3)
                const uint32_t addr = __builtin_amdgcn_perm(words[b >> 2], laneOff, 0x0c0c0000u | ((4u + (b & 3)) << 8));
                const unsigned char __attribute__((address_space(3))) *rowp = (const unsigned char __attribute__((address_space(3))) *) (uintptr_t) addr;
                uint32_t P[R];
#pragma unroll
                for (int k = 0; k < R / 4; k++) {
                    const u32x4 v = *(const u32x4 __attribute__((address_space(3))) *) (rowp + k * CHB);
                    P[4 * k + 0] = v.x; P[4 * k + 1] = v.y; P[4 * k + 2] = v.z; P[4 * k + 3] = v.w;
                }
                const uint32_t prev = __builtin_amdgcn_mov_dpp(S[R - 1], 0x111, 0xf, 0xf, true);
                const uint32_t in = __builtin_amdgcn_perm(prev, S[R - 1], sel);
#pragma unroll
                for (int r = R - 1; r >= 1; r--) S[r] = pk_addc_f16(S[r - 1], P[r]);
                S[0] = pk_addc_f16(in, P[0]);
#pragma unroll
                for (int r = 0; r < R; r += 4) {
                    M = pk_max3_f16(M, S[r], S[r + 1]);
                    M2 = pk_max3_f16(M2, S[r + 2], S[r + 3]);
                }
            }
        }
        M = pk_max3_f16(M, M2, M2);
        int m = max((int) (M & 0xffff), (int) (M >> 16));
        m = max(m, __shfl_xor(m, 1));
        m = max(m, __shfl_xor(m, 2));
        m = max(m, __shfl_xor(m, 4));
        const uint32_t tid = stripe * kStripeTargets + j;
        if (g == 0 && tid < a.nTargets) {
            int sc = (int) (__half2float(__ushort_as_half((unsigned short) m)) * 2048.0f + 0.5f);
            sc = sc < a.cap ? sc : a.cap;
            a.scores[tid] = (uint8_t) sc;
        }
    }
}


// ---- candidate 5: stripe scheduling.  SCHED 0: static round robin (no atomics); SCHED 1: atomic ticket taken one
//      stripe ahead, next stripe's metadata + first chunk loaded while the current stripe computes ----
template <int R, int SCHED>
__global__ __launch_bounds__(256) void k_v5(AblArgs a, const uint4 *) {
    constexpr int CHB = (kAlphabet + 1) * 256;
    constexpr int NCH = (R + 3) / 4;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int L = a.L;
        constexpr int nDw = (kAlphabet + 1) * NCH * 2 * 8 * 4;
        for (int idx = threadIdx.x; idx < nDw; idx += blockDim.x) {
            int w = idx & 3, g = (idx >> 2) & 7, copy = (idx >> 5) & 1;
            int k = (idx >> 6) % NCH, row = (idx >> 6) / NCH;
            int r = 4 * k + w;
            int qlo = g * 2 * R + r, qhi = qlo + R;
            uint32_t v;
            if (row == kDeadCode || r >= R) v = kDead2;
            else if (SCHED == 3) v = (uint32_t) ((qlo * 7 + row) & 7) * 0x00010001u;       // no global reads in the image build
            else {
                int lo = qlo < L ? (int) a.pssm[row * L + qlo] : 0;
                int hi = qhi < L ? (int) a.pssm[row * L + qhi] : 0;
                v = f16ScaledBits(lo) | (f16ScaledBits(hi) << 16);
            }
            *(uint32_t *) (smem + k * CHB + row * 256 + copy * 128 + g * 16 + w * 4) = v;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = lane >> 3, g = lane & 7;
    const uint32_t laneOff = (uint32_t) (((j >> 1) & 1) * 128 + g * 16);
    const uint32_t sel = (g == 0) ? 0x01000c0cu : 0x01000706u;
    const uint32_t nWaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t waveId = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    auto ticket = [&]() -> uint32_t {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(a.queue, 1u);
        return __builtin_amdgcn_readfirstlane(w);
    };
    uint32_t w = SCHED == 1 ? ticket() : waveId;       // SCHED 2: first stripe static, then tickets (offset by nWaves)
    // metadata of the current stripe
    uint32_t stripe = 0, len16 = 0; uint64_t soff = 0; uint4 first = make_uint4(0, 0, 0, 0);
    auto fetchMeta = [&](uint32_t ww, uint32_t &st, uint32_t &ln, uint64_t &so, uint4 &f) {
        if (ww < a.nStripes) {
            st = a.order[ww]; ln = a.stripeLen[st]; so = a.stripeOff[st];
            f = (a.scan + so + j)[0];
        }
    };
    fetchMeta(w, stripe, len16, soff, first);
    while (w < a.nStripes) {
        // next stripe: ticket + metadata in flight while this one computes
        uint32_t wN = SCHED == 0 ? w + nWaves : (SCHED == 1 ? ticket() : nWaves + ticket());
        uint32_t stripeN = 0, len16N = 0; uint64_t soffN = 0; uint4 firstN = make_uint4(0, 0, 0, 0);
        fetchMeta(wN, stripeN, len16N, soffN, firstN);
        const uint4 *src = a.scan + soff + j;
        uint32_t S[R];
        uint32_t M = 0, M2 = 0;
#pragma unroll
        for (int r = 0; r < R; r++) S[r] = 0;
        uint4 nxt = first;
        for (uint32_t c = 0; c < len16; c++) {
            const uint4 cur = nxt;
            if (c + 1 < len16) nxt = src[(size_t) (c + 1) * 8];
            const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const uint32_t addr = __builtin_amdgcn_perm(words[b >> 2], laneOff, 0x0c0c0000u | ((4u + (b & 3)) << 8));
                const unsigned char __attribute__((address_space(3))) *rowp = (const unsigned char __attribute__((address_space(3))) *) (uintptr_t) addr;
                uint32_t P[4 * NCH];
#pragma unroll
                for (int k = 0; k < NCH; k++) {
                    const u32x4 v = *(const u32x4 __attribute__((address_space(3))) *) (rowp + k * CHB);
                    P[4 * k + 0] = v.x; P[4 * k + 1] = v.y; P[4 * k + 2] = v.z; P[4 * k + 3] = v.w;
                }
                const uint32_t prev = __builtin_amdgcn_mov_dpp(S[R - 1], 0x111, 0xf, 0xf, true);
                const uint32_t in = __builtin_amdgcn_perm(prev, S[R - 1], sel);
#pragma unroll
                for (int r = R - 1; r >= 1; r--) S[r] = pk_addc_f16(S[r - 1], P[r]);
                S[0] = pk_addc_f16(in, P[0]);
#pragma unroll
                for (int r = 0; r + 3 < R; r += 4) {
                    M = pk_max3_f16(M, S[r], S[r + 1]);
                    M2 = pk_max3_f16(M2, S[r + 2], S[r + 3]);
                }
            }
        }
        M = pk_max3_f16(M, M2, M2);
        int m = max((int) (M & 0xffff), (int) (M >> 16));
        m = max(m, __shfl_xor(m, 1));
        m = max(m, __shfl_xor(m, 2));
        m = max(m, __shfl_xor(m, 4));
        const uint32_t tid = stripe * kStripeTargets + j;
        if (g == 0 && tid < a.nTargets) {
            int sc = (int) (__half2float(__ushort_as_half((unsigned short) m)) * 2048.0f + 0.5f);
            sc = sc < a.cap ? sc : a.cap;
            a.scores[tid] = (uint8_t) sc;
        }
        w = wN; stripe = stripeN; len16 = len16N; soff = soffN; first = firstN;
    }
}

// ---- candidate 6 (from 5): one 16-byte record per item {stripe, len16, soff}; SCHED 0: ticket at stripe end (2 dependent loads after it), SCHED 1: next ticket taken at stripe START, consumed at its end ----
// candidate-5 text follows:  SCHED 0: static round robin (no atomics); SCHED 1: atomic ticket taken one
//      stripe ahead, next stripe's metadata + first chunk loaded while the current stripe computes ----
template <int R, int SCHED>
__global__ __launch_bounds__(256) void k_v6(AblArgs a, const uint4 *scan16) {
    constexpr int CHB = (kAlphabet + 1) * 256;
    constexpr int NCH = (R + 3) / 4;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int L = a.L;
        constexpr int nDw = (kAlphabet + 1) * NCH * 2 * 8 * 4;
        for (int idx = threadIdx.x; idx < nDw; idx += blockDim.x) {
            int w = idx & 3, g = (idx >> 2) & 7, copy = (idx >> 5) & 1;
            int k = (idx >> 6) % NCH, row = (idx >> 6) / NCH;
            int r = 4 * k + w;
            int qlo = g * 2 * R + r, qhi = qlo + R;
            uint32_t v;
            if (row == kDeadCode || r >= R) v = kDead2;
            else if (SCHED == 3) v = (uint32_t) ((qlo * 7 + row) & 7) * 0x00010001u;       // no global reads in the image build
            else {
                int lo = qlo < L ? (int) a.pssm[row * L + qlo] : 0;
                int hi = qhi < L ? (int) a.pssm[row * L + qhi] : 0;
                v = f16ScaledBits(lo) | (f16ScaledBits(hi) << 16);
            }
            *(uint32_t *) (smem + k * CHB + row * 256 + copy * 128 + g * 16 + w * 4) = v;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = lane >> 3, g = lane & 7;
    const uint32_t laneOff = (uint32_t) (((j >> 1) & 1) * 128 + g * 16);
    const uint32_t sel = (g == 0) ? 0x01000c0cu : 0x01000706u;
    const uint32_t nWaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t waveId = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    auto ticket = [&]() -> uint32_t {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(a.queue, 1u);
        return __builtin_amdgcn_readfirstlane(w);
    };
    const uint4 *recs = (const uint4 *) scan16;         // {stripe, len16, soff lo, soff hi}
    uint32_t w = waveId;                                 // first item static, then tickets offset by nWaves
    while (w < a.nStripes) {
        uint32_t wN = 0;
        if (SCHED == 1) wN = nWaves + ticket();          // in flight while this stripe computes
        const uint4 rec = recs[w];
        const uint32_t stripe = rec.x, len16 = rec.y;
        const uint64_t soff = ((uint64_t) rec.w << 32) | rec.z;
        const uint4 first = (a.scan + soff + j)[0];
        const uint4 *src = a.scan + soff + j;
        uint32_t S[R];
        uint32_t M = 0, M2 = 0;
#pragma unroll
        for (int r = 0; r < R; r++) S[r] = 0;
        uint4 nxt = first;
        for (uint32_t c = 0; c < len16; c++) {
            const uint4 cur = nxt;
            if (c + 1 < len16) nxt = src[(size_t) (c + 1) * 8];
            const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const uint32_t addr = __builtin_amdgcn_perm(words[b >> 2], laneOff, 0x0c0c0000u | ((4u + (b & 3)) << 8));
                const unsigned char __attribute__((address_space(3))) *rowp = (const unsigned char __attribute__((address_space(3))) *) (uintptr_t) addr;
                uint32_t P[4 * NCH];
#pragma unroll
                for (int k = 0; k < NCH; k++) {
                    const u32x4 v = *(const u32x4 __attribute__((address_space(3))) *) (rowp + k * CHB);
                    P[4 * k + 0] = v.x; P[4 * k + 1] = v.y; P[4 * k + 2] = v.z; P[4 * k + 3] = v.w;
                }
                const uint32_t prev = __builtin_amdgcn_mov_dpp(S[R - 1], 0x111, 0xf, 0xf, true);
                const uint32_t in = __builtin_amdgcn_perm(prev, S[R - 1], sel);
#pragma unroll
                for (int r = R - 1; r >= 1; r--) S[r] = pk_addc_f16(S[r - 1], P[r]);
                S[0] = pk_addc_f16(in, P[0]);
#pragma unroll
                for (int r = 0; r + 3 < R; r += 4) {
                    M = pk_max3_f16(M, S[r], S[r + 1]);
                    M2 = pk_max3_f16(M2, S[r + 2], S[r + 3]);
                }
            }
        }
        M = pk_max3_f16(M, M2, M2);
        int m = max((int) (M & 0xffff), (int) (M >> 16));
        m = max(m, __shfl_xor(m, 1));
        m = max(m, __shfl_xor(m, 2));
        m = max(m, __shfl_xor(m, 4));
        const uint32_t tid = stripe * kStripeTargets + j;
        if (g == 0 && tid < a.nTargets) {
            int sc = (int) (__half2float(__ushort_as_half((unsigned short) m)) * 2048.0f + 0.5f);
            sc = sc < a.cap ? sc : a.cap;
            a.scores[tid] = (uint8_t) sc;
        }
        if (SCHED == 0) wN = nWaves + ticket();
        w = wN;
    }
}

template <typename K>
static float runK2(K kern, int block, int lds, int blocks, AblArgs ga, const uint4 *s16, int reps) {
    hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < reps; it++) {
        hipMemsetAsync(ga.queue, 0, 4, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(block), lds, 0, ga, s16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) printf("  (launch error)\n");
    return best;
}

template <typename K, typename A>
static float runK(K kern, int block, int lds, int blocks, A ga, int reps) {
    hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < reps; it++) {
        hipMemsetAsync(ga.queue, 0, 4, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(block), lds, 0, ga);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) printf("  (launch error)\n");
    return best;
}

template <int R>
static void suite(int nStripes, int len16, int L) {
    const size_t chunkU4 = (size_t) nStripes * len16 * 8;
    std::vector<uint8_t> h(chunkU4 * 16);
    srand(1);
    for (auto &x : h) x = rand() % 20;
    std::vector<uint64_t> off(nStripes); std::vector<uint32_t> len(nStripes), ord(nStripes);
    for (int s = 0; s < nStripes; s++) { off[s] = (uint64_t) s * len16 * 8; len[s] = len16; ord[s] = s; }
    std::vector<int8_t> pssm(21 * L);
    for (auto &x : pssm) x = (int8_t) (rand() % 13 - 8);
    AblArgs ga{};
    void *d;
    hipMalloc(&d, h.size()); hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice); ga.scan = (const uint4 *) d;
    hipMalloc(&d, 8 * nStripes); hipMemcpy(d, off.data(), 8 * nStripes, hipMemcpyHostToDevice); ga.stripeOff = (const uint64_t *) d;
    hipMalloc(&d, 4 * nStripes); hipMemcpy(d, len.data(), 4 * nStripes, hipMemcpyHostToDevice); ga.stripeLen = (const uint32_t *) d;
    hipMalloc(&d, 4 * nStripes); hipMemcpy(d, ord.data(), 4 * nStripes, hipMemcpyHostToDevice); ga.order = (const uint32_t *) d;
    hipMalloc(&d, pssm.size()); hipMemcpy(d, pssm.data(), pssm.size(), hipMemcpyHostToDevice); ga.pssm = (const int8_t *) d;
    hipMalloc(&d, nStripes * 8); ga.scores = (uint8_t *) d;
    hipMalloc(&d, 4); ga.queue = (uint32_t *) d;
    {
        std::vector<uint4> items(nStripes);
        for (int s2 = 0; s2 < nStripes; s2++) items[s2] = make_uint4((uint32_t) s2, (uint32_t) len16, (uint32_t) off[s2], (uint32_t) (off[s2] >> 32));
        hipMalloc(&d, 16 * nStripes); hipMemcpy(d, items.data(), 16 * nStripes, hipMemcpyHostToDevice); ga.items = (const uint4 *) d; ga.nItems = nStripes;
    }
    {
        std::vector<uint32_t> ident(nStripes * 8);
        for (int t = 0; t < nStripes * 8; t++) ident[t] = t;
        hipMalloc(&d, 4 * ident.size()); hipMemcpy(d, ident.data(), 4 * ident.size(), hipMemcpyHostToDevice); ga.stripeTargets = (const uint32_t *) d;
    }
    ga.nStripes = nStripes; ga.nTargets = nStripes * 8; ga.L = L; ga.cap = 255; ga.firstTile = ga.lastTile = 1;
    const int lds = gaplessLdsBytes(R);
    const double cells = (double) nStripes * 8 * len16 * 16 * (16.0 * R);
    const double waveCols = (double) nStripes * len16 * 16;
    auto rep = [&](const char *name, float ms) {
        // cycles per wave-column per SIMD: ms * 2.4e6 cycles * 1024 SIMDs / waveCols
        printf("R=%2d %-34s %7.3f ms  %6.2f Tcell/s  %6.1f cyc/wave-column/SIMD\n", R, name, ms, cells / ms * 1e-9, ms * 2.4e6 * 1024 / waveCols);
    };
    runK(k_gapless<R, false>, kGaplessBlock, lds, 768, (GaplessArgs) ga, 40);      // warm-up: let the clocks ramp before anything is timed
    for (int perCU = 2; perCU <= 4; perCU++) {
        char nm[64]; snprintf(nm, sizeof nm, "product kernel, %d WG/CU", perCU);
        rep(nm, runK(k_gapless<R, false>, kGaplessBlock, lds, 256 * perCU, (GaplessArgs) ga, 4));
    }
    rep("variant 0 (same code)", runK(k_var<R, 0, 512>, 512, lds, 512, ga, 4));
    rep("1: no LDS reads", runK(k_var<R, 1, 512>, 512, lds, 512, ga, 4));
    rep("2: LDS reads only", runK(k_var<R, 2, 512>, 512, lds, 512, ga, 4));
    rep("3: no lane hand-off", runK(k_var<R, 3, 512>, 512, lds, 512, ga, 4));
    rep("4: max tree", runK(k_var<R, 4, 512>, 512, lds, 512, ga, 4));
    rep("5: 256-thread WG x4/CU", runK(k_var<R, 0, 256>, 256, lds, 1024, ga, 4));
    rep("5b: 256-thread WG x5/CU", runK(k_var<R, 0, 256>, 256, lds, 1280, ga, 4));
    {
        // 16-bit pre-scaled scan layout: per stripe chunk 8 targets x 32 bytes
        std::vector<uint16_t> h16(chunkU4 * 16);
        for (size_t u = 0; u < chunkU4; u++)            // u = (stripe-chunk)*8 + j ; 16 residues each
            for (int b = 0; b < 16; b++) h16[u * 16 + b] = (uint16_t) (h[u * 16 + b] * 2048);
        void *d16; hipMalloc(&d16, h16.size() * 2); hipMemcpy(d16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice);
        std::vector<uint8_t> ref(nStripes * 8), got(nStripes * 8);
        runK(k_gapless<R, false>, kGaplessBlock, lds, 768, (GaplessArgs) ga, 2);
        hipMemcpy(ref.data(), ga.scores, ref.size(), hipMemcpyDeviceToHost);
        auto chk = [&](const char *nm, float ms) {
            hipMemcpy(got.data(), ga.scores, got.size(), hipMemcpyDeviceToHost);
            size_t bad = 0; for (size_t i = 0; i < ref.size(); i++) bad += ref[i] != got[i];
            char b2[96]; snprintf(b2, sizeof b2, "%s%s", nm, bad ? " MISMATCH" : "");
            rep(b2, ms);
            hipMemset(ga.scores, 0, got.size());
        };
        hipMemset(ga.scores, 0, got.size());
        chk("v2 sel-perm, tree, 512", runK2(k_v2<R, false, 512, 0>, 512, lds, 512, ga, nullptr, 4));
        chk("v2 sel-perm, interleave, 512", runK2(k_v2<R, false, 512, 1>, 512, lds, 512, ga, nullptr, 4));
        chk("v2 sel-perm, tree, 256", runK2(k_v2<R, false, 256, 0>, 256, lds, 1024, ga, nullptr, 4));
        chk("v2 sel-perm, interleave, 256", runK2(k_v2<R, false, 256, 1>, 256, lds, 1024, ga, nullptr, 4));
        const int lds16 = 22 * 2048;
        chk("v2 u16, tree, 512", runK2(k_v2<R, true, 512, 0>, 512, lds16, 512, ga, (const uint4 *) d16, 4));
        chk("v2 u16, interleave, 512", runK2(k_v2<R, true, 512, 1>, 512, lds16, 512, ga, (const uint4 *) d16, 4));
        chk("v5 static round robin, 256 x3", runK2(k_v5<R, 0>, 256, lds, 768, ga, nullptr, 4));
        chk("v5 ticket+meta prefetch, 256 x3", runK2(k_v5<R, 1>, 256, lds, 768, ga, nullptr, 4));
        chk("v5 static first + prefetch, 256 x3", runK2(k_v5<R, 2>, 256, lds, 768, ga, nullptr, 4));
        rep("v5 same, image build without loads", runK2(k_v5<R, 3>, 256, lds, 768, ga, nullptr, 4));
        {
            std::vector<uint4> recs(nStripes);
            for (int s2 = 0; s2 < nStripes; s2++) recs[s2] = make_uint4((uint32_t) s2, (uint32_t) len16, (uint32_t) off[s2], (uint32_t) (off[s2] >> 32));
            void *dr; hipMalloc(&dr, 16 * nStripes); hipMemcpy(dr, recs.data(), 16 * nStripes, hipMemcpyHostToDevice);
            chk("v6 16B records, ticket at end", runK2(k_v6<R, 0>, 256, lds, 768, ga, (const uint4 *) dr, 4));
            chk("v6 16B records, ticket at start", runK2(k_v6<R, 1>, 256, lds, 768, ga, (const uint4 *) dr, 4));
        }
        chk("v4 perm-addr u8, 512 x2", runK2(k_v4<R, 512>, 512, lds, 512, ga, nullptr, 4));
        chk("v4 perm-addr u8, 512 x3", runK2(k_v4<R, 512>, 512, lds, 768, ga, nullptr, 4));
        chk("v4 perm-addr u8, 256 x3", runK2(k_v4<R, 256>, 256, lds, 768, ga, nullptr, 4));
        chk("v4 perm-addr u8, 256 x4", runK2(k_v4<R, 256>, 256, lds, 1024, ga, nullptr, 4));
        chk("v4 perm-addr u8, 256 x5", runK2(k_v4<R, 256>, 256, lds, 1280, ga, nullptr, 4));
        chk("v4w wpe4 u8, 512 x2", runK2(k_v4w<R, 512>, 512, lds, 512, ga, nullptr, 4));
        chk("v4w wpe4 u8, 256 x4", runK2(k_v4w<R, 256>, 256, lds, 1024, ga, nullptr, 4));
        chk("v4 perm-addr u8, 128 x8", runK2(k_v4<R, 128>, 128, lds, 2048, ga, nullptr, 4));
        chk("v3 prefetch u8, 512", runK2(k_v3<R, false, 512>, 512, lds, 512, ga, nullptr, 4));
        chk("v3 prefetch u8, 256", runK2(k_v3<R, false, 256>, 256, lds, 1024, ga, nullptr, 4));
        chk("v3 prefetch u16, 512", runK2(k_v3<R, true, 512>, 512, lds16, 512, ga, (const uint4 *) d16, 4));
        chk("v3 prefetch u16, 512 x3", runK2(k_v3<R, true, 512>, 512, lds16, 768, ga, (const uint4 *) d16, 4));
        chk("v3 prefetch u16, 256", runK2(k_v3<R, true, 256>, 256, lds16, 1024, ga, (const uint4 *) d16, 4));
        chk("v3 prefetch u16, 256 x5", runK2(k_v3<R, true, 256>, 256, lds16, 1280, ga, (const uint4 *) d16, 4));
        chk("v2 u16, tree, 256", runK2(k_v2<R, true, 256, 0>, 256, lds16, 1024, ga, (const uint4 *) d16, 4));
        chk("v2 u16, interleave, 256", runK2(k_v2<R, true, 256, 1>, 256, lds16, 1024, ga, (const uint4 *) d16, 4));
    }
    rep("6: 1024-thread WG x1/CU", runK(k_var<R, 0, 1024>, 1024, lds, 256, ga, 4));
}


// ---- realistic length mix: gamma(2.2) lengths with mean 350 clipped to [30, 2000] (foldseek_amd/synth.py), length-sorted,
//      stripes of 8, work items = whole stripes or column segments (cap / overlap like fsgpu.hip::gaplessItems) ----
#include <random>
#include <algorithm>
template <int R>
static void realMix(int nTargets, int L, int cap /* 0 = whole stripes */) {
    std::mt19937_64 rng(7);
    std::gamma_distribution<double> gd(2.2, 350.0 / 2.2);
    std::vector<int> len(nTargets);
    for (auto &x : len) x = std::min(2000, std::max(30, (int) std::lround(gd(rng))));
    std::sort(len.begin(), len.end());
    const int nStripes = (nTargets + 7) / 8;
    std::vector<uint32_t> sLen(nStripes);
    std::vector<uint64_t> sOff(nStripes);
    uint64_t totalU4 = 0, residues = 0;
    for (int s2 = 0; s2 < nStripes; s2++) {
        int mx = 0;
        for (int t = s2 * 8; t < std::min(nTargets, s2 * 8 + 8); t++) { mx = std::max(mx, len[t]); residues += len[t]; }
        sLen[s2] = (mx + 15) / 16; sOff[s2] = totalU4; totalU4 += (uint64_t) sLen[s2] * 8;
    }
    std::vector<uint8_t> h(totalU4 * 16, 21);
    for (int s2 = 0; s2 < nStripes; s2++)
        for (int j = 0; j < 8; j++) {
            const int t = s2 * 8 + j;
            if (t >= nTargets) continue;
            for (int c = 0; c < len[t]; c++) h[(sOff[s2] + (uint64_t) (c / 16) * 8 + j) * 16 + (c % 16)] = (uint8_t) (rng() % 20);
        }
    const int ov = R;
    std::vector<uint64_t> items;
    uint64_t units = 0;
    for (int s2 = 0; s2 < nStripes; s2++) {
        const uint32_t Ls = sLen[s2];
        if (cap == 0 || (int) Ls <= cap) { items.push_back(((uint64_t) s2 << 32) | Ls); units += Ls; continue; }
        const uint32_t K = (Ls + (cap - ov) - 1) / (cap - ov), fresh = (Ls + K - 1) / K;
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t b = k * fresh, e = std::min(Ls, (k + 1) * fresh);
            if (b >= e) break;
            const uint32_t b0 = b > (uint32_t) ov ? b - ov : 0;
            items.push_back(((uint64_t) s2 << 32) | (1ull << 31) | ((uint64_t) b0 << 16) | e); units += e - b0;
        }
    }
    std::stable_sort(items.begin(), items.end(), [](uint64_t a, uint64_t b) {
        return ((a & 0xffff) - ((a >> 16) & 0x7fff)) > ((b & 0xffff) - ((b >> 16) & 0x7fff)); });
    std::vector<int8_t> pssm(21 * L);
    for (auto &x : pssm) x = (int8_t) ((int) (rng() % 13) - 8);
    std::vector<uint32_t> ident(nStripes * 8);
    for (int t = 0; t < nStripes * 8; t++) ident[t] = t < nTargets ? t : 0xffffffffu;
    GaplessArgs ga{};
    void *d;
    hipMalloc(&d, h.size()); hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice); ga.scan = (const uint4 *) d;
    hipMalloc(&d, 8 * nStripes); hipMemcpy(d, sOff.data(), 8 * nStripes, hipMemcpyHostToDevice); ga.stripeOff = (const uint64_t *) d;
    hipMalloc(&d, 4 * nStripes); hipMemcpy(d, sLen.data(), 4 * nStripes, hipMemcpyHostToDevice); ga.stripeLen = (const uint32_t *) d;
    hipMalloc(&d, 4 * ident.size()); hipMemcpy(d, ident.data(), 4 * ident.size(), hipMemcpyHostToDevice); ga.stripeTargets = (const uint32_t *) d;
    {
        std::vector<uint4> recs(items.size());
        for (size_t i = 0; i < items.size(); i++) { const uint32_t st = (uint32_t) (items[i] >> 32); recs[i] = make_uint4(st, (uint32_t) items[i], (uint32_t) sOff[st], (uint32_t) (sOff[st] >> 32)); }
        hipMalloc(&d, 16 * recs.size()); hipMemcpy(d, recs.data(), 16 * recs.size(), hipMemcpyHostToDevice); ga.items = (const uint4 *) d; ga.nItems = (uint32_t) recs.size();
    }
    hipMalloc(&d, pssm.size()); hipMemcpy(d, pssm.data(), pssm.size(), hipMemcpyHostToDevice); ga.pssm = (const int8_t *) d;
    hipMalloc(&d, nStripes * 8); hipMemset(d, 0, nStripes * 8); ga.scores = (uint8_t *) d;
    hipMalloc(&d, 4); ga.queue = (uint32_t *) d;
    ga.nTargets = nTargets; ga.L = L; ga.cap = 255; ga.firstTile = ga.lastTile = 1;
    const int lds = gaplessLdsBytes(R);
    runK(k_gapless<R, false>, kGaplessBlock, lds, 768, ga, 20);
    for (int perCU : {2, 3, 4}) {
        const float ms = runK(k_gapless<R, false>, kGaplessBlock, lds, 256 * perCU, ga, 6);
        const double cellsReal = (double) residues * L, cellsDone = (double) units * 16 * 8 * (16.0 * R);
        printf("real mix R=%2d L=%3d cap=%3d items %6zu units %7llu  %d WG/CU  %7.3f ms  %6.2f Tcell/s useful  %6.2f Tcell/s issued  %6.1f cyc/wave-column/SIMD\n",
               R, L, cap, items.size(), (unsigned long long) units, perCU, ms, cellsReal / ms * 1e-9, cellsDone / ms * 1e-9, ms * 2.4e6 * 1024 / ((double) units * 16));
    }
}

int main(int argc, char **argv) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device CUs %d clock %d MHz\n", p.multiProcessorCount, p.clockRate / 1000);
    if (argc > 1 && !strcmp(argv[1], "short")) { suite<8>(12288, 22, 125); suite<12>(12288, 22, 190); return 0; }
    if (argc > 1 && !strcmp(argv[1], "real")) {
        realMix<21>(100000, 332, 0); realMix<21>(100000, 332, 91); realMix<21>(100000, 332, 64);
        realMix<24>(100000, 380, 0); realMix<24>(100000, 380, 91);
        return 0;
    }
    if (argc > 1) {     // fixed-cost probe: same stripe count, 1x / 2x / 4x columns -> the intercept is launch + LDS image build
        for (int len16 : {6, 11, 22, 44, 88}) { printf("len16 = %d\n", len16); suite<24>(12288, len16, 380); }
        return 0;
    }
    suite<24>(12288, 22, 380);
    suite<16>(12288, 22, 250);
    suite<32>(12288, 22, 500);
    return 0;
}
