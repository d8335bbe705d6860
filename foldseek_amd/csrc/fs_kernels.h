// fs_kernels.h -- shared constants / layout functions used by the HIP kernels and by the host code that
// builds the LDS images.  gfx950 (MI355X) only: wave64, 64 LDS banks, ds_read_b128 serviced in 16-lane groups.
#pragma once
#include <stdint.h>

namespace fs {

constexpr int kAlphabet = 21;          // 20 states + X
constexpr int kDeadCode = 21;          // extra profile row: "past the end of this target", never scores
constexpr int kGaplessLanes = 8;       // lanes per target in the gapless scan (8 targets per wave64)
constexpr int kStripeTargets = 8;      // targets interleaved per 128-byte line of the scan layout
#ifndef FSGPU_GAPLESS_BLOCK
#define FSGPU_GAPLESS_BLOCK 256
#endif
constexpr int kGaplessBlock = FSGPU_GAPLESS_BLOCK;   // threads per workgroup of the gapless scan (one LDS image each)
constexpr int kGaplessMaxR = 32;       // register rows per strip per lane of a ROW TILE: a tile covers 16*R <= 512 query rows
// Queries of up to 16 * kGaplessMaxRUntiled = 896 residues run untiled as well: R = 33..56 registers per lane (92 VGPRs at R = 48, no
// scratch), an LDS image of up to 14 chunks = 79 KB, two workgroups of 8 waves per CU instead of three of 4 (measured on 513..896-residue
// queries at 1M targets: 4 / 6 / 8 waves per workgroup = 5.00 / 4.97 / 4.89 ms per query).  Per column the 3 non-DP instructions then
// weigh 4 % instead of the row-tiled kernel's border hand-over.
constexpr int kGaplessMaxRUntiled = 56;
// threads per workgroup for a register count
#ifndef FSGPU_GAPLESS_BIGBLOCK
#define FSGPU_GAPLESS_BIGBLOCK (2 * FSGPU_GAPLESS_BLOCK)
#endif
__host__ __device__ constexpr int gaplessBlockThreads(int R) { return R <= 36 ? kGaplessBlock : FSGPU_GAPLESS_BIGBLOCK; }
constexpr int kSwMaxR = 8;             // register rows per lane in the SW wavefront -> 64*R <= 512 rows per tile
constexpr uint32_t kFloor2 = 0x80008000u; // packed (INT16_MIN, INT16_MIN): the gapless recurrence's "zero"

// ------------------------------------------------------------------------------------------------------------
// Gapless scan LDS image.  Lane g (0..7) of a target group owns query rows [g*2R, g*2R+2R); register r packs
// (row g*2R + r) in the low half and (row g*2R + R + r) in the high half, so the diagonal hand-off is a whole-
// register move.  One 256-byte LDS bank row holds, for one profile row and one 4-register chunk k, two copies
// (A: groups 0,1,4,5; B: groups 2,3,6,7) x 8 lanes x 16 B.  The image is chunk-major: chunk k of profile row c sits
// at k * gaplessChunkBytes() + c * 256, hence (1) the bank of an access depends only on (copy, g): every
// ds_read_b128 is conflict free by construction, and (2) the byte address of a row is (c << 8) | laneOffset with
// laneOffset < 256, which one v_perm_b32 assembles from the packed residue word.
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int gaplessChunkBytes() { return (kAlphabet + 1) * 256; }
__host__ __device__ constexpr int gaplessChunks(int R) { return (R + 3) / 4; }
__host__ __device__ constexpr int gaplessLdsBytes(int R) { return gaplessChunks(R) * gaplessChunkBytes(); }

// ------------------------------------------------------------------------------------------------------------
// SW wavefront LDS image: lane l owns query rows [l*R, l*R+R) of the current tile.  Registers are fetched in
// chunks of 4/2/1 dwords; chunk c occupies a contiguous region of 64 lanes x width dwords, so lane stride equals
// the access width and a wave's read of one chunk is a linear 64*width*4-byte sweep: conflict free although
// every lane reads a different profile row (lanes sit on different target columns of the anti-diagonal).
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int swChunkWidth(int R, int r) {
    return (r < (R / 4) * 4) ? 4 : ((R % 4) >= 2 && r < (R / 4) * 4 + 2) ? 2 : 1;
}
// dword index inside one profile row (of 64*R dwords) for (lane, register r)
__host__ __device__ constexpr int swDwordIndex(int R, int lane, int r) {
    int full = (R / 4) * 4;
    if (r < full) return (r / 4) * 256 + lane * 4 + (r % 4);
    int off = (R / 4) * 256;
    int rem = R % 4;
    if (rem >= 2) {
        if (r < full + 2) return off + lane * 2 + (r - full);
        off += 128;
        return off + lane;          // rem == 3: last single
    }
    return off + lane;              // rem == 1
}
__host__ __device__ constexpr int swRowDwords(int R) { return 64 * R; }

} // namespace fs
