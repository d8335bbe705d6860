// fsgpu_sw3.hip -- instantiations and launchers of k_sw3 / k_sw3_image (k_sw3.hpp).  Compiled twice (FS_SW3_AA = 0 / 1: the 3Di-only and the
// 3Di + AA kernels) so that the two halves build in parallel; the orchestration (fsgpu_sw_multi_dir_c) lives in fsgpu.hip.
#include <hip/hip_runtime.h>
#include "fsgpu_ctx.h"
#if !FS_SW3_AA
#define FS_SW3_DEFINE_IMAGE_KERNEL 1
#endif
#include "k_sw3.hpp"
#include "fsgpu_sw3.h"

#ifndef FS_SW3_AA
#error "compile with -DFS_SW3_AA=0 or 1"
#endif

namespace {

template <int HL, int RLO>
int launchT(fsgpu_ctx *ctx, const Sw3Args &sa, int nBlocks, int waves, int lds, hipStream_t stream) {
    constexpr bool AA = FS_SW3_AA != 0;
    static thread_local uint64_t attrDevs = 0;       // devices on which this thread has set the attribute (it is per device)
    const uint64_t devBit = 1ull << (ctx->device & 63);
    if (!(attrDevs & devBit)) {
        HIPCHK(hipFuncSetAttribute((const void *) k_sw3<AA, HL, RLO>, hipFuncAttributeMaxDynamicSharedMemorySize, sw3LdsBytes(RLO + 7, HL, AA, 8)));
        attrDevs |= devBit;
    }
    hipLaunchKernelGGL((k_sw3<AA, HL, RLO>), dim3(nBlocks), dim3(64 * waves), lds, stream, sa);
    HIPCHK(hipGetLastError());
    return FSGPU_OK;
}

} // namespace

// rlo: 1 (queries with 1..8 rows per lane), 9 (9..16) or -- 16 lanes per target pair only -- 17 (17..24); lds: dynamic LDS of the largest class among the launch's workgroups
#if FS_SW3_AA
int fsgpuLaunchSw3AA(fsgpu_ctx *ctx, int rlo, int HL, const Sw3Args &sa, int nBlocks, int waves, int lds, hipStream_t stream) {
#else
int fsgpuLaunchSw3NA(fsgpu_ctx *ctx, int rlo, int HL, const Sw3Args &sa, int nBlocks, int waves, int lds, hipStream_t stream) {
#endif
    if (HL == 16 && rlo == 1) return launchT<16, 1>(ctx, sa, nBlocks, waves, lds, stream);
    if (HL == 16 && rlo == 9) return launchT<16, 9>(ctx, sa, nBlocks, waves, lds, stream);
    if (HL == 16 && rlo == 17) return launchT<16, 17>(ctx, sa, nBlocks, waves, lds, stream);
    if (HL == 32 && rlo == 1) return launchT<32, 1>(ctx, sa, nBlocks, waves, lds, stream);
    if (HL == 32 && rlo == 9) return launchT<32, 9>(ctx, sa, nBlocks, waves, lds, stream);
    if (HL == 64 && rlo == 1) return launchT<64, 1>(ctx, sa, nBlocks, waves, lds, stream);
    if (HL == 64 && rlo == 9) return launchT<64, 9>(ctx, sa, nBlocks, waves, lds, stream);
    ctx->err = "internal: bad k_sw3 class";
    return FSGPU_E_ARG;
}

#if !FS_SW3_AA
int fsgpuLaunchSw3Image(fsgpu_ctx *ctx, const Sw3ImgQuery *dq, int nq, int maxDwords, const uint8_t *data, const int8_t *mat3, const int8_t *matA,
                        uint32_t *img, bool hasAA, hipStream_t stream) {
    if (nq <= 0) return FSGPU_OK;
    const int bx = std::max(1, std::min(64, (maxDwords + 1023) / 1024));
    // a query can have two images (32- and 64-lane shape): up to 2 x 65535 of them per call, more than one grid's y extent
    for (int i0 = 0; i0 < nq; i0 += 65535) {
        hipLaunchKernelGGL(k_sw3_image, dim3(bx, std::min(65535, nq - i0)), dim3(256), 0, stream, dq + i0, data, mat3, matA, img, hasAA ? 1 : 0);
        HIPCHK(hipGetLastError());
    }
    return FSGPU_OK;
}
#endif
