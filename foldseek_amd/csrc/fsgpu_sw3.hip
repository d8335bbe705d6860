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

template <int R, int HL>
int launchT(fsgpu_ctx *ctx, const Sw3Args &sa, int nBlocks, int waves, hipStream_t stream) {
    constexpr bool AA = FS_SW3_AA != 0;
    const int lds = sw3LdsBytes(R, HL, AA, waves);
    static thread_local uint64_t attrDevs = 0;       // devices on which this thread has set the attribute (it is per device)
    const uint64_t devBit = 1ull << (ctx->device & 63);
    if (!(attrDevs & devBit)) {
        HIPCHK(hipFuncSetAttribute((const void *) k_sw3<R, AA, HL>, hipFuncAttributeMaxDynamicSharedMemorySize, sw3LdsBytes(R, HL, AA, 8)));
        attrDevs |= devBit;
    }
    hipLaunchKernelGGL((k_sw3<R, AA, HL>), dim3(nBlocks), dim3(64 * waves), lds, stream, sa);
    HIPCHK(hipGetLastError());
    return FSGPU_OK;
}

} // namespace

#if FS_SW3_AA
int fsgpuLaunchSw3AA(fsgpu_ctx *ctx, int R, int HL, const Sw3Args &sa, int nBlocks, int waves, hipStream_t stream) {
#else
int fsgpuLaunchSw3NA(fsgpu_ctx *ctx, int R, int HL, const Sw3Args &sa, int nBlocks, int waves, hipStream_t stream) {
#endif
#define FS_C32(RR) case RR: return launchT<RR, 32>(ctx, sa, nBlocks, waves, stream);
#define FS_C64(RR) case RR: return launchT<RR, 64>(ctx, sa, nBlocks, waves, stream);
    if (HL == 32) {
        switch (R) {
            FS_C32(1) FS_C32(2) FS_C32(3) FS_C32(4) FS_C32(5) FS_C32(6) FS_C32(7) FS_C32(8)
            FS_C32(9) FS_C32(10) FS_C32(11) FS_C32(12) FS_C32(13) FS_C32(14) FS_C32(15) FS_C32(16)
            default: break;
        }
    } else if (HL == 64) {
        switch (R) {
            FS_C64(1) FS_C64(2) FS_C64(3) FS_C64(4) FS_C64(5) FS_C64(6) FS_C64(7) FS_C64(8)
            FS_C64(9) FS_C64(10) FS_C64(11) FS_C64(12) FS_C64(13) FS_C64(14) FS_C64(15) FS_C64(16)
            default: break;
        }
    }
#undef FS_C32
#undef FS_C64
    ctx->err = "internal: bad k_sw3 class";
    return FSGPU_E_ARG;
}

#if !FS_SW3_AA
int fsgpuLaunchSw3Image(fsgpu_ctx *ctx, const Sw3ImgQuery *dq, int nq, int maxDwords, const uint8_t *data, const int8_t *mat3, const int8_t *matA,
                        uint32_t *img, bool hasAA, hipStream_t stream) {
    if (nq <= 0) return FSGPU_OK;
    const int bx = std::max(1, std::min(64, (maxDwords + 1023) / 1024));
    hipLaunchKernelGGL(k_sw3_image, dim3(bx, nq), dim3(256), 0, stream, dq, data, mat3, matA, img, hasAA ? 1 : 0);
    HIPCHK(hipGetLastError());
    return FSGPU_OK;
}
#endif
