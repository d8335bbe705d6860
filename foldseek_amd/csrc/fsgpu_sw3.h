// fsgpu_sw3.h -- private: launchers of the k_sw3 family (fsgpu_sw3.hip, built once without and once with the AA table).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct fsgpu_ctx;
namespace fs {
struct Sw3Args;
struct Sw3ImgQuery;
}

// one launch for the workgroups of all queries with rlo <= rows per lane < rlo + 8 (rlo = 1 or 9); lds = dynamic LDS of the largest of them
int fsgpuLaunchSw3NA(fsgpu_ctx *ctx, int rlo, int HL, const fs::Sw3Args &sa, int nBlocks, int waves, int lds, hipStream_t stream);
int fsgpuLaunchSw3AA(fsgpu_ctx *ctx, int rlo, int HL, const fs::Sw3Args &sa, int nBlocks, int waves, int lds, hipStream_t stream);
// images of nq queries; maxDwords = dwords of the largest image (sizes the grid)
int fsgpuLaunchSw3Image(fsgpu_ctx *ctx, const fs::Sw3ImgQuery *dq, int nq, int maxDwords, const uint8_t *data, const int8_t *mat3, const int8_t *matA,
                        uint32_t *img, bool hasAA, hipStream_t stream);
