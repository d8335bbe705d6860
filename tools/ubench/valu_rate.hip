// valu_rate.hip -- instruction issue-rate microbenchmark for gfx950 (tools, not product): cycles per wave64
// instruction for the integer ops the DP kernels are built from.  One workgroup per CU, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, asmline)                                                                       \
    __global__ void name(uint32_t *out, int iters) {                                                \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = threadIdx.x * 3 + 1, c = 0x00010001u;                                           \
        for (int i = 0; i < iters; i++) {                                                           \
            REP16(asm volatile(asmline : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) \
        }                                                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;        \
    }
// each asm block = 8 independent instructions
#define OP8(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define OP8S(op, suf) op " %0, %0, %8 " suf "\n" op " %1, %1, %8 " suf "\n" op " %2, %2, %8 " suf "\n" op " %3, %3, %8 " suf "\n" op " %4, %4, %8 " suf "\n" op " %5, %5, %8 " suf "\n" op " %6, %6, %8 " suf "\n" op " %7, %7, %8 " suf "\n"
#define OP8_3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"

KERNEL(k_pk_add_i16_clamp, OP8S("v_pk_add_i16", "clamp"))
KERNEL(k_pk_add_i16, OP8("v_pk_add_i16"))
KERNEL(k_pk_max_i16, OP8("v_pk_max_i16"))
KERNEL(k_pk_sub_u16_clamp, OP8S("v_pk_sub_u16", "clamp"))
KERNEL(k_add_u32, OP8("v_add_u32"))
KERNEL(k_max_i32, OP8("v_max_i32"))
KERNEL(k_max3_i32, OP8_3("v_max3_i32"))
KERNEL(k_add3_u32, OP8_3("v_add3_u32"))
KERNEL(k_and_b32, OP8("v_and_b32"))
KERNEL(k_perm_b32, OP8_3("v_perm_b32"))
KERNEL(k_fma_f32, OP8_3("v_fma_f32"))
KERNEL(k_pk_fma_f32_dummy, OP8("v_pk_add_f16"))
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_mov_dpp_wave, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dot4_i32_i8, OP8_3("v_dot4_i32_i8"))
KERNEL(k_sad_u8, OP8_3("v_sad_u8"))
KERNEL(k_pk_mad_i16, OP8_3("v_pk_mad_i16"))
KERNEL(k_pk_min_u16, OP8("v_pk_min_u16"))
KERNEL(k_lshl_or, OP8_3("v_lshl_or_b32"))
KERNEL(k_pk_maximum3_f16, OP8_3("v_pk_maximum3_f16"))
KERNEL(k_pk_max_f16, OP8("v_pk_max_f16"))
KERNEL(k_pk_add_f16_clamp, OP8S("v_pk_add_f16", "clamp"))
KERNEL(k_max_f32, OP8("v_max_f32"))
KERNEL(k_add_f32, OP8("v_add_f32"))
KERNEL(k_maximum3_f32, OP8_3("v_maximum3_f32"))
KERNEL(k_max3_f32, OP8_3("v_max3_f32"))
KERNEL(k_max3_i16, OP8_3("v_max3_i16"))
KERNEL(k_pk_fma_f16, OP8_3("v_pk_fma_f16"))
KERNEL(k_bfi, OP8_3("v_bfi_b32"))

__global__ void k_lds_b128(uint32_t *out, int iters) {
    extern __shared__ uint4 sm[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = make_uint4(i, i, i, i);
    __syncthreads();
    uint4 acc = make_uint4(0, 0, 0, 0);
    int idx = threadIdx.x & 63;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint4 v = sm[idx + 64 * ((k + i) & 63)];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main() {
    int dev = 0; hipSetDevice(dev);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    int cus = p.multiProcessorCount;
    double clk = p.clockRate * 1e3;
    printf("device %s CUs %d clock %.0f MHz\n", p.name, cus, clk / 1e6);
    uint32_t *out; hipMalloc(&out, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct K { const char *name; void (*fn)(uint32_t *, int); };
    K ks[] = {{"v_pk_add_i16 clamp", k_pk_add_i16_clamp}, {"v_pk_add_i16", k_pk_add_i16}, {"v_pk_max_i16", k_pk_max_i16},
              {"v_pk_sub_u16 clamp", k_pk_sub_u16_clamp}, {"v_pk_min_u16", k_pk_min_u16}, {"v_pk_mad_i16", k_pk_mad_i16}, {"v_add_u32", k_add_u32}, {"v_max_i32", k_max_i32},
              {"v_max3_i32", k_max3_i32}, {"v_add3_u32", k_add3_u32}, {"v_and_b32", k_and_b32}, {"v_perm_b32", k_perm_b32},
              {"v_lshl_or_b32", k_lshl_or}, {"v_bfi_b32", k_bfi}, {"v_pk_maximum3_f16", k_pk_maximum3_f16}, {"v_pk_max_f16", k_pk_max_f16}, {"v_pk_add_f16 clamp", k_pk_add_f16_clamp},
              {"v_max_f32", k_max_f32}, {"v_add_f32", k_add_f32}, {"v_maximum3_f32", k_maximum3_f32}, {"v_max3_f32", k_max3_f32}, {"v_max3_i16", k_max3_i16}, {"v_pk_fma_f16", k_pk_fma_f16},
              {"v_fma_f32", k_fma_f32}, {"v_pk_add_f16", k_pk_fma_f32_dummy}, {"v_mov_dpp row_shr", k_mov_dpp}, {"v_mov_dpp wave_shr", k_mov_dpp_wave},
              {"v_dot4_i32_i8", k_dot4_i32_i8}, {"v_sad_u8", k_sad_u8}};
    const int iters = 2000;
    for (int wavesPerSimd : {1, 2, 4}) {
        printf("--- %d wave(s) per SIMD (block %d threads, %d blocks) ---\n", wavesPerSimd, 256 * wavesPerSimd, cus);
        for (auto &k : ks) {
            hipLaunchKernelGGL(k.fn, dim3(cus), dim3(256 * wavesPerSimd), 0, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(cus), dim3(256 * wavesPerSimd), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double instrPerSimd = (double) iters * 16 * 8 * wavesPerSimd;
            printf("%-22s %8.3f ms  -> %.2f ns / wave-instr / SIMD  (= %.2f cycles @2.4GHz)\n", k.name, ms, ms * 1e6 / instrPerSimd, ms * 1e6 / instrPerSimd * 2.4);
        }
    }
    for (int wavesPerSimd : {1, 2, 4}) {
        hipLaunchKernelGGL(k_lds_b128, dim3(cus), dim3(256 * wavesPerSimd), 65536, 0, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_lds_b128, dim3(cus), dim3(256 * wavesPerSimd), 65536, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = (double) iters * 16 * 16 * 256 * wavesPerSimd;   // per CU
        printf("ds_read_b128 %d w/SIMD: %8.3f ms -> %.1f B/ns/CU (= %.1f B/clk @2.4GHz)\n", wavesPerSimd, ms, bytes / (ms * 1e6), bytes / (ms * 1e6) / 2.4);
    }
    return 0;
}
