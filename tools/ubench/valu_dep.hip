// valu_dep.hip -- (tools) does a dependent packed-op chain with the mandatory s_nop, or a 3-source op whose operands
// share a VGPR bank, issue slower than independent ops?  4 waves / SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define K(name, body, clob...)                                                        \
    __global__ void name(uint32_t *out, int iters) {                                  \
        uint32_t r = 0;                                                               \
        asm volatile("v_mov_b32 v0, 0\nv_mov_b32 v1, 0\nv_mov_b32 v2, 0\nv_mov_b32 v3, 0\nv_mov_b32 v4, 0\nv_mov_b32 v5, 0\nv_mov_b32 v6, 0\nv_mov_b32 v7, 0\n" \
                     "v_mov_b32 v8, 0\nv_mov_b32 v9, 0\nv_mov_b32 v10, 0\nv_mov_b32 v11, 0\nv_mov_b32 v12, 0\nv_mov_b32 v13, 0\nv_mov_b32 v14, 0\nv_mov_b32 v15, 0\n" ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15"); \
        for (int i = 0; i < iters; i++) {                                             \
            REP8(asm volatile(body ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15");) \
        }                                                                             \
        asm volatile("v_mov_b32 %0, v0" : "=v"(r));                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                               \
    }
// each body = 8 VALU instructions
K(k_indep_add, "v_pk_add_f16 v0, v0, v8 clamp\nv_pk_add_f16 v1, v1, v9 clamp\nv_pk_add_f16 v2, v2, v10 clamp\nv_pk_add_f16 v3, v3, v11 clamp\nv_pk_add_f16 v4, v4, v12 clamp\nv_pk_add_f16 v5, v5, v13 clamp\nv_pk_add_f16 v6, v6, v14 clamp\nv_pk_add_f16 v7, v7, v15 clamp\n")
K(k_dep_add_nop, "v_pk_add_f16 v0, v0, v8 clamp\ns_nop 0\nv_pk_add_f16 v0, v0, v9 clamp\ns_nop 0\nv_pk_add_f16 v0, v0, v10 clamp\ns_nop 0\nv_pk_add_f16 v0, v0, v11 clamp\ns_nop 0\nv_pk_add_f16 v0, v0, v12 clamp\ns_nop 0\nv_pk_add_f16 v0, v0, v13 clamp\ns_nop 0\nv_pk_add_f16 v0, v0, v14 clamp\ns_nop 0\nv_pk_add_f16 v0, v0, v15 clamp\ns_nop 0\n")
K(k_shift_add, "v_pk_add_f16 v7, v6, v15 clamp\nv_pk_add_f16 v6, v5, v14 clamp\nv_pk_add_f16 v5, v4, v13 clamp\nv_pk_add_f16 v4, v3, v12 clamp\nv_pk_add_f16 v3, v2, v11 clamp\nv_pk_add_f16 v2, v1, v10 clamp\nv_pk_add_f16 v1, v0, v9 clamp\nv_pk_add_f16 v0, v7, v8 clamp\n")
K(k_max3_diffbank, "v_pk_maximum3_f16 v0, v0, v5, v10\nv_pk_maximum3_f16 v1, v1, v6, v11\nv_pk_maximum3_f16 v2, v2, v7, v8\nv_pk_maximum3_f16 v3, v3, v4, v9\nv_pk_maximum3_f16 v0, v0, v5, v10\nv_pk_maximum3_f16 v1, v1, v6, v11\nv_pk_maximum3_f16 v2, v2, v7, v8\nv_pk_maximum3_f16 v3, v3, v4, v9\n")
K(k_max3_samebank, "v_pk_maximum3_f16 v0, v0, v4, v8\nv_pk_maximum3_f16 v1, v1, v5, v9\nv_pk_maximum3_f16 v2, v2, v6, v10\nv_pk_maximum3_f16 v3, v3, v7, v11\nv_pk_maximum3_f16 v0, v0, v4, v8\nv_pk_maximum3_f16 v1, v1, v5, v9\nv_pk_maximum3_f16 v2, v2, v6, v10\nv_pk_maximum3_f16 v3, v3, v7, v11\n")
K(k_max3_twobank, "v_pk_maximum3_f16 v0, v0, v4, v9\nv_pk_maximum3_f16 v1, v1, v5, v10\nv_pk_maximum3_f16 v2, v2, v6, v11\nv_pk_maximum3_f16 v3, v3, v7, v8\nv_pk_maximum3_f16 v0, v0, v4, v9\nv_pk_maximum3_f16 v1, v1, v5, v10\nv_pk_maximum3_f16 v2, v2, v6, v11\nv_pk_maximum3_f16 v3, v3, v7, v8\n")
K(k_max3_dep_nop, "v_pk_maximum3_f16 v0, v0, v5, v10\ns_nop 0\nv_pk_maximum3_f16 v0, v0, v6, v11\ns_nop 0\nv_pk_maximum3_f16 v0, v0, v7, v9\ns_nop 0\nv_pk_maximum3_f16 v0, v0, v5, v10\ns_nop 0\nv_pk_maximum3_f16 v0, v0, v6, v11\ns_nop 0\nv_pk_maximum3_f16 v0, v0, v7, v9\ns_nop 0\nv_pk_maximum3_f16 v0, v0, v5, v10\ns_nop 0\nv_pk_maximum3_f16 v0, v0, v6, v11\ns_nop 0\n")
K(k_add_samebank, "v_pk_add_f16 v0, v4, v8 clamp\nv_pk_add_f16 v1, v5, v9 clamp\nv_pk_add_f16 v2, v6, v10 clamp\nv_pk_add_f16 v3, v7, v11 clamp\nv_pk_add_f16 v0, v4, v8 clamp\nv_pk_add_f16 v1, v5, v9 clamp\nv_pk_add_f16 v2, v6, v10 clamp\nv_pk_add_f16 v3, v7, v11 clamp\n")
K(k_add_diffbank, "v_pk_add_f16 v0, v5, v10 clamp\nv_pk_add_f16 v1, v6, v11 clamp\nv_pk_add_f16 v2, v7, v8 clamp\nv_pk_add_f16 v3, v4, v9 clamp\nv_pk_add_f16 v0, v5, v10 clamp\nv_pk_add_f16 v1, v6, v11 clamp\nv_pk_add_f16 v2, v7, v8 clamp\nv_pk_add_f16 v3, v4, v9 clamp\n")
// DP-like mix: 4 adds (shift chain) + 2 max3
K(k_mix, "v_pk_add_f16 v3, v2, v11 clamp\nv_pk_add_f16 v2, v1, v10 clamp\nv_pk_add_f16 v1, v0, v9 clamp\nv_pk_add_f16 v0, v7, v8 clamp\nv_pk_maximum3_f16 v4, v4, v3, v2\nv_pk_maximum3_f16 v5, v5, v1, v0\nv_pk_add_f16 v6, v6, v12 clamp\nv_pk_add_f16 v7, v7, v13 clamp\n")
template <typename Kn> static void run(const char *nm, Kn k, int waves, uint32_t *out) {
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    const int iters = 20000;
    float best = 1e9;
    for (int t = 0; t < 3; t++) {
        (void) hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256 * waves), 0, 0, out, iters);
        (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        float ms; (void) hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double ninstr = (double) iters * 64 * waves;   // VALU instr per SIMD
    printf("%-18s %d w/SIMD %7.3f ms -> %.2f cycles / VALU instr / SIMD\n", nm, waves, best, best * 2.4e6 / ninstr);
}
int main() {
    uint32_t *out; (void) hipMalloc(&out, 256 * 1024 * 4);
    for (int w : {1, 2, 4}) {
        run("indep add", k_indep_add, w, out);
        run("dep add + s_nop", k_dep_add_nop, w, out);
        run("shift-chain add", k_shift_add, w, out);
        run("add same bank", k_add_samebank, w, out);
        run("add diff bank", k_add_diffbank, w, out);
        run("max3 diff banks", k_max3_diffbank, w, out);
        run("max3 two banks", k_max3_twobank, w, out);
        run("max3 same bank", k_max3_samebank, w, out);
        run("max3 dep + s_nop", k_max3_dep_nop, w, out);
        run("mix 6add+2max3", k_mix, w, out);
    }
    return 0;
}
