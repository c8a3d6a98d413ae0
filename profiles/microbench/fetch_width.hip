// fetch_width.hip -- is a lone wave's issue interval beside busy neighbours (4.0 -> 4.4-4.65 clocks per instruction) an
// instruction-FETCH effect?  The same dependent FMA chain encoded in 4 bytes (v_fmac_f32_e32), 8 bytes (v_fma_f32, VOP3) and 8 bytes
// packed (v_pk_fma_f32), one wave per CU / one per SIMD / two per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define KERNEL(name, ins)                                                                                              \
    __global__ __launch_bounds__(64) void name(float* out, int iters) {                                                \
        float r = threadIdx.x * 1e-3f;                                                                                 \
        asm volatile("v_mov_b32 v20, %0\nv_mov_b32 v21, 0x3f7fff00\nv_mov_b32 v22, 0x3a000000\nv_mov_b32 v23, 0x3a000000\n" \
                     "v_mov_b32 v24, 0x3f7fff00\nv_mov_b32 v25, 0x3f7fff00\n"                                              \
                     "1:\n" REP64(ins) REP64(ins) REP64(ins) REP64(ins)                                                  \
                     "s_sub_u32 %1, %1, 1\ns_cmp_lg_u32 %1, 0\ns_cbranch_scc1 1b\nv_mov_b32 %0, v20\n"                    \
                     : "+v"(r), "+s"(iters) : : "v20", "v21", "v22", "v23", "v24", "v25", "scc");                       \
        out[blockIdx.x * 64 + threadIdx.x] = r;                                                                        \
    }
KERNEL(k_e32, "v_fmac_f32_e32 v20, v21, v22\n")
KERNEL(k_vop3, "v_fma_f32 v20, v20, v21, v22\n")
KERNEL(k_pk, "v_pk_fma_f32 v[20:21], v[20:21], v[24:25], v[22:23]\n")
KERNEL(k_lit, "v_fmac_f32_e32 v20, 0x3f7fff00, v22\n")
int main() {
    float* d; (void)hipMalloc(&d, 4 * 64 * 8192);
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    struct { const char* n; void (*k)(float*, int); int bytes; } ks[] = { {"v_fmac_f32_e32 (4 bytes)", k_e32, 4}, {"v_fma_f32 VOP3 (8 bytes)", k_vop3, 8},
                                                                         {"v_pk_fma_f32 (8 bytes)", k_pk, 8}, {"v_fmac_f32_e32 + literal (8 bytes)", k_lit, 8} };
    for (auto& k : ks)
        for (int wg : {256, 1024, 2048}) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k.k, dim3(wg), dim3(64), 0, 0, d, iters);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            std::printf("{\"instruction\": \"%s\", \"waves\": %d, \"clocks_per_instruction_per_wave\": %.3f}\n", k.n, wg,
                        ms * 1e-3 * clk * 1e3 / iters / 256.0);
        }
    return 0;
}
