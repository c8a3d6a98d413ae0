// int_rate.hip -- issue rate of the integer / packed-int16 vector instructions the lower-MAC decoder is made of (gfx950).
// Chip-level: 2048 workgroups x 16 waves, each REP x 64 copies of one instruction on 16 independent registers (or one dependent
// chain), wall time from HIP events -> wave64 instructions per clock per CU (4 = every SIMD issues one per 4 clocks).
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o int_rate int_rate.hip && ./int_rate
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP 256

template <int VARIANT> __global__ __launch_bounds__(1024) void k(unsigned* out, unsigned seed) {
    unsigned a[16];
    for (int i = 0; i < 16; i++) a[i] = seed * (i + 1);
    unsigned x = seed ^ 0x12345678u, y = seed + 77u;
#pragma unroll 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (VARIANT == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                if (VARIANT == 1) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                if (VARIANT == 2) asm volatile("v_pk_add_u16 %0, %0, %1 op_sel:[1,1] op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(x));
                if (VARIANT == 3) asm volatile("v_pk_sub_i16 %0, %0, %1 op_sel_hi:[0,1]" : "+v"(a[i]) : "v"(x));
                if (VARIANT == 4) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                if (VARIANT == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                if (VARIANT == 6) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                if (VARIANT == 7) asm volatile("v_bfe_i32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(x));
                if (VARIANT == 8) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
                if (VARIANT == 9) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                if (VARIANT == 10) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                if (VARIANT == 11) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[0]) : "v"(x));                 // dependent chain
                if (VARIANT == 12) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[0]) : "v"(x));                 // dependent chain
                if (VARIANT == 13) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(x));                    // dependent chain
                if (VARIANT == 14) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(a[i]) : "v"(x), "v"(y));
                if (VARIANT == 15) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(x), "v"(y));
            }
        }
    }
    unsigned acc = 0;
    for (int i = 0; i < 16; i++) acc ^= a[i];
    out[threadIdx.x] = acc;
}

int main() {
    unsigned* out;
    (void)hipMalloc(&out, 1024 * sizeof(unsigned));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // waves per SIMD resident at a time: 16-wave workgroups = 4 per SIMD; 4-wave workgroups x many per CU also fill; a 1-wave-per-SIMD
    // variant: 256 workgroups of 4 waves (one workgroup per CU)
#define CHIP(V, NAME) for (int mode = 0; mode < 3; mode++) { \
        const int threads = mode == 0 ? 1024 : 256, blocks = mode == 0 ? 2048 : mode == 1 ? 256 : 512; \
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, out, 3u); (void)hipEventRecord(e0, 0); \
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, out, 3u); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); \
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); const double instr = (double)blocks * (threads / 64) * REP * 64; \
        printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave64_instr_per_clk_per_CU_at_2.4GHz\": %.3f}\n", NAME, mode == 0 ? 4 : mode == 1 ? 1 : 2, ms, \
               instr / (ms * 1e-3) / 2.4e9 / 256); }
    CHIP(0, "v_add_u32") CHIP(1, "v_pk_add_u16") CHIP(2, "v_pk_add_u16 op_sel swap") CHIP(3, "v_pk_sub_i16 op_sel broadcast") CHIP(4, "v_pk_max_i16")
    CHIP(5, "v_perm_b32") CHIP(6, "v_bfi_b32") CHIP(7, "v_bfe_i32") CHIP(8, "v_lshrrev_b32") CHIP(9, "v_and_or_b32") CHIP(10, "v_add3_u32")
    CHIP(11, "v_pk_add_u16 dependent chain") CHIP(12, "v_pk_max_i16 dependent chain") CHIP(13, "v_add_u32 dependent chain") CHIP(14, "v_bitop3_b32")
    CHIP(15, "v_bitop3_b32 (xor3)")
    return 0;
}
