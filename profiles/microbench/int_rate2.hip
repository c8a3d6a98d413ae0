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
                if (VARIANT == 0) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 1) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 3) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 4) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 5) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 6) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 7) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 9) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 10) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 11) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 12) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 13) asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 14) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 15) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 16) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 17) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 18) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 19) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 20) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 21) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 22) asm volatile("v_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 23) asm volatile("v_pk_lshrrev_b16 %0, 1, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 24) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 25) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 26) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 27) asm volatile("v_dot4_i32_i8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 28) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 29) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
                if (VARIANT == 30) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
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
    CHIP(0, "v_and_b32")
    CHIP(1, "v_or_b32")
    CHIP(2, "v_xor_b32")
    CHIP(3, "v_sub_u32")
    CHIP(4, "v_max_i32")
    CHIP(5, "v_min_u32")
    CHIP(6, "v_lshlrev_b32")
    CHIP(7, "v_ashrrev_i32")
    CHIP(8, "v_mov_b32")
    CHIP(9, "v_lshl_add_u32")
    CHIP(10, "v_lshl_or_b32")
    CHIP(11, "v_or3_b32")
    CHIP(12, "v_alignbit_b32")
    CHIP(13, "v_bfe_u32")
    CHIP(14, "v_mad_u32_u24")
    CHIP(15, "v_max_i16")
    CHIP(16, "v_add_u16")
    CHIP(17, "v_sub_u16")
    CHIP(18, "v_fma_f32")
    CHIP(19, "v_add_f32")
    CHIP(20, "v_add_co_u32 (vcc)")
    CHIP(21, "v_cndmask_b32")
    CHIP(22, "v_min_i16 ")
    CHIP(23, "v_pk_lshrrev_b16")
    CHIP(24, "v_pk_sub_u16")
    CHIP(25, "v_pk_min_i16")
    CHIP(26, "v_sad_u8")
    CHIP(27, "v_dot4_i32_i8")
    CHIP(28, "v_mul_u32_u24")
    CHIP(29, "v_mul_lo_u32")
    CHIP(30, "v_bfrev_b32")
    return 0;
}
