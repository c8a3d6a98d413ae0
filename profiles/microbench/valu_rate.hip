// valu_rate.hip -- issue-rate / latency microbenchmark for the VALU instructions the demodulator is made of (gfx950).
// One wave64 on one SIMD (then one wave on each SIMD, then two and four waves per SIMD); each variant runs REP x 64 copies of one instruction (independent accumulators, or one
// dependent chain) between two s_memtime reads.  Prints shader-clock cycles per instruction.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP 256

template <int VARIANT> __global__ __launch_bounds__(1024) void k(float* out, long long* cyc, float seed) {
    f2 a[16];
    float s[16];
    for (int i = 0; i < 16; i++) { a[i] = f2{ seed + i, seed - i }; s[i] = seed * i; }
    f2 x = f2{ seed, 1.0f - seed }, y = f2{ 0.5f, 0.25f };
    float xs = seed, ys = 0.5f;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (VARIANT == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(xs), "v"(ys));
                if (VARIANT == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if (VARIANT == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "v"(y));
                if (VARIANT == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                if (VARIANT == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                if (VARIANT == 5) asm volatile("v_mov_b32_dpp %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(xs));
                if (VARIANT == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[0]) : "v"(xs), "v"(ys));          // dependent chain
                if (VARIANT == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));         // dependent chain
                if (VARIANT == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(xs));
                if (VARIANT == 9) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[i]) : "v"(xs), "v"(ys));
                if (VARIANT == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(y));          // one SGPR-pair operand
                if (VARIANT == 11) asm volatile("v_rndne_f32 %0, %0" : "+v"(s[i]));
                if (VARIANT == 12) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(xs), "v"(ys));
                if (VARIANT == 13) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[i]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int i = 0; i < 16; i++) acc += a[i].x + a[i].y + s[i];
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[VARIANT + 16 * (blockDim.x == 64 ? 0 : blockDim.x == 256 ? 1 : blockDim.x == 512 ? 2 : 3)] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 1024 * sizeof(float));
    (void)hipMallocManaged(&cyc, 64 * sizeof(long long));
    const char* names[14] = { "v_fma_f32 (independent)", "v_pk_fma_f32 (independent, 3 VGPR pairs)", "v_pk_fma_f32 op_sel broadcast of src1",
                              "v_pk_mul_f32", "v_pk_add_f32", "v_mov_b32_dpp row_shr:2", "v_fma_f32 dependent chain", "v_pk_fma_f32 dependent chain",
                              "v_add_f32", "v_fmac_f32", "v_pk_fma_f32 with an SGPR-pair operand", "v_rndne_f32", "v_med3_f32", "v_sqrt_f32" };
#define RUN(V) for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, 0.37f); \
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(256), 0, 0, out, cyc, 0.37f); hipLaunchKernelGGL(k<V>, dim3(1), dim3(512), 0, 0, out, cyc, 0.37f); \
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(1024), 0, 0, out, cyc, 0.37f); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
    (void)hipDeviceSynchronize();
    for (int v = 0; v < 14; v++)
        printf("{\"instr\": \"%s\", \"ticks_per_instr_1_wave_per_simd\": %.3f, \"same_4_waves_one_per_simd\": %.3f, \"ticks_per_instr_per_simd_2_waves_per_simd\": %.3f, \"ticks_per_instr_per_simd_4_waves_per_simd\": %.3f}\n",
               names[v], (double)cyc[v] / (REP * 64.0), (double)cyc[v + 16] / (REP * 64.0), (double)cyc[v + 32] / (REP * 64.0) / 2.0, (double)cyc[v + 48] / (REP * 64.0) / 4.0);
    // chip-level check of the same loops: 2048 workgroups x 16 waves, wall time from HIP events -> wave64 instructions per
    // clock per CU (256 CUs; clock taken as 2.4 GHz)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
#define CHIP(V, NAME) { hipLaunchKernelGGL(k<V>, dim3(2048), dim3(1024), 0, 0, out, cyc, 0.37f); (void)hipEventRecord(e0, 0); \
        hipLaunchKernelGGL(k<V>, dim3(2048), dim3(1024), 0, 0, out, cyc, 0.37f); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); \
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); const double instr = 2048.0 * 16 * REP * 64; \
        printf("{\"chip\": \"%s\", \"ms\": %.4f, \"wave64_instr_per_us\": %.1f, \"wave64_instr_per_clk_per_CU_at_2.4GHz\": %.3f}\n", NAME, ms, instr / ms / 1e3, instr / (ms * 1e-3) / 2.4e9 / 256); }
    CHIP(0, "v_fma_f32") CHIP(1, "v_pk_fma_f32") CHIP(8, "v_add_f32") CHIP(5, "v_mov_b32_dpp")
    return 0;
}
