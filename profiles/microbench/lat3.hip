// lat3.hip -- round-2 VALU issue / latency microbenchmark for gfx950, second take.
// lat2.hip (and round 1's valu_rate.hip) put every instruction in its own `asm volatile`; the compiler's hazard recogniser
// then drops an `s_nop 0` between two asm statements that touch the same register, so their "dependent chain" numbers
// contained one s_nop per link.  Here every measured block is ONE asm statement (nothing can be inserted inside), and the
// wait states the ISA really needs (VALU write -> DPP read: s_nop 1; trans result -> VALU: s_nop 0) are written out.
// Also calibrates s_memtime ticks against wall time (HIP events) for 1 workgroup, 256 and 2048 workgroups.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o lat3 lat3.hip && ./lat3
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

#define R2(x) x x
#define R4(x) R2(R2(x))
#define R8(x) R2(R4(x))
#define R16(x) R2(R8(x))
#define R32(x) R2(R16(x))
#define R64(x) R2(R32(x))

#define REP 64
#define CALREP 2048

// operands: %0 d (float, chain), %1 p (f2, chain), %2..%9 s0..s7 (float accumulators), %10..%13 a0..a3 (f2 accumulators),
//           %14 x, %15 y (float), %16 q (f2)
#define OPS                                                                                                     \
    : "+v"(d), "+v"(p), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7), "+v"(a0), \
      "+v"(a1), "+v"(a2), "+v"(a3)                                                                              \
    : "v"(x), "v"(y), "v"(q)                                                                                    \
    : "vcc", "scc", "s20"

// name, instructions per unit (for the report), 64 units per block
#define CASES                                                                                                   \
    X(0, "chain v_add_f32 (VOP2)", 1, R64("v_add_f32 %0, %0, %15\n"))                                           \
    X(1, "chain v_mul_f32 (VOP2)", 1, R64("v_mul_f32 %0, %0, %14\n"))                                           \
    X(2, "chain v_fmac_f32 (VOP2, accumulator)", 1, R64("v_fmac_f32 %0, %14, %15\n"))                           \
    X(3, "chain v_fma_f32 (VOP3, src0)", 1, R64("v_fma_f32 %0, %0, %14, %15\n"))                                \
    X(4, "chain v_fma_f32 (VOP3, src2)", 1, R64("v_fma_f32 %0, %14, %15, %0\n"))                                \
    X(5, "chain v_fmaak_f32", 1, R64("v_fmaak_f32 %0, %0, %14, 0x3e800000\n"))                                  \
    X(6, "chain v_xor_b32", 1, R64("v_xor_b32 %0, %0, %15\n"))                                                  \
    X(7, "chain v_rndne_f32", 1, R64("v_rndne_f32 %0, %0\n"))                                                   \
    X(8, "chain v_med3_f32", 1, R64("v_med3_f32 %0, %0, %14, %15\n"))                                           \
    X(9, "chain v_pk_mul_f32", 1, R64("v_pk_mul_f32 %1, %1, %16\n"))                                            \
    X(10, "chain v_pk_add_f32", 1, R64("v_pk_add_f32 %1, %1, %16\n"))                                           \
    X(11, "chain v_pk_fma_f32 (src0)", 1, R64("v_pk_fma_f32 %1, %1, %16, %16\n"))                               \
    X(12, "chain v_pk_fma_f32 (src2)", 1, R64("v_pk_fma_f32 %1, %16, %16, %1\n"))                               \
    X(13, "chain v_cndmask_b32 (vcc fixed)", 1, R64("v_cndmask_b32 %0, %0, %14, vcc\n"))                        \
    X(14, "chain v_cmp_gt_f32 -> v_cndmask_b32 (2 instr)", 2, R64("v_cmp_gt_f32 vcc, %0, %15\n v_cndmask_b32 %0, %0, %14, vcc\n")) \
    X(15, "chain v_bfi_b32", 1, R64("v_bfi_b32 %0, %15, %0, %14\n"))                                            \
    X(16, "chain v_floor_f32", 1, R64("v_floor_f32 %0, %0\n"))                                                  \
    X(17, "chain v_cvt_i32_f32 -> v_cvt_f32_i32 (2 instr)", 2, R64("v_cvt_i32_f32 %0, %0\n v_cvt_f32_i32 %0, %0\n")) \
    X(18, "chain v_sqrt_f32 ; s_nop 0 ; v_add_f32 (2 VALU)", 2, R64("v_sqrt_f32 %0, %0\n s_nop 0\n v_add_f32 %0, %0, %14\n")) \
    X(19, "chain s_nop 1 ; v_mov_b32_dpp d,d row_shr:1", 1, R64("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")) \
    X(20, "chain v_add ; s_nop 1 ; v_mov_dpp ; (2 VALU)", 2, R64("v_add_f32 %2, %0, %15\n s_nop 1\n v_mov_b32_dpp %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n")) \
    X(21, "chain v_max_f32 with |abs| (VOP3)", 1, R64("v_max_f32 %0, |%0|, |%14|\n"))                           \
    X(22, "indep v_fmac_f32 x8 accumulators (VOP2)", 8, R8("v_fmac_f32 %2, %14, %15\n v_fmac_f32 %3, %14, %15\n v_fmac_f32 %4, %14, %15\n v_fmac_f32 %5, %14, %15\n v_fmac_f32 %6, %14, %15\n v_fmac_f32 %7, %14, %15\n v_fmac_f32 %8, %14, %15\n v_fmac_f32 %9, %14, %15\n")) \
    X(23, "indep v_fma_f32 x8 accumulators (VOP3)", 8, R8("v_fma_f32 %2, %14, %15, %2\n v_fma_f32 %3, %14, %15, %3\n v_fma_f32 %4, %14, %15, %4\n v_fma_f32 %5, %14, %15, %5\n v_fma_f32 %6, %14, %15, %6\n v_fma_f32 %7, %14, %15, %7\n v_fma_f32 %8, %14, %15, %8\n v_fma_f32 %9, %14, %15, %9\n")) \
    X(24, "indep v_pk_fma_f32 x4 accumulators", 4, R16("v_pk_fma_f32 %10, %16, %16, %10\n v_pk_fma_f32 %11, %16, %16, %11\n v_pk_fma_f32 %12, %16, %16, %12\n v_pk_fma_f32 %13, %16, %16, %13\n")) \
    X(25, "indep v_mov_b32_dpp x4 (source untouched)", 4, R16("v_mov_b32_dpp %2, %14 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %14 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %14 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %14 row_shr:2 row_mask:0xf bank_mask:0xf\n")) \
    X(26, "link v_add + 1 indep v_fmac", 2, R64("v_add_f32 %0, %0, %15\n v_fmac_f32 %2, %14, %15\n"))             \
    X(27, "link v_add + 2 indep v_fmac", 3, R64("v_add_f32 %0, %0, %15\n v_fmac_f32 %2, %14, %15\n v_fmac_f32 %3, %14, %15\n")) \
    X(28, "link v_add + 3 indep v_fmac", 4, R64("v_add_f32 %0, %0, %15\n v_fmac_f32 %2, %14, %15\n v_fmac_f32 %3, %14, %15\n v_fmac_f32 %4, %14, %15\n")) \
    X(29, "link v_add + 1 indep v_pk_fma", 2, R64("v_add_f32 %0, %0, %15\n v_pk_fma_f32 %10, %16, %16, %10\n")) \
    X(30, "link v_add + 2 indep v_pk_fma", 3, R64("v_add_f32 %0, %0, %15\n v_pk_fma_f32 %10, %16, %16, %10\n v_pk_fma_f32 %11, %16, %16, %11\n")) \
    X(31, "link v_fma(VOP3) + 1 indep v_fma(VOP3)", 2, R64("v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %14, %15, %2\n")) \
    X(32, "two interleaved chains v_add", 2, R64("v_add_f32 %0, %0, %15\n v_add_f32 %2, %2, %15\n"))              \
    X(33, "three interleaved chains v_add", 3, R64("v_add_f32 %0, %0, %15\n v_add_f32 %2, %2, %15\n v_add_f32 %3, %3, %15\n")) \
    X(34, "indep v_fmac + s_nop 0 after each", 1, R8("v_fmac_f32 %2, %14, %15\n s_nop 0\n v_fmac_f32 %3, %14, %15\n s_nop 0\n v_fmac_f32 %4, %14, %15\n s_nop 0\n v_fmac_f32 %5, %14, %15\n s_nop 0\n v_fmac_f32 %6, %14, %15\n s_nop 0\n v_fmac_f32 %7, %14, %15\n s_nop 0\n v_fmac_f32 %8, %14, %15\n s_nop 0\n v_fmac_f32 %9, %14, %15\n s_nop 0\n")) \
    X(35, "indep v_fmac + s_mov_b32 after each", 1, R8("v_fmac_f32 %2, %14, %15\n s_mov_b32 s20, 1\n v_fmac_f32 %3, %14, %15\n s_mov_b32 s20, 2\n v_fmac_f32 %4, %14, %15\n s_mov_b32 s20, 3\n v_fmac_f32 %5, %14, %15\n s_mov_b32 s20, 4\n v_fmac_f32 %6, %14, %15\n s_mov_b32 s20, 5\n v_fmac_f32 %7, %14, %15\n s_mov_b32 s20, 6\n v_fmac_f32 %8, %14, %15\n s_mov_b32 s20, 7\n v_fmac_f32 %9, %14, %15\n s_mov_b32 s20, 8\n")) \
    X(36, "chain v_add + s_mov_b32 after each", 1, R64("v_add_f32 %0, %0, %15\n s_mov_b32 s20, 1\n"))            \
    X(37, "chain v_add ; v_readfirstlane ; s_cmp ; (loop-carried scalar test)", 2, R64("v_add_f32 %0, %0, %15\n v_readfirstlane_b32 s20, %0\n s_cmp_eq_u32 s20, 0\n")) \
    X(38, "chain v_mul -> v_add alternating (VOP2)", 2, R64("v_mul_f32 %0, %0, %14\n v_add_f32 %0, %0, %15\n")) \
    X(39, "chain v_mul_f32 e64 with neg (VOP3)", 1, R64("v_mul_f32 %0, -%0, %14\n"))

template <int V> __global__ __launch_bounds__(1024) void k_case(long long* cyc, float* out, float seed, int lanes, int reps) {
    float d = seed, x = 0.999f + seed * 1e-9f, y = 1e-9f * seed;
    f2 p = f2{ seed, -seed }, q = f2{ 0.999f, 1.001f };
    float s0 = seed, s1 = seed * 2, s2 = seed * 3, s3 = seed * 4, s4 = seed * 5, s5 = seed * 6, s6 = seed * 7, s7 = seed * 8;
    f2 a0 = f2{ seed, 1 }, a1 = f2{ seed, 2 }, a2 = f2{ seed, 3 }, a3 = f2{ seed, 4 };
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(x), "v"(y) : "vcc");
    long long t0 = 0, t1 = 0;
    if ((int)(threadIdx.x & 63) < lanes) {
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int r = 0; r < reps; r++) {
#define X(N, NAME, NI, ASM) if (V == N) asm volatile(ASM OPS);
            CASES
#undef X
        }
        t1 = __builtin_readcyclecounter();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = d + p.x + p.y + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + a0.x + a1.x + a2.x + a3.x + a0.y + a1.y + a2.y + a3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// two waves on one SIMD: wave 0 = dependent v_add chain (timed), wave 4 = filler stream (4x the count), waves 1..3 exit
template <int FILL, bool PRIO> __global__ void k_prio(long long* cyc, float* out, float seed) {
    const int wave = threadIdx.x >> 6;
    float d = seed, x = 0.999f, y = 1e-9f * seed;
    f2 p = f2{ seed, -seed }, q = f2{ 0.999f, 1.001f };
    float s0 = seed, s1 = seed * 2, s2 = seed * 3, s3 = seed * 4, s4 = seed * 5, s5 = seed * 6, s6 = seed * 7, s7 = seed * 8;
    f2 a0 = f2{ seed, 1 }, a1 = f2{ seed, 2 }, a2 = f2{ seed, 3 }, a3 = f2{ seed, 4 };
    if (wave == 0) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int r = 0; r < REP; r++) asm volatile(R64("v_add_f32 %0, %0, %15\n") OPS);
        long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
    } else if (wave == 4 && FILL != 0) {
        long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int r = 0; r < 4 * REP; r++) {
            if (FILL == 1) asm volatile(R8("v_fmac_f32 %2, %14, %15\n v_fmac_f32 %3, %14, %15\n v_fmac_f32 %4, %14, %15\n v_fmac_f32 %5, %14, %15\n v_fmac_f32 %6, %14, %15\n v_fmac_f32 %7, %14, %15\n v_fmac_f32 %8, %14, %15\n v_fmac_f32 %9, %14, %15\n") OPS);
            if (FILL == 2) asm volatile(R16("v_pk_fma_f32 %10, %16, %16, %10\n v_pk_fma_f32 %11, %16, %16, %11\n v_pk_fma_f32 %12, %16, %16, %12\n v_pk_fma_f32 %13, %16, %16, %13\n") OPS);
        }
        long long t1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) cyc[1] = t1 - t0;
    }
    out[threadIdx.x] = d + p.x + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + a0.x + a1.x + a2.x + a3.x;
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 2048 * 1024 * sizeof(float));
    (void)hipMallocManaged(&cyc, 8 * sizeof(long long));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("== one wave alone (64 threads): s_memtime ticks per instruction\n");
#define X(N, NAME, NI, ASM) {                                                                   \
        for (int rep = 0; rep < 2; rep++) { cyc[0] = 0; hipLaunchKernelGGL(k_case<N>, dim3(1), dim3(64), 0, 0, cyc, out, 0.37f, 64, REP); (void)hipDeviceSynchronize(); } \
        const double t1 = (double)cyc[0] / (REP * 64.0);                                         \
        for (int rep = 0; rep < 2; rep++) { cyc[0] = 0; hipLaunchKernelGGL(k_case<N>, dim3(1), dim3(256), 0, 0, cyc, out, 0.37f, 64, REP); (void)hipDeviceSynchronize(); } \
        const double t4 = (double)cyc[0] / (REP * 64.0);                                         \
        for (int rep = 0; rep < 2; rep++) { cyc[0] = 0; hipLaunchKernelGGL(k_case<N>, dim3(1), dim3(512), 0, 0, cyc, out, 0.37f, 64, REP); (void)hipDeviceSynchronize(); } \
        const double t8 = (double)cyc[0] / (REP * 64.0);                                         \
        for (int rep = 0; rep < 2; rep++) { cyc[0] = 0; hipLaunchKernelGGL(k_case<N>, dim3(1), dim3(1024), 0, 0, cyc, out, 0.37f, 64, REP); (void)hipDeviceSynchronize(); } \
        const double t16 = (double)cyc[0] / (REP * 64.0);                                        \
        printf("{\"case\": \"%s\", \"valu_per_unit\": %d, \"ticks_per_unit_alone\": %.2f, \"one_wave_per_simd\": %.2f, \"two_per_simd\": %.2f, \"four_per_simd\": %.2f}\n", NAME, NI, t1, t4, t8, t16); }
    CASES
#undef X
    printf("== partially filled waves, 4 waves per SIMD: ticks per instruction per wave\n");
    for (int lanes : { 64, 32, 16, 8, 1 }) {
        cyc[0] = 0; hipLaunchKernelGGL(k_case<22>, dim3(1), dim3(1024), 0, 0, cyc, out, 0.37f, lanes, REP); (void)hipDeviceSynchronize();
        const double a = (double)cyc[0] / (REP * 64.0);
        cyc[0] = 0; hipLaunchKernelGGL(k_case<24>, dim3(1), dim3(1024), 0, 0, cyc, out, 0.37f, lanes, REP); (void)hipDeviceSynchronize();
        const double b = (double)cyc[0] / (REP * 64.0);
        printf("{\"lanes\": %d, \"v_fmac_ticks_per_instr_per_wave\": %.2f, \"v_pk_fma_ticks_per_instr_per_wave\": %.2f}\n", lanes, a, b);
    }
    printf("== dependent v_add chain (wave 0) vs filler (wave 4, same SIMD)\n");
#define PRIORUN(FILL, PRIO, NAME) { for (int rep = 0; rep < 2; rep++) { cyc[0] = cyc[1] = 0; hipLaunchKernelGGL((k_prio<FILL, PRIO>), dim3(1), dim3(320), 0, 0, cyc, out, 0.37f); (void)hipDeviceSynchronize(); } \
        printf("{\"prio\": \"%s\", \"setprio\": %d, \"chain_ticks_per_link\": %.2f, \"filler_ticks_per_instr\": %.2f}\n", NAME, (int)PRIO, cyc[0] / (REP * 64.0), cyc[1] / (4.0 * REP * 64.0)); }
    PRIORUN(0, false, "no filler") PRIORUN(1, false, "v_fmac filler") PRIORUN(1, true, "v_fmac filler") PRIORUN(2, false, "v_pk_fma filler") PRIORUN(2, true, "v_pk_fma filler")
    printf("== tick calibration: independent v_fmac / v_pk_fma stream, 16 waves per workgroup; ticks of workgroup 0 vs wall time\n");
    for (int grid : { 1, 256, 2048 }) {
        for (int pk = 0; pk < 2; pk++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                cyc[0] = 0;
                (void)hipEventRecord(e0, 0);
                for (int it = 0; it < 4; it++) {
                    if (pk) hipLaunchKernelGGL(k_case<24>, dim3(grid), dim3(1024), 0, 0, cyc, out, 0.37f, 64, CALREP);
                    else hipLaunchKernelGGL(k_case<22>, dim3(grid), dim3(1024), 0, 0, cyc, out, 0.37f, 64, CALREP);
                }
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double instr_per_wave = CALREP * 64.0;
            printf("{\"grid\": %d, \"stream\": \"%s\", \"ticks_per_instr_per_wave\": %.2f, \"wall_us_per_launch\": %.2f, \"wave_instr_per_us_per_CU\": %.1f}\n", grid,
                   pk ? "v_pk_fma_f32" : "v_fmac_f32", cyc[0] / instr_per_wave, ms * 1e3 / 4, (double)grid * 16 * instr_per_wave / (ms * 1e3 / 4) / (grid < 256 ? grid : 256));
        }
    }
    return 0;
}
