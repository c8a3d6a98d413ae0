// lat4.hip -- how many instructions per clock ONE SIMD issues when it hosts 1, 2, 3 or 4 wavefronts (gfx950).
// lat3.hip timed only the oldest wave of a SIMD, which the arbiter serves first; here every wave of the workgroup runs the
// same stream, each records its start and end (s_memtime), and the host reports the SIMD-level cost
//     (last end - first start) / (instructions per wave x waves on that SIMD).
// Workgroup = 4 x W waves (W per SIMD).  Streams: plain VOP2, packed FP32, DPP, a mix shaped like the FLL loop wave, and
// a dependent chain next to an independent stream.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o lat4 lat4.hip && ./lat4
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

#define OPS                                                                                                     \
    : "+v"(d), "+v"(p), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7), "+v"(a0), \
      "+v"(a1), "+v"(a2), "+v"(a3)                                                                              \
    : "v"(x), "v"(y), "v"(q), "v"(addr)                                                                          \
    : "vcc", "scc", "s20", "memory"

// 64 instructions per block in every case
#define CASES                                                                                                   \
    X(0, "VOP2 independent (v_fmac x8 accumulators)", R8("v_fmac_f32 %2, %14, %15\n v_fmac_f32 %3, %14, %15\n v_fmac_f32 %4, %14, %15\n v_fmac_f32 %5, %14, %15\n v_fmac_f32 %6, %14, %15\n v_fmac_f32 %7, %14, %15\n v_fmac_f32 %8, %14, %15\n v_fmac_f32 %9, %14, %15\n")) \
    X(1, "VOP2 dependent chain (v_add)", R64("v_add_f32 %0, %0, %15\n"))                                          \
    X(2, "packed FP32 independent (v_pk_fma x4)", R16("v_pk_fma_f32 %10, %16, %16, %10\n v_pk_fma_f32 %11, %16, %16, %11\n v_pk_fma_f32 %12, %16, %16, %12\n v_pk_fma_f32 %13, %16, %16, %13\n")) \
    X(3, "DPP moves independent", R16("v_mov_b32_dpp %2, %14 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %14 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %14 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %14 row_shr:2 row_mask:0xf bank_mask:0xf\n")) \
    X(4, "mix like the FLL loop wave (per 16: 9 VOP2, 3 pk, 2 dpp, 1 ds_read_b64, 1 s_nop)", R4("v_fmac_f32 %2, %14, %15\n v_fmac_f32 %3, %14, %15\n v_pk_fma_f32 %10, %16, %16, %10\n v_fmac_f32 %4, %14, %15\n v_mov_b32_dpp %5, %14 row_shr:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %6, %14, %15\n v_pk_fma_f32 %11, %16, %16, %11\n v_fmac_f32 %7, %14, %15\n ds_read_b64 %1, %17\n v_fmac_f32 %8, %14, %15\n v_mov_b32_dpp %9, %14 row_shr:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %2, %14, %15\n v_pk_fma_f32 %12, %16, %16, %12\n v_fmac_f32 %3, %14, %15\n s_nop 0\n v_fmac_f32 %4, %14, %15\n")) \
    X(5, "SALU only (s_add_u32)", R64("s_add_u32 s20, s20, 1\n"))                                                \
    X(6, "VOP2 + SALU alternating", R32("v_fmac_f32 %2, %14, %15\n s_add_u32 s20, s20, 1\n"))                    \
    X(7, "LDS reads only (ds_read_b64, waitcnt every 16)", R4("ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n ds_read_b64 %1, %17\n s_waitcnt lgkmcnt(0)\n")) \
    X(8, "VOP3 (v_fma_f32) independent", R8("v_fma_f32 %2, %14, %15, %2\n v_fma_f32 %3, %14, %15, %3\n v_fma_f32 %4, %14, %15, %4\n v_fma_f32 %5, %14, %15, %5\n v_fma_f32 %6, %14, %15, %6\n v_fma_f32 %7, %14, %15, %7\n v_fma_f32 %8, %14, %15, %8\n v_fma_f32 %9, %14, %15, %9\n"))

template <int V> __global__ __launch_bounds__(1024) void k_case(long long* t_start, long long* t_end, float* out, float seed, int reps) {
    __shared__ f2 buf[1024];
    buf[threadIdx.x] = f2{ seed, seed };
    float d = seed, x = 0.999f + seed * 1e-9f, y = 1e-9f * seed;
    f2 p = f2{ seed, -seed }, q = f2{ 0.999f, 1.001f };
    float s0 = seed, s1 = seed * 2, s2 = seed * 3, s3 = seed * 4, s4 = seed * 5, s5 = seed * 6, s6 = seed * 7, s7 = seed * 8;
    f2 a0 = f2{ seed, 1 }, a1 = f2{ seed, 2 }, a2 = f2{ seed, 3 }, a3 = f2{ seed, 4 };
    const unsigned addr = (unsigned)(size_t)(&buf[0]) + threadIdx.x * 8;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#define X(N, NAME, ASM) if (V == N) asm volatile(ASM OPS);
        CASES
#undef X
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = d + p.x + p.y + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + a0.x + a1.x + a2.x + a3.x + a0.y + a1.y + a2.y + a3.y;
    if ((threadIdx.x & 63) == 0) { t_start[threadIdx.x >> 6] = t0; t_end[threadIdx.x >> 6] = t1; }
}

int main() {
    float* out; long long *ts, *te;
    (void)hipMalloc(&out, 1024 * sizeof(float));
    (void)hipMallocManaged(&ts, 16 * sizeof(long long));
    (void)hipMallocManaged(&te, 16 * sizeof(long long));
    printf("== clocks per instruction PER SIMD with W waves on each SIMD (all waves run the same stream; first start -> last end)\n");
#define X(N, NAME, ASM) {                                                                      \
        double r[5] = { 0 };                                                                     \
        for (int W = 1; W <= 4; W++) {                                                           \
            for (int rep = 0; rep < 2; rep++) {                                                  \
                hipLaunchKernelGGL(k_case<N>, dim3(1), dim3(256 * W), 0, 0, ts, te, out, 0.37f, REP); \
                (void)hipDeviceSynchronize();                                                    \
            }                                                                                    \
            long long a = ts[0], b = te[0];                                                      \
            for (int w = 0; w < 4 * W; w++) { if (ts[w] < a) a = ts[w]; if (te[w] > b) b = te[w]; } \
            r[W] = (double)(b - a) / (REP * 64.0 * W);                                           \
        }                                                                                        \
        printf("{\"stream\": \"%s\", \"W1\": %.2f, \"W2\": %.2f, \"W3\": %.2f, \"W4\": %.2f}\n", NAME, r[1], r[2], r[3], r[4]); }
    CASES
#undef X
    return 0;
}
