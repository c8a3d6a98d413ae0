// lat2.hip -- round-2 microbenchmarks for the k_fused restructuring (gfx950): dependent-issue latency per instruction
// class, how many independent instructions hide between two links of a dependent chain, VALU/SALU/LDS co-issue from one
// wave, s_setprio between two waves that share a SIMD, ALU cost of partially filled waves, LDS flag ping-pong between
// two waves (the hand-over latency of a barrier-free FLL helper wave), dependent LDS read latency, s_barrier cost.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o lat2 lat2.hip && ./lat2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP 64

// ---------------------------------------------------------------------------------------------------------------
// A. one wave, one dependent chain of 64*REP instructions of one kind
// ---------------------------------------------------------------------------------------------------------------
#define CHAIN_CASES                                                                                            \
    X(0, "v_add_f32 (VOP2)", "v_add_f32 %0, %0, %2")                                                            \
    X(1, "v_mul_f32 (VOP2)", "v_mul_f32 %0, %0, %2")                                                            \
    X(2, "v_fmac_f32 (VOP2) through the accumulator", "v_fmac_f32 %0, %2, %3")                                  \
    X(3, "v_fma_f32 (VOP3) through src0", "v_fma_f32 %0, %0, %2, %3")                                           \
    X(4, "v_fmaak_f32 through src0", "v_fmaak_f32 %0, %0, %2, 0x3e800000")                                      \
    X(5, "v_xor_b32", "v_xor_b32 %0, %0, %2")                                                                   \
    X(6, "v_rndne_f32", "v_rndne_f32 %0, %0")                                                                   \
    X(7, "v_max_f32 (VOP2)", "v_max_f32 %0, %0, %2")                                                            \
    X(8, "v_med3_f32", "v_med3_f32 %0, %0, %2, %3")                                                             \
    X(9, "v_mov_b32_dpp row_shr:1 (dst = dpp(dst))", "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(10, "v_pk_mul_f32", "v_pk_mul_f32 %1, %1, %4")                                                            \
    X(11, "v_pk_add_f32", "v_pk_add_f32 %1, %1, %4")                                                            \
    X(12, "v_pk_fma_f32 through src0", "v_pk_fma_f32 %1, %1, %4, %4")                                           \
    X(13, "v_cndmask_b32 (vcc fixed)", "v_cndmask_b32 %0, %0, %2, vcc")                                         \
    X(14, "v_cmp_gt_f32 + v_cndmask_b32 pair (per pair)", "v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %3, vcc") \
    X(15, "v_mul_f32 with |abs| modifier (VOP3)", "v_mul_f32 %0, |%0|, %2")                                     \
    X(16, "v_sub_f32 (VOP2)", "v_sub_f32 %0, %0, %2")                                                           \
    X(17, "v_bfi_b32 (copysign)", "v_bfi_b32 %0, %2, %0, %3")                                                   \
    X(18, "v_floor_f32", "v_floor_f32 %0, %0")                                                                  \
    X(19, "v_cvt_i32_f32 + v_cvt_f32_i32 pair (per pair)", "v_cvt_i32_f32 %0, %0\n v_cvt_f32_i32 %0, %0")       \
    X(20, "v_sqrt_f32", "v_sqrt_f32 %0, %0")                                                                    \
    X(21, "v_fma_f32 (VOP3) through src2 (accumulator)", "v_fma_f32 %0, %2, %3, %0")                            \
    X(22, "v_pk_fma_f32 through src2", "v_pk_fma_f32 %1, %4, %4, %1")

template <int V> __global__ void k_chain(long long* cyc, float* out, float seed) {
    float d = seed, x = 0.999f + seed * 1e-9f, y = 1e-9f * seed;
    f2 p = f2{ seed, -seed }, q = f2{ 0.999f, 1.001f };
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(x), "v"(y) : "vcc");
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 64; i++) {
#define X(N, NAME, ASM) if (V == N) asm volatile(ASM : "+v"(d), "+v"(p) : "v"(x), "v"(y), "v"(q) : "vcc");
            CHAIN_CASES
#undef X
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = d + p.x + p.y;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------
// B. one wave: a dependent chain with K independent instructions between consecutive links.
//    KIND 0: links v_fma_f32, fillers v_fma_f32;  1: links v_add_f32 (VOP2), fillers v_fmac_f32 (VOP2);
//    2: links v_fma_f32, fillers v_pk_fma_f32;  3: links v_fma, fillers v_mov_dpp
// ---------------------------------------------------------------------------------------------------------------
template <int KIND, int K> __global__ void k_fill(long long* cyc, float* out, float seed) {
    float d = seed, x = 0.999f, y = 1e-9f * seed;
    float s[8];
    f2 a[8];
    f2 q = f2{ 0.999f, 1.001f };
    for (int i = 0; i < 8; i++) { s[i] = seed * i; a[i] = f2{ seed + i, seed - i }; }
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 64; i++) {
            if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(y));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(x), "v"(y));
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int j = (i * K + k) & 7;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[j]) : "v"(x), "v"(y));
                if (KIND == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[j]) : "v"(x), "v"(y));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[j]) : "v"(q));
                if (KIND == 3) asm volatile("v_mov_b32_dpp %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(s[j]) : "v"(x));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = d;
    for (int i = 0; i < 8; i++) acc += s[i] + a[i].x + a[i].y;
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------
// C. one wave: independent v_fma_f32 stream with another instruction class after every VALU instruction
//    KIND 0: nothing; 1: s_add_u32; 2: s_nop 0; 3: ds_read_b64 (no wait; waitcnt every 8); 4: ds_write_b64; 5: v_nop
// ---------------------------------------------------------------------------------------------------------------
template <int KIND> __global__ void k_mix(long long* cyc, float* out, float seed) {
    __shared__ f2 buf[1024];
    buf[threadIdx.x] = f2{ seed, seed };
    __syncthreads();
    float x = 0.999f, y = 1e-9f * seed;
    float s[8];
    f2 ld[8];
    for (int i = 0; i < 8; i++) { s[i] = seed * i; ld[i] = f2{ 0, 0 }; }
    unsigned sa = 0;
    const unsigned addr = (unsigned)(size_t)(&buf[0]) + threadIdx.x * 8;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 64; i++) {
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i & 7]) : "v"(x), "v"(y));
            if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sa));
            if (KIND == 2) asm volatile("s_nop 0");
            if (KIND == 3) {
                asm volatile("ds_read_b64 %0, %1" : "=v"(ld[i & 7]) : "v"(addr));
                if ((i & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)");
            }
            if (KIND == 4) {
                asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(ld[i & 7]));
                if ((i & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)");
            }
            if (KIND == 5) asm volatile("v_nop");
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = (float)sa;
    for (int i = 0; i < 8; i++) acc += s[i] + ld[i].x;
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------
// D. two waves on one SIMD (waves 0 and 4 of a 320-thread workgroup; waves 1..3 exit): wave 0 runs a dependent
//    v_fma chain (timed), wave 4 an independent filler stream.  PRIO: wave 0 raises its priority to 3.
//    FILL 0 none, 1 v_fma_f32, 2 v_pk_fma_f32, 3 dpp
// ---------------------------------------------------------------------------------------------------------------
template <int FILL, bool PRIO> __global__ void k_prio(long long* cyc, float* out, float seed) {
    const int wave = threadIdx.x >> 6;
    float x = 0.999f, y = 1e-9f * seed;
    if (wave == 0) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        float d = seed;
        long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(x), "v"(y));
        }
        long long t1 = __builtin_readcyclecounter();
        out[threadIdx.x] = d;
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
    } else if (wave == 4 && FILL != 0) {
        float s[8];
        f2 a[8];
        f2 q = f2{ 0.999f, 1.001f };
        for (int i = 0; i < 8; i++) { s[i] = seed * i; a[i] = f2{ seed + i, seed - i }; }
        long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int r = 0; r < 4 * REP; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) {
                if (FILL == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i & 7]) : "v"(x), "v"(y));
                if (FILL == 2) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i & 7]) : "v"(q));
                if (FILL == 3) asm volatile("v_mov_b32_dpp %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(s[i & 7]) : "v"(x));
            }
        }
        long long t1 = __builtin_readcyclecounter();
        float acc = 0;
        for (int i = 0; i < 8; i++) acc += s[i] + a[i].x + a[i].y;
        out[threadIdx.x] = acc;
        if ((threadIdx.x & 63) == 0) cyc[1] = t1 - t0;   // filler: 4x the instruction count
    }
}

// ---------------------------------------------------------------------------------------------------------------
// E. ALU cost of partially filled waves: 16 waves (4 per SIMD), independent v_fma_f32 / v_pk_fma_f32 stream,
//    only the first LANES lanes of every wave enabled.
// ---------------------------------------------------------------------------------------------------------------
template <int LANES, bool PK> __global__ __launch_bounds__(1024) void k_exec(long long* cyc, float* out, float seed) {
    float x = 0.999f, y = 1e-9f * seed;
    float s[8];
    f2 a[8];
    f2 q = f2{ 0.999f, 1.001f };
    for (int i = 0; i < 8; i++) { s[i] = seed * i; a[i] = f2{ seed + i, seed - i }; }
    long long t0 = 0, t1 = 0;
    if ((threadIdx.x & 63) < LANES) {
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 64; i++) {
                if (!PK) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i & 7]) : "v"(x), "v"(y));
                else asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i & 7]) : "v"(q));
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += s[i] + a[i].x + a[i].y;
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------
// F. LDS flag ping-pong between wave 0 (SIMD 0) and wave 1 (SIMD 1): clocks per round trip (two hand-overs).
//    PAYLOAD: each hand-over also moves a 16-byte record per lane written before the flag and read after it.
// ---------------------------------------------------------------------------------------------------------------
template <bool PAYLOAD> __global__ void k_pingpong(long long* cyc, float* out, int n) {
    __shared__ volatile int flag_a, flag_b;
    __shared__ f4 rec_a[64], rec_b[64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) { flag_a = 0; flag_b = 0; }
    __syncthreads();
    f4 v = f4{ (float)lane, 1, 2, 3 };
    if (wave == 0) {
        long long t0 = __builtin_readcyclecounter();
        for (int i = 1; i <= n; i++) {
            if (PAYLOAD) { ((volatile f4*)rec_a)[lane] = v; }
            if (lane == 0) flag_a = i;
            while (flag_b != i) {}
            if (PAYLOAD) { v = ((volatile f4*)rec_b)[lane]; v.x += 1.0f; }
        }
        long long t1 = __builtin_readcyclecounter();
        if (lane == 0) cyc[0] = t1 - t0;
    } else {
        for (int i = 1; i <= n; i++) {
            while (flag_a != i) {}
            if (PAYLOAD) { v = ((volatile f4*)rec_a)[lane]; v.y += 1.0f; ((volatile f4*)rec_b)[lane] = v; }
            if (lane == 0) flag_b = i;
        }
    }
    out[threadIdx.x] = v.x + v.y;
}

// ---------------------------------------------------------------------------------------------------------------
// G. dependent LDS reads from one wave: address = value read.  WIDTH 32 / 64 / 128 bits.
// ---------------------------------------------------------------------------------------------------------------
template <int WIDTH> __global__ void k_ldslat(long long* cyc, float* out, int n) {
    __shared__ __attribute__((aligned(16))) unsigned tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) tab[i] = (unsigned)(size_t)(&tab[0]) + (((i & ~3) * 4 + 272) & 16383);
    __syncthreads();
    unsigned a = (unsigned)(size_t)(&tab[0]) + threadIdx.x * 16;
    unsigned b1 = 0, b2 = 0, b3 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        if (WIDTH == 32) asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a));
        if (WIDTH == 64) {
            unsigned long long r;
            asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a));
            a = (unsigned)r; b1 += (unsigned)(r >> 32);
        }
        if (WIDTH == 128) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 r;
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a));
            a = r.x; b1 += r.y; b2 += r.z; b3 += r.w;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = (float)(a + b1 + b2 + b3);
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// H. s_barrier with 6 waves
__global__ void k_barrier(long long* cyc, float* out, int n) {
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = 0;
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4096 * sizeof(float));
    (void)hipMallocManaged(&cyc, 8 * sizeof(long long));
    const double NI = REP * 64.0;
#define LAUNCH(KERNEL, THREADS, ...)                                                   \
    for (int rep_ = 0; rep_ < 2; rep_++) {                                             \
        cyc[0] = cyc[1] = 0;                                                           \
        hipLaunchKernelGGL(KERNEL, dim3(1), dim3(THREADS), 0, 0, cyc, out, __VA_ARGS__); \
        (void)hipDeviceSynchronize();                                                  \
    }
    printf("== A. dependent chains, one wave alone: clocks per instruction\n");
#define X(N, NAME, ASM) { LAUNCH(k_chain<N>, 64, 0.37f) printf("{\"chain\": \"%s\", \"clk\": %.2f}\n", NAME, cyc[0] / NI); }
    CHAIN_CASES
#undef X
    printf("== B. dependent chain + K independent fillers per link, one wave alone: clocks per link\n");
#define FILLRUN(KIND, K, NAME) { LAUNCH((k_fill<KIND, K>), 64, 0.37f) printf("{\"fill\": \"%s\", \"k\": %d, \"clk_per_link\": %.2f}\n", NAME, K, cyc[0] / NI); }
    FILLRUN(0, 0, "v_fma links + v_fma fillers") FILLRUN(0, 1, "v_fma links + v_fma fillers") FILLRUN(0, 2, "v_fma links + v_fma fillers") FILLRUN(0, 3, "v_fma links + v_fma fillers")
    FILLRUN(1, 0, "v_add links + v_fmac fillers") FILLRUN(1, 1, "v_add links + v_fmac fillers") FILLRUN(1, 2, "v_add links + v_fmac fillers") FILLRUN(1, 3, "v_add links + v_fmac fillers")
    FILLRUN(2, 1, "v_fma links + v_pk_fma fillers") FILLRUN(2, 2, "v_fma links + v_pk_fma fillers")
    FILLRUN(3, 1, "v_fma links + dpp fillers") FILLRUN(3, 2, "v_fma links + dpp fillers")
    printf("== C. independent v_fma stream + one other instruction after each, one wave alone: clocks per pair\n");
#define MIXRUN(KIND, NAME) { LAUNCH(k_mix<KIND>, 64, 0.37f) printf("{\"mix\": \"%s\", \"clk_per_pair\": %.2f}\n", NAME, cyc[0] / NI); }
    MIXRUN(0, "v_fma only") MIXRUN(1, "+ s_add_u32") MIXRUN(2, "+ s_nop 0") MIXRUN(3, "+ ds_read_b64") MIXRUN(4, "+ ds_write_b64") MIXRUN(5, "+ v_nop")
    printf("== D. dependent v_fma chain (wave 0) vs filler stream (wave 4, same SIMD): clocks per chain link; filler clocks per instruction\n");
#define PRIORUN(FILL, PRIO, NAME) { LAUNCH((k_prio<FILL, PRIO>), 320, 0.37f) printf("{\"prio\": \"%s\", \"setprio\": %d, \"chain_clk_per_link\": %.2f, \"filler_clk_per_instr\": %.2f}\n", NAME, (int)PRIO, cyc[0] / NI, cyc[1] / (4 * NI)); }
    PRIORUN(0, false, "no filler") PRIORUN(1, false, "v_fma filler") PRIORUN(1, true, "v_fma filler") PRIORUN(2, false, "v_pk_fma filler")
    PRIORUN(2, true, "v_pk_fma filler") PRIORUN(3, false, "dpp filler") PRIORUN(3, true, "dpp filler")
    printf("== E. 4 waves per SIMD, independent stream, first LANES lanes enabled: clocks per instruction per SIMD\n");
#define EXECRUN(LANES, PK) { LAUNCH((k_exec<LANES, PK>), 1024, 0.37f) printf("{\"exec_lanes\": %d, \"pk\": %d, \"clk_per_instr_per_simd\": %.3f}\n", LANES, (int)PK, cyc[0] / NI / 4.0); }
    EXECRUN(64, false) EXECRUN(32, false) EXECRUN(16, false) EXECRUN(8, false) EXECRUN(64, true) EXECRUN(32, true) EXECRUN(16, true)
    printf("== F. LDS flag ping-pong between two waves on different SIMDs: clocks per round trip\n");
    { LAUNCH(k_pingpong<false>, 128, 2000) printf("{\"pingpong\": \"flag only\", \"clk_per_round_trip\": %.1f}\n", cyc[0] / 2000.0); }
    { LAUNCH(k_pingpong<true>, 128, 2000) printf("{\"pingpong\": \"flag + 16 B payload per lane\", \"clk_per_round_trip\": %.1f}\n", cyc[0] / 2000.0); }
    printf("== G. dependent LDS reads, one wave alone: clocks per read\n");
    { LAUNCH(k_ldslat<32>, 64, 2000) printf("{\"lds_dependent_read\": \"b32\", \"clk\": %.1f}\n", cyc[0] / 2000.0); }
    { LAUNCH(k_ldslat<64>, 64, 2000) printf("{\"lds_dependent_read\": \"b64\", \"clk\": %.1f}\n", cyc[0] / 2000.0); }
    { LAUNCH(k_ldslat<128>, 64, 2000) printf("{\"lds_dependent_read\": \"b128\", \"clk\": %.1f}\n", cyc[0] / 2000.0); }
    printf("== H. s_barrier, 6 waves: clocks per barrier\n");
    { LAUNCH(k_barrier, 384, 2000) printf("{\"barrier_6_waves_clk\": %.1f}\n", cyc[0] / 2000.0); }
    return 0;
}
