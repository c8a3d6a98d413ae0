// clk.hip -- what does the shader clock do under load?  Each wave runs a dependent chain of N packed FMAs (4 clocks each at any
// frequency: 1 instruction per SIMD turn) and records s_memtime (shader clock counter) and s_memrealtime (100 MHz) around it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int iters, int heavy) {
    float r = threadIdx.x * 1e-3f;
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if (heavy) {
        asm volatile("v_mov_b32 v20, %0\nv_mov_b32 v21, %0\nv_mov_b32 v22, 0x3f7fff00\nv_mov_b32 v23, 0x3f7fff00\n"
                     "1:\n"
                     "v_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[22:23]\nv_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[22:23]\n"
                     "v_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[22:23]\nv_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[22:23]\n"
                     "v_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[22:23]\nv_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[22:23]\n"
                     "v_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[22:23]\nv_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[22:23]\n"
                     "s_sub_u32 %1, %1, 1\ns_cmp_lg_u32 %1, 0\ns_cbranch_scc1 1b\nv_add_f32 %0, v20, v21\n"
                     : "+v"(r), "+s"(iters) : : "v20", "v21", "v22", "v23", "scc");
    } else {
        asm volatile("v_mov_b32 v20, %0\nv_mov_b32 v22, 0x3f7fff00\n"
                     "1:\n"
                     "v_fma_f32 v20, v20, v22, v22\nv_fma_f32 v20, v20, v22, v22\nv_fma_f32 v20, v20, v22, v22\nv_fma_f32 v20, v20, v22, v22\n"
                     "v_fma_f32 v20, v20, v22, v22\nv_fma_f32 v20, v20, v22, v22\nv_fma_f32 v20, v20, v22, v22\nv_fma_f32 v20, v20, v22, v22\n"
                     "s_sub_u32 %1, %1, 1\ns_cmp_lg_u32 %1, 0\ns_cbranch_scc1 1b\nv_mov_b32 %0, v20\n"
                     : "+v"(r), "+s"(iters) : : "v20", "v22", "scc");
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = w1 - w0; }
    if (r == 12345.f) out[0] = 0;
}
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 16 * 65536);
    std::vector<unsigned long long> h(2 * 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400000;
    for (int heavy : {0, 1})
    for (int wg : {1, 256, 1024, 4096, 8192}) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(wg), dim3(64), 0, 0, d, iters, heavy);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        (void)hipMemcpy(h.data(), d, 16 * wg, hipMemcpyDeviceToHost);
        double st = 0, rt = 0;
        for (int i = 0; i < wg; i++) { st += h[2 * i]; rt += h[2 * i + 1]; }
        st /= wg; rt /= wg;
        std::printf("{\"heavy\": %d, \"waves\": %d, \"event_ms\": %.3f, \"memtime_ticks\": %.0f, \"realtime_ticks_100MHz\": %.0f, "
                    "\"memtime_MHz\": %.1f, \"instr_per_us\": %.1f, \"ns_per_instr\": %.3f}\n",
                    heavy, wg, ms, st, rt, st / (rt / 100.0), 8.0 * iters / (rt / 100.0), (rt * 10.0) / (8.0 * iters));
    }
    return 0;
}
