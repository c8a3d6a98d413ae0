// cores.hip -- when do two workgroups of the fused kernel's shape (384 threads, ~73 KB LDS, ~128 VGPRs) share a CU?
// Each workgroup spins for a fixed number of shader clocks; 512 workgroups take as long as 256 if they co-reside (2 per CU)
// and twice as long if they do not.  Sweeps the LDS size and a VGPR floor.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDSB, int NV> __global__ __launch_bounds__(384) void k(float* p, long long spin) {
    extern __shared__ float dyn[];
    __shared__ float s[LDSB / 4];
    float r[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) r[i] = p[(threadIdx.x + i) & 1023];
    s[threadIdx.x] = r[0];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) {
#pragma unroll
        for (int i = 0; i < NV; i++) r[i] = r[i] * 1.0001f + s[(threadIdx.x + i) % 384];
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < NV; i++) acc += r[i];
    if (acc == 12345.678f) p[threadIdx.x] = acc;
}

template <int LDSB, int NV> void run(float* d, const char* what) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms[3] = { 0, 0, 0 };
    const int grids[3] = { 256, 512, 1024 };
    for (int g = 0; g < 3; g++) {
        hipLaunchKernelGGL((k<LDSB, NV>), dim3(grids[g]), dim3(384), 0, 0, d, 2000000LL);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((k<LDSB, NV>), dim3(grids[g]), dim3(384), 0, 0, d, 2000000LL);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms[g], a, b);
    }
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<LDSB, NV>, 384, 0);
    printf("{\"case\": \"%s\", \"lds_bytes\": %d, \"occupancy_api_blocks_per_cu\": %d, \"ms_256wg\": %.3f, \"ms_512wg\": %.3f, \"ms_1024wg\": %.3f}\n",
           what, LDSB, nb, ms[0], ms[1], ms[2]);
}

int main() {
    float* d; hipMalloc(&d, 4096 * 4); hipMemset(d, 0, 4096 * 4);
    run<16384, 8>(d, "16 KB LDS, few VGPRs");
    run<40960, 8>(d, "40 KB LDS, few VGPRs");
    run<45056, 8>(d, "44 KB LDS, few VGPRs");
    run<49152, 8>(d, "48 KB LDS, few VGPRs");
    run<53248, 8>(d, "52 KB LDS, few VGPRs");
    run<57344, 8>(d, "56 KB LDS, few VGPRs");
    run<61440, 8>(d, "60 KB LDS, few VGPRs");
    run<65536, 8>(d, "64 KB LDS, few VGPRs");
    run<73056, 8>(d, "73 KB LDS, few VGPRs");
    run<81920, 8>(d, "80 KB LDS, few VGPRs");
    run<32768, 8>(d, "32 KB LDS, few VGPRs");
    run<27648, 8>(d, "27 KB LDS, few VGPRs");
    run<73056, 100>(d, "73 KB LDS, >100 VGPRs");
    run<32768, 100>(d, "32 KB LDS, >100 VGPRs");
    return 0;
}
