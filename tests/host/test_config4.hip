// test_config4.hip -- BASELINE config 4 at full size on whatever GPUs the box has: 32768 channels as 8 shards of 4096
// (SURVEY.md 8(e)), one process, dsp::demod::PI4DQPSKMultiBank::processDevice, input generated ON the GPU.
// Test driver only (the product side, host/pi4dqpsk_gpu.cpp, is plain C++ and needs no HIP): the one kernel here expands a
// small base of modulated channels into the full bank with EXACT float operations, so the Python side can rebuild any
// channel's input bit for bit for the oracle:
//     channel c  =  base[c % B]  x  amp[(c / B) % A]  x  j^((c / B) % 4)          (A = 8 amplitudes: a period of B * A channels)
// Usage: test_config4 <base.f32 [B][n] complex64> <B> <n> <C> <shards> <rows.u8> <nbits.i32> <sel.txt> [dev ...]
//   rows.u8   full bit rows [len(sel)][stride] of the channels listed in sel.txt (one index per line)
//   nbits.i32 counts of ALL C channels
// stdout: "stride S", one "shard g: channels [a, b) on device d" line per shard, "kernel_ms ..." per shard, and
// "period_check P mismatches M": every channel c >= P compared byte for byte (row and count) with channel c - P, which
// received the same samples -- whatever shard, workgroup and lane either ran on.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sdrpp-tetra-demodulator_amd/host/pi4dqpsk_gpu.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 10; } } while (0)

static const int kAmps = 8;
__constant__ float c_amp[kAmps];

// out[cl][i] for the shard's local channel cl = global channel first + cl
__global__ void k_expand(const float2* __restrict__ base, int B, int n, int first, int count, float2* __restrict__ out) {
    const long long total = (long long)count * n;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int cl = (int)(t / n), i = (int)(t % n);
        const int c = first + cl, k = c / B;
        const float a = c_amp[k % kAmps];
        float2 v = base[(long long)(c % B) * n + i];
        v.x *= a; v.y *= a;
        switch (k & 3) {                                  // x j^k: exact
        case 1: v = make_float2(-v.y, v.x); break;
        case 2: v = make_float2(-v.x, -v.y); break;
        case 3: v = make_float2(v.y, -v.x); break;
        default: break;
        }
        out[t] = v;
    }
}

int main(int argc, char** argv) {
    if (argc < 9) { std::fprintf(stderr, "usage: see the head of test_config4.hip\n"); return 1; }
    const int B = std::atoi(argv[2]), n = std::atoi(argv[3]), C = std::atoi(argv[4]), G = std::atoi(argv[5]);
    std::vector<int> devs;
    for (int i = 9; i < argc; i++) devs.push_back(std::atoi(argv[i]));
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    if (devs.empty())
        for (int g = 0; g < G; g++) devs.push_back(g % ndev);          // a GPU per shard where there are that many, folded otherwise
    if ((int)devs.size() != G) return 1;
    std::vector<float> base((size_t)B * n * 2);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(base.data(), sizeof(float), base.size(), f) != base.size()) return 2;
    std::fclose(f);
    std::vector<int> sel;
    f = std::fopen(argv[8], "r");
    if (!f) return 2;
    for (int v; std::fscanf(f, "%d", &v) == 1;) sel.push_back(v);
    std::fclose(f);
    const float amps[kAmps] = { 1.0f, 0.37f, 0.81f, 0.052f, 0.6f, 0.23f, 0.95f, 0.11f };

    tetra_demod_config_t cfg;
    tetra_demod_default_config(&cfg);
    cfg.n_channels = C;
    cfg.max_samples = n;
    dsp::demod::PI4DQPSKMultiBank mb;
    int rc = mb.init(cfg, devs);
    if (rc != TETRA_OK) { std::fprintf(stderr, "init failed: %s\n", tetra_demod_strerror(rc)); return 3; }
    const int stride = mb.bitsStride(n);
    std::printf("stride %d\n", stride);

    std::vector<dsp::complex_t*> dIn((size_t)G, nullptr);
    std::vector<uint8_t*> dBits((size_t)G, nullptr);
    std::vector<int32_t*> dNb((size_t)G, nullptr);
    std::vector<float2*> dBase((size_t)ndev, nullptr);
    for (int g = 0; g < G; g++) {
        int first, count, dev;
        mb.shardInfo(g, first, count, dev);
        std::printf("shard %d: channels [%d, %d) on device %d\n", g, first, first + count, dev);
        CK(hipSetDevice(dev));
        if (!dBase[dev]) {
            CK(hipMalloc((void**)&dBase[dev], sizeof(float) * base.size()));
            CK(hipMemcpy(dBase[dev], base.data(), sizeof(float) * base.size(), hipMemcpyHostToDevice));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(c_amp), amps, sizeof(amps)));
        }
        CK(hipMalloc((void**)&dIn[g], sizeof(dsp::complex_t) * (size_t)count * n));
        CK(hipMalloc((void**)&dBits[g], (size_t)count * stride));
        CK(hipMemset(dBits[g], 0xee, (size_t)count * stride));
        CK(hipMalloc((void**)&dNb[g], sizeof(int32_t) * count));
        hipLaunchKernelGGL(k_expand, dim3(4096), dim3(256), 0, 0, dBase[dev], B, n, first, count, reinterpret_cast<float2*>(dIn[g]));
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
    }
    rc = mb.processDevice(n, dIn.data(), dBits.data(), dNb.data());
    if (rc != TETRA_OK) { std::fprintf(stderr, "processDevice failed: %s\n", tetra_demod_strerror(rc)); return 5; }

    std::vector<uint8_t> bits((size_t)C * stride);
    std::vector<int32_t> nb((size_t)C);
    for (int g = 0; g < G; g++) {
        int first, count, dev;
        mb.shardInfo(g, first, count, dev);
        CK(hipSetDevice(dev));
        CK(hipMemcpy(bits.data() + (size_t)first * stride, dBits[g], (size_t)count * stride, hipMemcpyDeviceToHost));
        CK(hipMemcpy(nb.data() + first, dNb[g], sizeof(int32_t) * count, hipMemcpyDeviceToHost));
    }
    // every channel against the one a period earlier: same samples -> same count and bits
    const int P = B * kAmps;
    long long mism = 0;
    for (int c = P; c < C; c++) {
        const int r = c - P;
        if (nb[c] != nb[r] || nb[c] < 0 || nb[c] > stride || std::memcmp(bits.data() + (size_t)c * stride, bits.data() + (size_t)r * stride, (size_t)nb[c])) mism++;
    }
    std::printf("period_check %d mismatches %lld\n", P, mism);
    FILE* fb = std::fopen(argv[6], "wb");
    for (int c : sel) {
        if (c < 0 || c >= C) return 6;
        std::fwrite(bits.data() + (size_t)c * stride, 1, (size_t)stride, fb);
    }
    std::fclose(fb);
    FILE* fn = std::fopen(argv[7], "wb");
    std::fwrite(nb.data(), sizeof(int32_t), (size_t)C, fn);
    std::fclose(fn);
    for (int g = 0; g < G; g++) { (void)hipFree(dIn[g]); (void)hipFree(dBits[g]); (void)hipFree(dNb[g]); }
    for (auto p : dBase) if (p) (void)hipFree(p);
    return 0;
}
