// mfma_fir.hip -- the FIR bulk of the demodulator chain on the matrix pipe, one channel per COLUMN (microbenchmark for DESIGN.md
// section 9: what a throughput mapping for >= 16k channels per GPU has to spend on the band-edge and RRC sums).
//
// For a block of 16 consecutive output samples of 64 channels (= 128 columns: channel x {re, im}) the sums over the taps that are
// at least 16 samples old are Toeplitz-matrix x sample-block products:
//     D[m][col] = sum_k T[m][k] * x[j0 + k][col],     T[m][k] = h[k - m] for 0 <= k - m < NFAR, else 0        (m = 0..15)
// i.e. D[m] = sum_{t < NFAR} h[t] * x[j0 + m + t]: ascending t = ascending tap index = oldest sample first, the order of the
// arithmetic contract's fmaf chain -- and chained v_mfma_f32_16x16x4_f32 over ascending k ARE that chain (zero entries of T add
// +-0 to a chain that started at +0 and leave it untouched).  Three tap sets per block: the band-edge pair's a[] and b[] (48 of
// their 65 taps; the newest 17 stay with the per-sample loop) and the RRC's 65 taps (all of them: the RRC has no feedback).
// Per 16 samples x 64 channels: 8 column tiles x (16 + 16 + 20) k-steps = 416 MFMAs = 13.3 k cycles of one SIMD's matrix pipe
// -> 13 SIMD-cycles = 3.25 CU-clocks per channel-sample.
//
// The program (a) checks a small case against the fmaf chain on the host, bit for bit, and (b) times the kernel on
// C channels x N samples with the results folded into a checksum (on chip the sums would feed the recurrence, not HBM).
// Build + run:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_fir mfma_fir.hip && ./mfma_fir
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBE = 48;        // band-edge taps handled here (the far 48 of 65)
constexpr int kRRC = 65;       // RRC taps
constexpr int kKbe = (kBE + 15 + 3) / 4;      // 16 k-steps
constexpr int kKrrc = (kRRC + 15 + 3) / 4;    // 20 k-steps

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

// x: time-major [N + pad][2 C] floats (frame n = the 2 C floats of sample n).  out (optional): [3][N][2 C]; sum: per-wave checksum.
// One wave = ONE column tile (16 columns = 8 channels) over all of time: 32 waves per 256 channels keep several waves per SIMD in
// flight, so the operand loads of one hide behind the matrix instructions of the others.
__global__ __launch_bounds__(256) void k_fir(const float* __restrict__ x, int C, int N, const float* __restrict__ ta,
                                             const float* __restrict__ tb, const float* __restrict__ tr, float* __restrict__ out,
                                             float* __restrict__ sum) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int col0 = (blockIdx.x * 4 + w) * 16;
    const int W = 2 * C;
    // constant A fragments: T[m = c][k = 4 s + g] = h[k - m]
    float fa[kKbe], fb[kKbe], fr[kKrrc];
#pragma unroll
    for (int s = 0; s < kKbe; s++) {
        const int t = 4 * s + g - c;
        fa[s] = t >= 0 && t < kBE ? ta[t] : 0.f;
        fb[s] = t >= 0 && t < kBE ? tb[t] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < kKrrc; s++) {
        const int t = 4 * s + g - c;
        fr[s] = t >= 0 && t < kRRC ? tr[t] : 0.f;
    }
    float acc_sum = 0.f;
    // The samples reach the matrix pipe from LDS, like in a kernel whose own FLL stage produces them: per wave a ring of 128 time
    // rows x 16 columns (8 KB); a block of 16 outputs reads rows j0 .. j0 + 79 and brings in the 16 rows behind them.  A fragment
    // read (row 4 s + g, column c) covers 64 consecutive floats: conflict-free.
    __shared__ float ring[4][128][16];
    float (*R)[16] = ring[w];
    const int lr = lane >> 2, lc = (lane & 3) * 4;          // this lane's share of a 16-row load: row lr, columns lc .. lc + 3
    auto fetch = [&](int row0) {                             // rows row0 .. row0 + 15 of this wave's column tile -> ring
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)(row0 + lr) * W + col0 + lc);
        *reinterpret_cast<float4*>(&R[(row0 + lr) & 127][lc]) = v;
    };
    for (int r0 = 0; r0 < 80; r0 += 16) fetch(r0);
    for (int j0 = 0; j0 + 16 <= N; j0 += 16) {
        {
            const int nt = 0;
            f32x4 da = { 0.f, 0.f, 0.f, 0.f }, db = da, dr = da;
#pragma unroll
            for (int s = 0; s < kKrrc; s++) {
                const float b = R[(j0 + 4 * s + g) & 127][c];
                dr = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[s], b, dr, 0, 0, 0);
                if (s < kKbe) {
                    da = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], b, da, 0, 0, 0);
                    db = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[s], b, db, 0, 0, 0);
                }
            }
            fetch(j0 + 80);                                  // (the input is padded by 96 rows behind N)
            if (out) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const size_t o = (size_t)(j0 + 4 * g + r) * W + col0 + 16 * nt + c;
                    out[o] = da[r];
                    out[(size_t)N * W + o] = db[r];
                    out[2 * (size_t)N * W + o] = dr[r];
                }
            } else {
                acc_sum += (da[0] + da[1] + da[2] + da[3]) + (db[0] + db[1] + db[2] + db[3]) + (dr[0] + dr[1] + dr[2] + dr[3]);
            }
        }
    }
    if (!out) sum[blockIdx.x * 256 + threadIdx.x] = acc_sum;
}

int main() {
    std::vector<float> ta(kBE), tb(kBE), tr(kRRC);
    srand(3);
    auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
    for (auto& v : ta) v = rnd() * 0.1f;
    for (auto& v : tb) v = rnd() * 0.1f;
    for (auto& v : tr) v = rnd() * 0.1f;
    ta[5] = 0.f; tb[7] = -0.f; tr[11] = 1e-42f;      // zero, minus zero, a subnormal tap
    float *d_ta, *d_tb, *d_tr;
    CK(hipMalloc(&d_ta, sizeof(float) * kBE)); CK(hipMalloc(&d_tb, sizeof(float) * kBE)); CK(hipMalloc(&d_tr, sizeof(float) * kRRC));
    CK(hipMemcpy(d_ta, ta.data(), sizeof(float) * kBE, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tb, tb.data(), sizeof(float) * kBE, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tr, tr.data(), sizeof(float) * kRRC, hipMemcpyHostToDevice));
    {   // (a) exactness: 256 channels x 64 samples against the fmaf chain
        const int C = 256, N = 64, W = 2 * C, P = 96;
        std::vector<float> x((size_t)(N + P) * W);
        for (auto& v : x) v = rnd();
        x[5] = 0.f; x[W + 9] = -0.f; x[3 * W + 1] = 3e-41f;
        float *d_x, *d_o, *d_s;
        CK(hipMalloc(&d_x, sizeof(float) * x.size())); CK(hipMalloc(&d_o, sizeof(float) * 3 * (size_t)N * W)); CK(hipMalloc(&d_s, 65536));
        CK(hipMemcpy(d_x, x.data(), sizeof(float) * x.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_fir, dim3(C / 32), dim3(256), 0, 0, d_x, C, N, d_ta, d_tb, d_tr, d_o, d_s);
        CK(hipDeviceSynchronize());
        std::vector<float> o(3 * (size_t)N * W);
        CK(hipMemcpy(o.data(), d_o, sizeof(float) * o.size(), hipMemcpyDeviceToHost));
        long long bad = 0;
        for (int n = 0; n < N; n++)
            for (int q = 0; q < W; q++) {
                float a = 0.f, b = 0.f, r = 0.f;
                for (int t = 0; t < kRRC; t++) {
                    const float v = x[(size_t)(n + t) * W + q];
                    if (t < kBE) { a = fmaf(ta[t], v, a); b = fmaf(tb[t], v, b); }
                    r = fmaf(tr[t], v, r);
                }
                const size_t i = (size_t)n * W + q;
                bad += __builtin_memcmp(&a, &o[i], 4) != 0;
                bad += __builtin_memcmp(&b, &o[(size_t)N * W + i], 4) != 0;
                bad += __builtin_memcmp(&r, &o[2 * (size_t)N * W + i], 4) != 0;
            }
        std::printf("{\"check\": \"%d channels x %d samples x 3 sums vs the fmaf chain\", \"differing_bit_patterns\": %lld}\n", C, N, bad);
        CK(hipFree(d_x)); CK(hipFree(d_o)); CK(hipFree(d_s));
        if (bad) return 1;
    }
    for (int C : { 16384, 65536 }) {   // (b) rate
        const int N = 4096, W = 2 * C, P = 96;
        float *d_x, *d_s;
        CK(hipMalloc(&d_x, sizeof(float) * (size_t)(N + P) * W));
        CK(hipMemset(d_x, 0, sizeof(float) * (size_t)(N + P) * W));
        CK(hipMalloc(&d_s, sizeof(float) * (size_t)C * 8));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_fir, dim3(C / 32), dim3(256), 0, 0, d_x, C, N, d_ta, d_tb, d_tr, (float*)nullptr, d_s);
        CK(hipEventRecord(e0, 0));
        const int reps = 5;
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_fir, dim3(C / 32), dim3(256), 0, 0, d_x, C, N, d_ta, d_tb, d_tr, (float*)nullptr, d_s);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        int clk = 0, cus = 0;
        CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
        CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
        const double cs = (double)C * N;
        const double mfma = cs / (16.0 * 64.0) * 416.0;
        std::printf("{\"channels\": %d, \"samples\": %d, \"ms\": %.4f, \"gsamples_per_s\": %.1f, \"cu_clocks_per_channel_sample\": %.2f, "
                    "\"mfma_flop_per_s_T\": %.1f, \"frac_of_157_TFLOPs\": %.3f}\n",
                    C, N, ms, cs / ms / 1e6, ms * 1e-3 * clk * 1e3 * cus / cs, mfma * 2048.0 / (ms * 1e-3) / 1e12,
                    mfma * 2048.0 / (ms * 1e-3) / 157.3e12);
        CK(hipFree(d_x)); CK(hipFree(d_s));
    }
    return 0;
}
