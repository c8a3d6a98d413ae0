// tetra_chan.hip -- polyphase channeliser front-end (see include/tetra_chan.h): HIP kernels + C ABI.
//
// Every output frame = weighted overlap-add of the L = P*M newest samples onto M bins indexed by ABSOLUTE sample time mod M (so the
// DFT needs no per-frame phase correction), then an M-point DFT; frames go out time-major, out[m][k] -- exactly the
// TETRA_LAYOUT_TIME_MAJOR input of the demodulator.  Three kernels, chosen from the geometry (identical results within the float32
// tolerance the tests hold against the double-precision definition):
//   k_channelise_fft    M = 800 at D = M / 2 (BASELINE config 5; default there since round 5): the DFT as a 32 x 5 x 5 mixed-radix
//                       FFT in registers / LDS, 8 frames per workgroup sharing their sample loads -- 39 kflop per frame, bound by the
//                       kernel's 120 MB of HBM traffic per 12500 frames (36 us = 41 % of the HBM peak); lane code in chan_fft_core.hpp
//   k_channelise_mfma   M = 800 = 25 x 32, any decimation (round 4; TETRA_CHAN_FLAG_MATRIX_DFT forces it): both DFT stages as real
//                       matrix products with the 2 x 2 block form of the twiddle matrices, 104 + 128 chained v_mfma_f32_16x16x4_f32
//                       per frame (exact f32 fma chains; the f32 MFMA runs on the vector pipe on gfx950: 0.11 ms, pipe-bound)
//   k_channelise        any M = N1 * N2 with N1, N2 <= 64 (TETRA_CHAN_FLAG_VALU_DFT forces it): direct sums, N1 + N2 complex MACs
//                       per output, one workgroup per frame (0.21 ms for config 5's geometry)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/tetra_chan.h"
#include "chan_fft_core.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxFactor = 64;

struct ChanParams {
    const float2* xbuf;    // [L-1 history][n_in new]
    float2* out;           // [frames][M]
    const float* h;        // prototype [L]
    const float2* w1;      // exp(-j 2 pi i / N1), i < N1
    const float2* w2;      // exp(-j 2 pi i / N2), i < N2
    const float2* wm;      // exp(-j 2 pi i / M),  i < M
    int M, P, D, N1, N2;
    int ph0;               // samples already consumed towards the first frame of this call
    long long abs0;        // absolute index of xbuf[L-1] (the first new sample)
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__global__ __launch_bounds__(kThreads) void k_channelise(ChanParams p) {
    extern __shared__ float2 lds[];          // v[M] | b[N1][N2 + 1]
    float2* v = lds;
    __shared__ float2 tw1[kMaxFactor], tw2[kMaxFactor];     // the two short twiddle tables: few distinct entries per wave
    if (threadIdx.x < p.N1) tw1[threadIdx.x] = p.w1[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x - 64 < p.N2) tw2[threadIdx.x - 64] = p.w2[threadIdx.x - 64];
    float2* b = lds + p.M;                   // rows padded by one element: the row DFTs read b[k1][n2] with k1 across the lanes, and a
                                             // row stride of N2 = 32 complex (64 dwords) would put every lane on the same LDS banks
    const int M = p.M, L = p.M * p.P;
    const int j = blockIdx.x;
    const int newest = (j + 1) * p.D - 1 - p.ph0;                 // index into the new samples
    const long long n_abs = p.abs0 + newest;                        // absolute time of the frame's newest sample
    const int nm = (int)(n_abs % M);
    const float2* xn = p.xbuf + (L - 1) + newest;                   // xn[-l] = x[n_abs - l]
    // fold: v[r] = sum_p h[l0 + pM] * x[n_abs - l0 - pM],  l0 = (n_abs - r) mod M
    for (int r = threadIdx.x; r < M; r += kThreads) {
        int l0 = nm - r;
        if (l0 < 0) l0 += M;
        float2 acc = make_float2(0.f, 0.f);
        for (int q = 0; q < p.P; q++) {
            const int l = l0 + q * M;
            const float hv = p.h[l];
            const float2 xv = xn[-l];
            acc.x = fmaf(hv, xv.x, acc.x);
            acc.y = fmaf(hv, xv.y, acc.y);
        }
        v[r] = acc;
    }
    __syncthreads();
    // column DFTs + twiddle: b[k1][n2] = W_M^{n2 k1} * sum_{n1} v[n1*N2 + n2] * W_N1^{n1 k1}
    const int N1 = p.N1, N2 = p.N2;
    for (int o = threadIdx.x; o < M; o += kThreads) {
        const int k1 = o / N2, n2 = o % N2;
        float2 acc = make_float2(0.f, 0.f);
        int idx = 0;
        for (int n1 = 0; n1 < N1; n1++) {
            const float2 t = cmul(v[n1 * N2 + n2], tw1[idx]);
            acc.x += t.x;
            acc.y += t.y;
            idx += k1;
            if (idx >= N1) idx -= N1;
        }
        b[k1 * (N2 + 1) + n2] = cmul(acc, p.wm[(n2 * k1) % M]);
    }
    __syncthreads();
    // row DFTs: X[k1 + N1*k2] = sum_{n2} b[k1][n2] * W_N2^{n2 k2}
    float2* dst = p.out + (long long)j * M;
    for (int o = threadIdx.x; o < M; o += kThreads) {
        const int k2 = o / N1, k1 = o % N1;                          // o = k1 + N1*k2: consecutive threads, consecutive bins
        float2 acc = make_float2(0.f, 0.f);
        int idx = 0;
        for (int n2 = 0; n2 < N2; n2++) {
            const float2 t = cmul(b[k1 * (N2 + 1) + n2], tw2[idx]);
            acc.x += t.x;
            acc.y += t.y;
            idx += k2;
            if (idx >= N2) idx -= N2;
        }
        dst[o] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Matrix-pipe form for M = N1 x N2 = 25 x 32.  One workgroup = four waves, looping over frames; per frame:
//   fold       v[r] as before, written to LDS as the A operand of stage 1:  At[k][n2],  k = n1 (real parts) | N1 + n1 (imaginary)
//   stage 1    Bt[n2][k1] = sum_n1 V[n1][n2] W_N1^(n1 k1)        as  [n2 x 52] . [52 x 64]: columns k1 (re) | 32 + k1 (im);
//              constant operand Bc = [[Wr, Wi], [-Wi, Wr]] (zero-padded), 13 k-steps; then the twiddle W_M^(n2 k1) (four
//              constants per lane) and the result goes to LDS as the B operand of stage 2:  Bo[k][k1], k = n2 (re) | 32 + n2 (im)
//   stage 2    Xt[k2][k1] = sum_n2 W_N2^(n2 k2) Bt[n2][k1]        as  [64 x 64] . [64 x 32]: rows k2 (re) | 32 + k2 (im);
//              constant operand Ac = [[Wr, -Wi], [Wi, Wr]], 16 k-steps; lane (g = l >> 4, c = l & 15) ends up with re and im of
//              X[k1 + N1 k2] for k2 = 16 mt + 4 g + r, k1 = 16 nt + c: 16 consecutive bins per store = 128-byte runs.
// Wave w owns two output tiles per stage that share their data operand (stage 1: n2 rows 16 (w & 1) .., columns re / im of k1
// 16 (w >> 1) ..; stage 2: rows re / im of k2 16 (w & 1) .., columns k1 16 (w >> 1) ..), so a stage costs a wave 13 (16) LDS
// reads and 26 (32) MFMAs.  v_mfma_f32_16x16x4_f32 operand layout: a = A[row l & 15][k l >> 4], b = B[k l >> 4][col l & 15],
// acc[r] = D[row 4 (l >> 4) + r][col l & 15] (checked by tetra_demod_debug_mfma_selftest).
// ---------------------------------------------------------------------------------------------------------------------
typedef float chan_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMN1 = 25, kMN2 = 32;
constexpr int kMK1 = (2 * kMN1 + 3) / 4;          // 13 k-steps of stage 1 (K = 50 -> 52)
constexpr int kMK2 = 2 * kMN2 / 4;                // 16 k-steps of stage 2 (K = 64)
constexpr int kMAtStride = 48;                    // floats per At row (32 used): rows 16 banks apart -> the four k of a fragment read hit 2 x 16 banks
constexpr int kMBoStride = 36;                    // floats per Bo row (32 used)

struct ChanMfmaParams {
    const float2* xbuf;
    float2* out;
    const float* h;
    const float* bc;       // [4 kMK1][64] stage-1 constant operand
    const float* ac;       // [64][64]     stage-2 constant operand
    const float2* wm;      // exp(-j 2 pi i / M)
    int P, D, frames;
    int ph0;
    long long abs0;
};

template <int P> __global__ __launch_bounds__(kThreads, 3) void k_channelise_mfma(ChanMfmaParams p) {      // 3 waves per SIMD: <= 168 VGPRs
    constexpr int M = kMN1 * kMN2, L = M * P;
    __shared__ float At[4 * kMK1][kMAtStride];
    __shared__ float Bo[2 * kMN2][kMBoStride];
    __shared__ float hs[L];              // the prototype: every frame reads all of it (rotated by the frame's time mod M)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int wlo = w & 1, whi = w >> 1;
    for (int i = tid; i < L; i += kThreads) hs[i] = p.h[i];
    // constant operands of this wave's tiles (registers for the workgroup's life)
    float b1re[kMK1], b1im[kMK1], a2re[kMK2], a2im[kMK2];
#pragma unroll
    for (int s = 0; s < kMK1; s++) {
        b1re[s] = p.bc[(4 * s + g) * 64 + 16 * whi + c];
        b1im[s] = p.bc[(4 * s + g) * 64 + 32 + 16 * whi + c];
    }
#pragma unroll
    for (int s = 0; s < kMK2; s++) {
        a2re[s] = p.ac[(16 * wlo + c) * 64 + 4 * s + g];
        a2im[s] = p.ac[(32 + 16 * wlo + c) * 64 + 4 * s + g];
    }
    // W_M^(n2 k1) for this lane's four stage-1 results: n2 = 16 wlo + 4 g + r, k1 = 16 whi + c
    float2 tw[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int n2 = 16 * wlo + 4 * g + r, k1 = 16 * whi + c;
        tw[r] = k1 < kMN1 ? p.wm[(n2 * k1) % M] : make_float2(0.f, 0.f);
    }
    // padding rows of the stage-1 data operand (k = 50, 51) are zero for ever
    if (tid < 2 * kMAtStride) At[2 * kMN1 + tid / kMAtStride][tid % kMAtStride] = 0.f;
    constexpr int kRounds = (M + kThreads - 1) / kThreads;      // bins per thread (the last round is partial)
    // fold, v[r] = sum_q h[l0 + q M] x[n_abs - l0 - q M] with l0 = (n_abs - r) mod M.  ALL of a thread's samples of a frame are
    // requested at once (kRounds x P loads in flight), and a frame's samples are requested while the frame before it is still in
    // its DFT stages: the memory latency of the fold hides behind the matrix pipe.
    float2 xv[kRounds][P];
    int l0s[kRounds];
    auto request = [&](int j) {
        const int newest = (j + 1) * p.D - 1 - p.ph0;
        const long long n_abs = p.abs0 + newest;
        const int nm = (int)(n_abs % M);
        const float2* xn = p.xbuf + (L - 1) + newest;
#pragma unroll
        for (int i = 0; i < kRounds; i++) {
            const int r = tid + kThreads * i;
            int l0 = nm - r;
            l0 += l0 < 0 ? M : 0;
            l0 = r < M ? l0 : 0;
            l0s[i] = l0;
#pragma unroll
            for (int q = 0; q < P; q++) xv[i][q] = xn[-(l0 + q * M)];
        }
    };
    if ((int)blockIdx.x < p.frames) request(blockIdx.x);
    __syncthreads();              // hs and the zero rows of At are in place
    for (int j = blockIdx.x; j < p.frames; j += gridDim.x) {
#pragma unroll
        for (int i = 0; i < kRounds; i++) {
            const int r = tid + kThreads * i;
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int q = 0; q < P; q++) {
                const float hv = hs[l0s[i] + q * M];
                acc.x = fmaf(hv, xv[i][q].x, acc.x);
                acc.y = fmaf(hv, xv[i][q].y, acc.y);
            }
            if (r < M) {
                const int n1 = r / kMN2, n2 = r % kMN2;
                At[n1][n2] = acc.x;
                At[kMN1 + n1][n2] = acc.y;
            }
        }
        if (j + (int)gridDim.x < p.frames) request(j + gridDim.x);
        __syncthreads();
        {   // stage 1
            chan_f32x4 dre = { 0.f, 0.f, 0.f, 0.f }, dim_ = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int s = 0; s < kMK1; s++) {
                const float a = At[4 * s + g][16 * wlo + c];
                dre = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1re[s], dre, 0, 0, 0);
                dim_ = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1im[s], dim_, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float2 t = cmul(make_float2(dre[r], dim_[r]), tw[r]);
                const int n2 = 16 * wlo + 4 * g + r;
                Bo[n2][16 * whi + c] = t.x;
                Bo[kMN2 + n2][16 * whi + c] = t.y;
            }
        }
        __syncthreads();
        {   // stage 2
            chan_f32x4 xre = { 0.f, 0.f, 0.f, 0.f }, xim = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int s = 0; s < kMK2; s++) {
                const float b = Bo[4 * s + g][16 * whi + c];
                xre = __builtin_amdgcn_mfma_f32_16x16x4f32(a2re[s], b, xre, 0, 0, 0);
                xim = __builtin_amdgcn_mfma_f32_16x16x4f32(a2im[s], b, xim, 0, 0, 0);
            }
            const int k1 = 16 * whi + c;
            if (k1 < kMN1) {
                float2* dst = p.out + (long long)j * M + k1;
#pragma unroll
                for (int r = 0; r < 4; r++) dst[kMN1 * (16 * wlo + 4 * g + r)] = make_float2(xre[r], xim[r]);
            }
        }
        __syncthreads();      // the next frame's fold rewrites At (read in stage 1, behind the barrier above) -- and its stage 1 rewrites
                              // Bo, which this barrier puts behind every wave's stage-2 reads
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// FFT form for M = 800, D = M / 2 (BASELINE config 5; round 5): the frame's DFT as a 32 x 5 x 5 mixed-radix FFT in registers
// and LDS instead of a 25 x 32 matrix product -- 39 kflop per frame instead of 365, which moves the kernel from the vector pipe to
// its HBM traffic (40 MB in, 80 MB out per 12500 frames).  One workgroup = 4 waves = one block of 8 consecutive frames per pass:
//   fold       200 threads x 4 bins: P + 4 sample loads serve the 8 frames of a bin (consecutive frames at D = M / 2 share all
//              but every other sample); v_t[r] -> LDS, linear in r
//   stage 1    lane = n1 (25 of each 32; two frames per wave): 32-point FFT over n2 in registers, transposed in place in LDS
//   stage 2    lane = k2 (all 32; two frames per wave): twiddle, 5 x 5 DFT over n1 in registers, then for every k1 the 32 lanes store
//              32 consecutive bins: 256-byte runs of the [frame][channel] rows, two per store instruction
// The lane-level code is chan_fft_core.hpp (also compiled for the host: tests/emul/chan_emul.cpp).
// ---------------------------------------------------------------------------------------------------------------------
struct ChanFftParams {
    const void* x;         // the call's new samples (the caller's buffer) in the kernel's sample format
    const float2* hist;    // the L - 1 samples before them
    float2* out;
    const float* h;        // the prototype re-ordered for the fold [800][2][P]
    const float2* tw;      // [25][32] exp(-j 2 pi n1 k2 / 800)
    int frames, blocks, n_in;
    int ph0;
    int xcd_span;          // blocks per XCD (ceil(blocks / 8)); 0 = no remap
    int exp;               // ablation switches (profiles/measure_chan_fft.py; compile-time variants of the P = 8 kernel): 1 = one store per lane, 2 = no inter-stage twiddles
    long long abs0;
};

template <int P, int EXP = 0, int FMT = chanfft::kFmtC32> __global__ __launch_bounds__(kThreads, 2) void k_channelise_fft(ChanFftParams p) {
    using namespace chanfft;
    __shared__ c32 lds[kBlockFrames * kFrameLds];          // 52.8 KB: three workgroups per CU
    const int tid = threadIdx.x;
    BlockCtx c;
    c.x = p.x;
    c.hist = reinterpret_cast<const c32*>(p.hist);
    c.n_in = p.n_in;
    c.L = kM * P;
    c.out = reinterpret_cast<c32*>(p.out);
    c.h = p.h;
    c.tw = reinterpret_cast<const c32*>(p.tw);
    c.frames = p.frames; c.ph0 = p.ph0; c.abs0 = p.abs0;
    // One block of 8 frames per workgroup, no loop over blocks: nothing is carried from block to block, and a loop makes the compiler
    // keep the transforms' ~100 literal twiddles in VGPRs across it (256 VGPRs + spills instead of 147: three workgroups per CU).
    // Consecutive blocks share two thirds of their samples (a block reads 9600, 3200 of them new).  Workgroups are dealt to the 8 XCDs
    // round-robin, each with its own L2: with blk = blockIdx.x the three blocks that read a sample sit on three XCDs and every L2
    // fetches it again (measured: 2.9x the algorithmic read traffic).  So the XCD a workgroup lands on (blockIdx.x mod 8) takes a
    // CONTIGUOUS range of blocks, in order: neighbours in time meet in one L2.
    int blk = blockIdx.x;
    if (p.xcd_span > 0) {
        blk = (int)(blockIdx.x & 7) * p.xcd_span + (int)(blockIdx.x >> 3);
        if (blk >= p.blocks || (int)(blockIdx.x >> 3) >= p.xcd_span) return;
    }
    phase_fold<P, FMT>(c, blk, tid, lds);
    __syncthreads();
    c32 tw[kN1 - 1];
    load_twiddles(c, tid, tw);
    {
        c32 x[32];
        const bool live = phase_fft32_compute(tid, lds, x);
        // The transpose is IN PLACE: every lane of the frame's wavefront must have its 32 reads back before any lane writes.  The
        // data dependence (each output needs all 32 inputs) already orders a lane's own accesses; the wave-level fence states the
        // cross-lane half for the compiler (free on wave64: the lanes of a wavefront execute in lockstep).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (live) phase_fft32_store(tid, lds, x);
    }
    __syncthreads();
    phase_dft25_store<EXP>(c, (long long)kBlockFrames * blk, tid, lds, tw);
}

// integer samples -> complex64 (the staging buffer of the kernels that read [history | new] contiguously, and the delay line)
template <int FMT> __global__ __launch_bounds__(256) void k_chan_convert(const void* __restrict__ x, long long first, int n, float2* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const chanfft::c32 v = chanfft::load_sample<FMT>(x, first + i);
    out[i] = make_float2(v.x, v.y);
}

}  // namespace

struct tetra_chan {
    tetra_chan_config_t cfg;
    int device = 0, last_hip = 0;
    int M = 0, P = 0, D = 0, L = 0, N1 = 0, N2 = 0, max_in = 0;
    std::vector<float> proto;
    float2* xbuf = nullptr;     // [L-1 + max_in]: [history | new samples] of the call in flight
    float2* xalt = nullptr;     // same size: receives the next call's history (one copy, then the two swap roles)
    float* d_h = nullptr;
    float2 *d_w1 = nullptr, *d_w2 = nullptr, *d_wm = nullptr;
    float *d_bc = nullptr, *d_ac = nullptr;   // matrix-pipe form: the two constant block-twiddle operands (k_channelise_mfma)
    bool mfma = false;
    bool fft = false;           // M = 800, D = 400: the mixed-radix FFT kernel (k_channelise_fft)
    float2* d_tw = nullptr;     // its [25][32] inter-stage twiddles
    float* d_ht = nullptr;      // and the prototype re-ordered for its fold [800][2][P]
    int cus = 256;
    float2* st_out = nullptr;   // host-path staging
    size_t st_out_frames = 0;
    int phase = 0;              // samples consumed towards the next frame
    long long consumed = 0;     // absolute index of the next input sample
    hipEvent_t ev[2] = { nullptr, nullptr };
    bool ev_valid = false;
};

#define CH_TRY(h, expr)                                   \
    do {                                                  \
        hipError_t e__ = (expr);                          \
        if (e__ != hipSuccess) {                          \
            (h)->last_hip = (int)e__;                     \
            return TETRA_ERR_HIP;                         \
        }                                                 \
    } while (0)

namespace {

double bessel_i0(double x) {
    double s = 1.0, t = 1.0;
    for (int k = 1; k < 60; k++) {
        t *= (x / (2.0 * k)) * (x / (2.0 * k));
        s += t;
        if (t < 1e-18 * s) break;
    }
    return s;
}

// Kaiser(beta 9)-windowed sinc, cutoff fc = cutoff_rel / (2M) cycles/sample, unity DC gain.
void design_prototype(int M, int P, double cutoff_rel, std::vector<float>& h) {
    const int L = M * P;
    const double pi = 3.14159265358979323846, fc = cutoff_rel / (2.0 * (double)M), beta = 9.0;
    std::vector<double> t(L);
    double sum = 0.0;
    for (int l = 0; l < L; l++) {
        const double u = (double)l - 0.5 * (double)(L - 1);
        const double sinc = (u == 0.0) ? 2.0 * fc : std::sin(2.0 * pi * fc * u) / (pi * u);
        const double r = 2.0 * u / (double)(L - 1);
        const double w = bessel_i0(beta * std::sqrt(1.0 - r * r > 0 ? 1.0 - r * r : 0.0)) / bessel_i0(beta);
        t[l] = sinc * w;
        sum += t[l];
    }
    h.resize(L);
    for (int l = 0; l < L; l++) h[l] = (float)(t[l] / sum);
}

bool factor(int M, int& n1, int& n2) {
    int best = -1;
    for (int a = 1; a <= kMaxFactor; a++)
        if (M % a == 0 && M / a <= kMaxFactor) {
            const int bq = M / a;
            if (best < 0 || std::abs(a - bq) < std::abs(best - M / best)) best = a;
        }
    if (best < 0) return false;
    n1 = best;
    n2 = M / best;
    return true;
}

struct Guard {
    int prev = -1;
    bool ok;
    explicit Guard(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; ok = hipSetDevice(d) == hipSuccess; }
    ~Guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int upload_twiddles(tetra_chan* h) {
    const double pi = 3.14159265358979323846;
    auto tw = [&](int n) {
        std::vector<float2> w(n);
        for (int i = 0; i < n; i++) w[i] = make_float2((float)std::cos(-2.0 * pi * i / n), (float)std::sin(-2.0 * pi * i / n));
        return w;
    };
    auto w1 = tw(h->N1), w2 = tw(h->N2), wm = tw(h->M);
    CH_TRY(h, hipMemcpy(h->d_w1, w1.data(), sizeof(float2) * w1.size(), hipMemcpyHostToDevice));
    CH_TRY(h, hipMemcpy(h->d_w2, w2.data(), sizeof(float2) * w2.size(), hipMemcpyHostToDevice));
    CH_TRY(h, hipMemcpy(h->d_wm, wm.data(), sizeof(float2) * wm.size(), hipMemcpyHostToDevice));
    CH_TRY(h, hipMemcpy(h->d_h, h->proto.data(), sizeof(float) * h->proto.size(), hipMemcpyHostToDevice));
    if (h->fft) {
        std::vector<float2> t((size_t)chanfft::kN1 * chanfft::kN2);
        for (int n1 = 0; n1 < chanfft::kN1; n1++)
            for (int k2 = 0; k2 < chanfft::kN2; k2++) {
                const double a = -2.0 * pi * (double)((n1 * k2) % chanfft::kM) / chanfft::kM;
                t[(size_t)n1 * chanfft::kN2 + k2] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
        CH_TRY(h, hipMemcpy(h->d_tw, t.data(), sizeof(float2) * t.size(), hipMemcpyHostToDevice));
        std::vector<float> ht((size_t)2 * chanfft::kM * h->P);
        chanfft::fold_transpose_prototype(h->proto.data(), h->P, ht.data());
        CH_TRY(h, hipMemcpy(h->d_ht, ht.data(), sizeof(float) * ht.size(), hipMemcpyHostToDevice));
    }
    if (h->mfma) {
        // stage 1: Bc[k][col], k = n1 | N1 + n1, col = k1 | 32 + k1:  re = Vr Wr - Vi Wi, im = Vr Wi + Vi Wr, W = W_N1^(n1 k1)
        std::vector<float> bc((size_t)4 * kMK1 * 64, 0.f), ac((size_t)64 * 64, 0.f);
        for (int n1 = 0; n1 < kMN1; n1++)
            for (int k1 = 0; k1 < kMN1; k1++) {
                const double a = -2.0 * pi * (double)((n1 * k1) % kMN1) / kMN1;
                const float wr = (float)std::cos(a), wi = (float)std::sin(a);
                bc[(size_t)n1 * 64 + k1] = wr;            bc[(size_t)n1 * 64 + 32 + k1] = wi;
                bc[(size_t)(kMN1 + n1) * 64 + k1] = -wi;  bc[(size_t)(kMN1 + n1) * 64 + 32 + k1] = wr;
            }
        // stage 2: Ac[row][k], row = k2 | 32 + k2, k = n2 | 32 + n2:  re = Wr Br - Wi Bi, im = Wi Br + Wr Bi, W = W_N2^(n2 k2)
        for (int k2 = 0; k2 < kMN2; k2++)
            for (int n2 = 0; n2 < kMN2; n2++) {
                const double a = -2.0 * pi * (double)((n2 * k2) % kMN2) / kMN2;
                const float wr = (float)std::cos(a), wi = (float)std::sin(a);
                ac[(size_t)k2 * 64 + n2] = wr;            ac[(size_t)k2 * 64 + 32 + n2] = -wi;
                ac[(size_t)(32 + k2) * 64 + n2] = wi;     ac[(size_t)(32 + k2) * 64 + 32 + n2] = wr;
            }
        CH_TRY(h, hipMemcpy(h->d_bc, bc.data(), sizeof(float) * bc.size(), hipMemcpyHostToDevice));
        CH_TRY(h, hipMemcpy(h->d_ac, ac.data(), sizeof(float) * ac.size(), hipMemcpyHostToDevice));
    }
    return TETRA_OK;
}

void free_all(tetra_chan* h) {
    void* ptrs[] = { h->xbuf, h->xalt, h->d_h, h->d_w1, h->d_w2, h->d_wm, h->d_bc, h->d_ac, h->d_tw, h->d_ht, h->st_out };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
}

}  // namespace

extern "C" {

int tetra_chan_default_config(tetra_chan_config_t* cfg) {
    if (!cfg) return TETRA_ERR_ARG;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->n_channels = 800;
    cfg->taps_per_channel = 8;
    cfg->decimation = 400;
    cfg->max_in = 1 << 20;
    cfg->device = -1;
    cfg->cutoff_rel = 1.2;
    return TETRA_OK;
}

int tetra_chan_create(const tetra_chan_config_t* cfg, tetra_chan_t** out) {
    if (!cfg || !out) return TETRA_ERR_ARG;
    *out = nullptr;
    if (cfg->n_channels < 2 || cfg->taps_per_channel < 1 || cfg->taps_per_channel > 32 || cfg->decimation < 1 ||
        cfg->max_in < 1 || !(cfg->cutoff_rel > 0))
        return TETRA_ERR_ARG;
    {   // only the documented flags: a stray bit must not select anything (the ablation switches exist in the profiling build only)
        int known = TETRA_CHAN_FLAG_VALU_DFT | TETRA_CHAN_FLAG_MATRIX_DFT;
#ifdef TETRA_CHAN_EXPERIMENTS
        known |= 0x700;
#endif
        if (cfg->reserved & ~known) return TETRA_ERR_ARG;
    }
    int n1, n2;
    if (!factor(cfg->n_channels, n1, n2)) return TETRA_ERR_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TETRA_ERR_NO_DEVICE;
    int dev = cfg->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return TETRA_ERR_NO_DEVICE;
    if (dev >= ndev) return TETRA_ERR_NO_DEVICE;
    tetra_chan* h = new (std::nothrow) tetra_chan();
    if (!h) return TETRA_ERR_NOMEM;
    h->cfg = *cfg;
    h->cfg.prototype = nullptr;
    h->device = dev;
    h->M = cfg->n_channels; h->P = cfg->taps_per_channel; h->D = cfg->decimation; h->L = h->M * h->P;
    h->N1 = n1; h->N2 = n2; h->max_in = cfg->max_in;
    const bool p_ok = cfg->taps_per_channel == 8 || cfg->taps_per_channel == 6 || cfg->taps_per_channel == 4;
    h->fft = cfg->n_channels == chanfft::kM && cfg->decimation == chanfft::kM / 2 && p_ok &&
             !(cfg->reserved & (TETRA_CHAN_FLAG_VALU_DFT | TETRA_CHAN_FLAG_MATRIX_DFT));
    h->mfma = !h->fft && n1 == kMN1 && n2 == kMN2 && p_ok && !(cfg->reserved & TETRA_CHAN_FLAG_VALU_DFT);
    if (cfg->prototype) h->proto.assign(cfg->prototype, cfg->prototype + h->L);
    else design_prototype(h->M, h->P, cfg->cutoff_rel, h->proto);
    Guard g(dev);
    if (!g.ok) { delete h; return TETRA_ERR_NO_DEVICE; }
    // (the FFT kernel reads the caller's samples in place: its handles keep only the L - 1 samples of delay line, twice)
    const size_t xelems = (size_t)h->L - 1 + (h->fft ? 0 : (size_t)h->max_in);
    bool ok = hipMalloc((void**)&h->xbuf, sizeof(float2) * xelems) == hipSuccess &&
              hipMalloc((void**)&h->xalt, sizeof(float2) * xelems) == hipSuccess &&
              hipMalloc((void**)&h->d_h, sizeof(float) * h->L) == hipSuccess &&
              hipMalloc((void**)&h->d_w1, sizeof(float2) * h->N1) == hipSuccess &&
              hipMalloc((void**)&h->d_w2, sizeof(float2) * h->N2) == hipSuccess &&
              hipMalloc((void**)&h->d_wm, sizeof(float2) * h->M) == hipSuccess &&
              (!h->fft || (hipMalloc((void**)&h->d_tw, sizeof(float2) * chanfft::kN1 * chanfft::kN2) == hipSuccess &&
                           hipMalloc((void**)&h->d_ht, sizeof(float) * 2 * chanfft::kM * h->P) == hipSuccess)) &&
              (!h->mfma || (hipMalloc((void**)&h->d_bc, sizeof(float) * 4 * kMK1 * 64) == hipSuccess &&
                            hipMalloc((void**)&h->d_ac, sizeof(float) * 64 * 64) == hipSuccess)) &&
              hipEventCreate(&h->ev[0]) == hipSuccess && hipEventCreate(&h->ev[1]) == hipSuccess;
    if (ok && hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) h->cus = 256;
    int rc = ok ? upload_twiddles(h) : TETRA_ERR_NOMEM;
    if (rc == TETRA_OK && hipMemset(h->xbuf, 0, sizeof(float2) * ((size_t)h->L - 1)) != hipSuccess) rc = TETRA_ERR_HIP;
    if (rc != TETRA_OK) { free_all(h); delete h; return rc; }
    *out = h;
    return TETRA_OK;
}

int tetra_chan_destroy(tetra_chan_t* h) {
    if (!h) return TETRA_ERR_ARG;
    Guard g(h->device);
    (void)hipDeviceSynchronize();
    free_all(h);
    delete h;
    return TETRA_OK;
}

int tetra_chan_frames_for(tetra_chan_t* h, int n_in) {
    if (!h || n_in < 0) return TETRA_ERR_ARG;
    return (h->phase + n_in) / h->D;
}

}  // extern "C"

namespace {
// bytes per sample of a format; copy `n` samples starting at `first` of the caller's buffer into a complex64 buffer
constexpr int kFmtBytes[3] = { 8, 4, 2 };
int copy_samples(tetra_chan* h, int fmt, const void* d_x, size_t first, size_t n, float2* dst, hipStream_t s) {
    if (n == 0) return TETRA_OK;
    if (fmt == chanfft::kFmtC32) {
        CH_TRY(h, hipMemcpyAsync(dst, static_cast<const float2*>(d_x) + first, sizeof(float2) * n, hipMemcpyDeviceToDevice, s));
        return TETRA_OK;
    }
    const dim3 grid((unsigned)((n + 255) / 256));
    if (fmt == chanfft::kFmtCs16) hipLaunchKernelGGL(k_chan_convert<chanfft::kFmtCs16>, grid, dim3(256), 0, s, d_x, (long long)first, (int)n, dst);
    else hipLaunchKernelGGL(k_chan_convert<chanfft::kFmtCs8>, grid, dim3(256), 0, s, d_x, (long long)first, (int)n, dst);
    CH_TRY(h, hipGetLastError());
    return TETRA_OK;
}
template <int FMT> void launch_fft(tetra_chan* h, const ChanFftParams& p, dim3 grid, hipStream_t s);

int process_any(tetra_chan_t* h, int fmt, const void* d_x, int n_in, float* d_out, int* n_frames, void* hip_stream) {
    if (!h || (!d_x && n_in > 0) || !d_out || !n_frames) return TETRA_ERR_ARG;
    if (n_in < 0 || n_in > h->max_in) return TETRA_ERR_SIZE;
    // samples are moved as whole units (the FFT kernel reads d_x in place with 8- / 4- / 2-byte loads; every kernel stores 8-byte units)
    if (((uintptr_t)d_x & (kFmtBytes[fmt] - 1)) || ((uintptr_t)d_out & 7)) return TETRA_ERR_ALIGN;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)hip_stream;
    const int frames = (h->phase + n_in) / h->D;
    *n_frames = frames;
    const size_t hist = (size_t)h->L - 1;
    // The FFT kernel reads the new samples where the caller left them (and the L - 1 before them from the handle's delay line): no
    // staging copy -- every sample crosses HBM once.  The other two kernels index one contiguous [history | new] buffer.
    if (n_in > 0 && !h->fft) {
        const int rc = copy_samples(h, fmt, d_x, 0, (size_t)n_in, h->xbuf + hist, s);
        if (rc != TETRA_OK) return rc;
    }
    CH_TRY(h, hipEventRecord(h->ev[0], s));
    if (frames > 0 && h->fft) {
        ChanFftParams p;
        p.x = d_x; p.hist = h->xbuf; p.out = reinterpret_cast<float2*>(d_out); p.h = h->d_ht; p.tw = h->d_tw;
        p.frames = frames; p.blocks = (frames + chanfft::kBlockFrames - 1) / chanfft::kBlockFrames; p.n_in = n_in;
        p.ph0 = h->phase; p.abs0 = h->consumed;
        // one block of 8 frames per workgroup: the hardware hands the next block to whichever CU is through first
        p.xcd_span = (p.blocks + 7) / 8;
        p.exp = 0;
#ifdef TETRA_CHAN_EXPERIMENTS
        // Ablation switches of the profiling build ONLY (profiles/build_exp.sh defines the macro; the product library rejects these
        // bits in tetra_chan_create): 0x100 no XCD remap, 0x200 one store per lane instead of 25, 0x400 no inter-stage twiddles --
        // the last two produce WRONG spectra by design.
        if (h->cfg.reserved & 0x100) p.xcd_span = 0;
        p.exp = (h->cfg.reserved >> 9) & 3;
#endif
        const dim3 grid(p.xcd_span > 0 ? 8 * p.xcd_span : p.blocks);
#ifdef TETRA_CHAN_EXPERIMENTS
        if (fmt == chanfft::kFmtC32 && h->P == 8 && p.exp == 1) hipLaunchKernelGGL((k_channelise_fft<8, 1>), grid, dim3(kThreads), 0, s, p);
        else if (fmt == chanfft::kFmtC32 && h->P == 8 && p.exp == 2) hipLaunchKernelGGL((k_channelise_fft<8, 2>), grid, dim3(kThreads), 0, s, p);
        else if (fmt == chanfft::kFmtC32 && h->P == 8 && p.exp == 3) hipLaunchKernelGGL((k_channelise_fft<8, 3>), grid, dim3(kThreads), 0, s, p);
        else
#endif
        if (fmt == chanfft::kFmtCs16) launch_fft<chanfft::kFmtCs16>(h, p, grid, s);
        else if (fmt == chanfft::kFmtCs8) launch_fft<chanfft::kFmtCs8>(h, p, grid, s);
        else launch_fft<chanfft::kFmtC32>(h, p, grid, s);
        CH_TRY(h, hipGetLastError());
    } else if (frames > 0 && h->mfma) {
        ChanMfmaParams p;
        p.xbuf = h->xbuf; p.out = reinterpret_cast<float2*>(d_out); p.h = h->d_h; p.bc = h->d_bc; p.ac = h->d_ac; p.wm = h->d_wm;
        p.P = h->P; p.D = h->D; p.frames = frames; p.ph0 = h->phase; p.abs0 = h->consumed;
        // workgroups loop over frames (the constant operands are loaded once per workgroup): a few per CU keep the matrix pipe,
        // the fold's loads and the stores of different frames overlapping
        const int grid = frames < 3 * h->cus ? frames : 3 * h->cus;      // three workgroups fit a CU (registers, LDS); each loops over its frames
        if (h->P == 8) hipLaunchKernelGGL(k_channelise_mfma<8>, dim3(grid), dim3(kThreads), 0, s, p);
        else if (h->P == 6) hipLaunchKernelGGL(k_channelise_mfma<6>, dim3(grid), dim3(kThreads), 0, s, p);
        else hipLaunchKernelGGL(k_channelise_mfma<4>, dim3(grid), dim3(kThreads), 0, s, p);
        CH_TRY(h, hipGetLastError());
    } else if (frames > 0) {
        ChanParams p;
        p.xbuf = h->xbuf; p.out = reinterpret_cast<float2*>(d_out); p.h = h->d_h;
        p.w1 = h->d_w1; p.w2 = h->d_w2; p.wm = h->d_wm;
        p.M = h->M; p.P = h->P; p.D = h->D; p.N1 = h->N1; p.N2 = h->N2;
        p.ph0 = h->phase; p.abs0 = h->consumed;
        hipLaunchKernelGGL(k_channelise, dim3(frames), dim3(kThreads), sizeof(float2) * ((size_t)h->M + (size_t)h->N1 * (h->N2 + 1)), s, p);
        CH_TRY(h, hipGetLastError());
    }
    CH_TRY(h, hipEventRecord(h->ev[1], s));
    h->ev_valid = true;
    // carry: the last L-1 samples of [history | new] become the next call's history -- ONE copy into the other buffer
    // (whatever n_in is; an in-place move would overlap for n_in < L-1), then the buffers swap roles.  Stream order keeps the
    // kernel above ahead of the copy and the copy ahead of the next call's writes.
    if (n_in > 0 && h->fft) {
        // the same from the two places the samples live in: what is left of the old delay line, then the tail of the caller's buffer
        const size_t from_x = (size_t)n_in < hist ? (size_t)n_in : hist, keep = hist - from_x;
        if (keep) CH_TRY(h, hipMemcpyAsync(h->xalt, h->xbuf + n_in, sizeof(float2) * keep, hipMemcpyDeviceToDevice, s));
        const int rc = copy_samples(h, fmt, d_x, (size_t)n_in - from_x, from_x, h->xalt + keep, s);
        if (rc != TETRA_OK) return rc;
        float2* t = h->xbuf; h->xbuf = h->xalt; h->xalt = t;
    } else if (n_in > 0) {
        CH_TRY(h, hipMemcpyAsync(h->xalt, h->xbuf + n_in, sizeof(float2) * hist, hipMemcpyDeviceToDevice, s));
        float2* t = h->xbuf; h->xbuf = h->xalt; h->xalt = t;
    }
    h->phase = (h->phase + n_in) % h->D;
    h->consumed += n_in;
    return TETRA_OK;
}
template <int FMT> void launch_fft(tetra_chan* h, const ChanFftParams& p, dim3 grid, hipStream_t s) {
    if (h->P == 8) hipLaunchKernelGGL((k_channelise_fft<8, 0, FMT>), grid, dim3(kThreads), 0, s, p);
    else if (h->P == 6) hipLaunchKernelGGL((k_channelise_fft<6, 0, FMT>), grid, dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL((k_channelise_fft<4, 0, FMT>), grid, dim3(kThreads), 0, s, p);
}
}  // namespace

extern "C" {

int tetra_chan_process_device(tetra_chan_t* h, const float* d_x, int n_in, float* d_out, int* n_frames, void* hip_stream) {
    return process_any(h, chanfft::kFmtC32, d_x, n_in, d_out, n_frames, hip_stream);
}
int tetra_chan_process_device_cs16(tetra_chan_t* h, const int16_t* d_x, int n_in, float* d_out, int* n_frames, void* hip_stream) {
    return process_any(h, chanfft::kFmtCs16, d_x, n_in, d_out, n_frames, hip_stream);
}
int tetra_chan_process_device_cs8(tetra_chan_t* h, const int8_t* d_x, int n_in, float* d_out, int* n_frames, void* hip_stream) {
    return process_any(h, chanfft::kFmtCs8, d_x, n_in, d_out, n_frames, hip_stream);
}

int tetra_chan_process(tetra_chan_t* h, const float* x, int n_in, float* out, int* n_frames) {
    if (!h || (!x && n_in > 0) || !out || !n_frames) return TETRA_ERR_ARG;
    if (n_in < 0 || n_in > h->max_in) return TETRA_ERR_SIZE;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    const size_t frames = (size_t)((h->phase + n_in) / h->D);
    if (frames > h->st_out_frames) {
        if (h->st_out) (void)hipFree(h->st_out);
        h->st_out = nullptr; h->st_out_frames = 0;
        CH_TRY(h, hipMalloc((void**)&h->st_out, sizeof(float2) * frames * (size_t)h->M));
        h->st_out_frames = frames;
    }
    struct Tmp {                      // freed on every return path
        float2* p = nullptr;
        ~Tmp() { if (p) (void)hipFree(p); }
    } d_x, d_dummy;
    if (n_in > 0) {
        CH_TRY(h, hipMalloc((void**)&d_x.p, sizeof(float2) * (size_t)n_in));
        CH_TRY(h, hipMemcpy(d_x.p, x, sizeof(float2) * (size_t)n_in, hipMemcpyHostToDevice));
    }
    if (!h->st_out) CH_TRY(h, hipMalloc((void**)&d_dummy.p, sizeof(float2)));
    int rc = tetra_chan_process_device(h, reinterpret_cast<const float*>(d_x.p), n_in,
                                       reinterpret_cast<float*>(h->st_out ? h->st_out : d_dummy.p), n_frames, nullptr);
    if (rc == TETRA_OK) {
        hipError_t e = hipStreamSynchronize(0);
        if (e == hipSuccess && frames) e = hipMemcpy(out, h->st_out, sizeof(float2) * frames * (size_t)h->M, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { h->last_hip = (int)e; rc = TETRA_ERR_HIP; }
    }
    return rc;
}

int tetra_chan_reset(tetra_chan_t* h) {
    if (!h) return TETRA_ERR_ARG;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    CH_TRY(h, hipDeviceSynchronize());
    CH_TRY(h, hipMemset(h->xbuf, 0, sizeof(float2) * ((size_t)h->L - 1)));
    h->phase = 0;
    h->consumed = 0;
    return TETRA_OK;
}

int tetra_chan_get_prototype(tetra_chan_t* h, float* proto) {
    if (!h || !proto) return TETRA_ERR_ARG;
    std::memcpy(proto, h->proto.data(), sizeof(float) * h->proto.size());
    return TETRA_OK;
}

int tetra_chan_last_kernel_ms(tetra_chan_t* h, float* ms) {
    if (!h || !ms || !h->ev_valid) return TETRA_ERR_ARG;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    CH_TRY(h, hipEventSynchronize(h->ev[1]));
    CH_TRY(h, hipEventElapsedTime(ms, h->ev[0], h->ev[1]));
    return TETRA_OK;
}

}  // extern "C"
