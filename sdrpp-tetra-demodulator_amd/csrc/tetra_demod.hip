// tetra_demod.hip -- HIP kernels (gfx950) + C ABI of the batched TETRA pi/4-DQPSK demodulator.
//
// The whole chain runs in ONE kernel, k_fused (kernel_fused.hpp): AGC -> band-edge FLL -> RRC matched filter -> ML timing
// recovery -> pi/4 Costas -> slicer -> differential decoder -> bit unpacker, as specialised waves connected by LDS rings.  Three
// workgroup shapes of the one template: 16 channels in six waves (one workgroup per CU up to 4096 channels), 32 channels in
// eight waves (more than 16 channels per CU) and 4 channels (at most 4 channels per CU); tetra_demod_create plans which
// channels take which shape, the results are identical bit for bit.  Filters of 73 .. 129 taps take the LONG variant of the 4- or
// the 16-channel shape (FLL rows of 16 x 9 / 8 x 17 taps, 128 delay-line samples); timing loops below 0.07 samples per symbol -- and, on request, the
// long filters -- run in k_generic (kernel_generic.hpp): one lane per channel, same arithmetic.  (The two-kernel pipeline of
// round 1 -- k1_agc_fll_rrc / k2_sync_slice with an HBM scratch in between -- was retired in ABI 2; `git log` has it.)
// Reference path replaced: src/dsp/pi4dqpsk.cpp:132-140, src/dsp/dqpsk_sym_extr.cpp:4-55,
// src/dsp/bit_unpacker.cpp:4-10 (see include/tetra_demod.h).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/tetra_demod.h"
#include "constellation_core.hpp"
#include "demod_core.hpp"
#include "design.hpp"

using namespace tdm;

namespace {

constexpr int kTinyCallSamples = 768;           // tetra_demod_process: calls of at most this many samples per channel run in place (below)
constexpr int kYHist = kInterpTaps - 1;         // COMPLEX_FD's delay buffer: 7 RRC outputs in front of the new ones
constexpr int kWg16ClocksPerSample = 258;       // measured shader clocks per sample of one workgroup round: 3.86 ms per 36000 samples (profiles/r03)
constexpr int kWg32ClocksPerSample = 348;       // 32-channel workgroup: 5.21 ms per 36000 samples
constexpr int kWg4ClocksPerSample = 223;        // 4-channel workgroup: 3.33 ms per 36000 samples

__device__ __forceinline__ Pair<float> ld_pair(const float2* p) {
    float2 v = *p;
    return Pair<float>(v.x, v.y);
}

}  // namespace

#include "kernel_fused.hpp"
#include "kernel_generic.hpp"

namespace {

// Device self-test of the primitives the arithmetic contract rests on (DPP row moves, sqrt, sincos).
// out[0][l] = row_shr1(old = 100+l, src = l), out[1][l] = row_shl1(old = 200+l, src = l),
// out[2][l] = sqrt(in[l]), out[3][l] / out[4][l] = sin / cos of in[64 + l].
__global__ void k_selftest(const float* in, float* out) {
    const int l = threadIdx.x;
    out[l] = row_shr1(100.0f + (float)l, (float)l);
    out[64 + l] = row_shl1(200.0f + (float)l, (float)l);
    out[128 + l] = v_sqrt(in[l]);
    float s, c;
    sincos_t<float>(in[64 + l], s, c);
    out[192 + l] = s;
    out[256 + l] = c;
}

// Matrix-pipe self-test: D = A . B accumulated over K in ASCENDING k by chained
// v_mfma_f32_16x16x4_f32 (M = N = 16) or v_mfma_f32_32x32x2_f32 (M = N = 32) from C = +0 -- to be compared on the host with
// the oracle's fmaf chain `for k: acc = fmaf(A[i][k], B[k][j], acc)`, bit for bit (incl. zero taps, -0, subnormals).
// A [M][K] row-major, B [K][N] row-major, D [M][N] row-major; K a multiple of 4; one wave.
typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));
typedef float mfma_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void k_mfma_selftest(int shape, int K, const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ D) {
    const int l = threadIdx.x;
    if (shape == 16) {
        mfma_f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
        for (int kb = 0; kb < K; kb += 4) {
            const float a = A[(l & 15) * K + kb + (l >> 4)];
            const float b = B[(kb + (l >> 4)) * 16 + (l & 15)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; r++) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
    } else {
        mfma_f32x16 acc;
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        for (int kb = 0; kb < K; kb += 2) {
            const float a = A[(l & 31) * K + kb + (l >> 5)];
            const float b = B[(kb + (l >> 5)) * 32 + (l & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
    }
}

// Small helper kernels for state management.
__global__ void k_fill_f32(float* p, float v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_fill_i32(int* p, int v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// Host-path helpers (tetra_demod_process_async).
// int16 IQ as most SDR hardware delivers it -> the complex float the chain computes on; x / 32768 is exact in binary32.
__global__ void k_cs16_to_cf32(const short2* __restrict__ in, float2* __restrict__ out, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const short2 v = in[i];
        out[i] = make_float2((float)v.x * (1.0f / 32768.0f), (float)v.y * (1.0f / 32768.0f));
    }
}
// int8 IQ (RTL-SDR / HackRF class front-ends): x / 128, exact.
__global__ void k_cs8_to_cf32(const char2* __restrict__ in, float2* __restrict__ out, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const char2 v = in[i];
        out[i] = make_float2((float)v.x * (1.0f / 128.0f), (float)v.y * (1.0f / 128.0f));
    }
}
// One time chunk's bits appended to the call's output rows: out[c][out_n[c] ..] = chunk[c][0 .. chunk_n[c]).
__global__ void k_append_bits(const uint8_t* __restrict__ chunk, int chunk_stride, const int* __restrict__ chunk_n,
                              uint8_t* __restrict__ out, int out_stride, int* __restrict__ out_n) {
    const int c = blockIdx.x;
    const int at = out_n[c];
    int n = chunk_n[c];
    n = at + n > out_stride ? out_stride - at : n;
    const uint8_t* src = chunk + (size_t)c * chunk_stride;
    uint8_t* dst = out + (size_t)c * out_stride + at;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) out_n[c] = at + n;
}
// TETRA_FLAG_QUALITY: DQPSKSymbolExtractor's statistic (dqpsk_sym_extr.cpp:8-31) brought up to date after a launch, one
// 64-lane workgroup per channel.  The reference pushes one angular distance per symbol into a 4096-entry ring and, every 256
// symbols, publishes the ring's mean -- summed in float, in ring-index order (:19-23).  Only the value at the LAST such
// boundary of a call can be observed, so the workgroup rebuilds the ring as it stood at that boundary in LDS (old entries,
// overwritten by this call's distances up to the boundary), one lane adds it up in exactly the reference's order, then the
// image and the few symbols behind the boundary go to the ring in memory and the two counters advance.
__global__ __launch_bounds__(64) void k_quality(const float2* __restrict__ sym, long long sym_stride, const int* __restrict__ n_bits,
                                                float* __restrict__ ring, int* __restrict__ q_ptr, int* __restrict__ q_disp,
                                                float* __restrict__ q_err, int* __restrict__ q_sync) {
    __shared__ __attribute__((aligned(16))) float img[4096];
    const int c = blockIdx.x, lane = threadIdx.x;
    const float2* z = sym + (long long)c * sym_stride;
    float* r = ring + (long long)c * 4096;
    const int n = n_bits[c] / 2, ptr0 = q_ptr[c], disp0 = q_disp[c];
    const int total = disp0 + n;
    const int b = total >= 256 ? n - (total & 255) : 0;     // symbols of this call consumed at the last boundary (0 = none)
    if (b > 0) {
        if (b < 4096)                                       // entries this call has not reached by then keep their old value
            for (int i = lane; i < 4096; i += 64) img[i] = r[i];
        __syncthreads();
        for (int j = (b > 4096 ? b - 4096 : 0) + lane; j < b; j += 64) img[(ptr0 + j) & 4095] = tdm::quality_distance(z[j].x, z[j].y);
        __syncthreads();
        if (lane == 0) {
            float xerr = 0.0f;                              // dqpsk_sym_extr.cpp:20-23: float accumulator, index order
            const float4* q = reinterpret_cast<const float4*>(img);
            for (int i0 = 0; i0 < 1024; i0 += 16) {        // sixteen LDS loads in flight, then their 64 adds in order
                float4 v[16];
                _Pragma("unroll")
                for (int k = 0; k < 16; k++) v[k] = q[i0 + k];
                _Pragma("unroll")
                for (int k = 0; k < 16; k++) { xerr += v[k].x; xerr += v[k].y; xerr += v[k].z; xerr += v[k].w; }
            }
            xerr = xerr / 4096.0f;
            q_err[c] = xerr;
            q_sync[c] = xerr >= 0.35f ? 0 : 1;
        }
        for (int i = lane; i < 4096; i += 64) r[i] = img[i];        // the ring at the boundary ...
        __syncthreads();
    }
    // ... and the (fewer than 256) symbols behind it
    for (int j = b + lane; j < n; j += 64) r[(ptr0 + j) & 4095] = tdm::quality_distance(z[j].x, z[j].y);
    if (lane == 0) {
        q_ptr[c] = (ptr0 + n) & 4095;
        q_disp[c] = total & 255;
    }
}

// TETRA_FLAG_CONSTELLATION: the plugin's constellation tap (src/main.cpp:85-89: Reshaper keep 1024 / skip 0 -> :376-383 copies each
// 1024-symbol block to the diagram) brought up to date after a launch, one workgroup per channel (constellation_core.hpp).
constexpr int kCdSyms = tetra_cd::kSyms;
static_assert(kCdSyms == TETRA_CONSTELLATION_SYMBOLS, "header and kernel agree on the block length");
__global__ __launch_bounds__(256) void k_constellation(const float2* __restrict__ sym, long long sym_stride, const int* __restrict__ n_bits,
                                                       float2* __restrict__ blk, float2* __restrict__ part, int* __restrict__ fill,
                                                       int* __restrict__ blocks) {
    const int c = blockIdx.x, tid = threadIdx.x;
    const float2* z = sym + (long long)c * sym_stride;
    float2* B = blk + (long long)c * kCdSyms;
    float2* P = part + (long long)c * kCdSyms;
    const int n = n_bits[c] / 2, f0 = fill[c];
    const tetra_cd::Plan p = tetra_cd::plan(f0, n);
    tetra_cd::assemble_block(p, tid, 256, z, P, B);
    __syncthreads();
    tetra_cd::carry_partial(p, f0, n, tid, 256, z, P);
    if (tid == 0) { fill[c] = p.r; blocks[c] += p.nb; }
}

__global__ void k_min_i32(int* p, int v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i] > v) p[i] = v;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Host side: handle + C ABI
// ------------------------------------------------------------------------------------------------
struct tetra_demod {
    tetra_demod_config_t cfg;
    host::DesignParams dp;
    host::Design design;
    int device = 0;
    int last_hip = 0;
    int C = 0;
    int max_samples = 0;
    // device memory
    float *agc_g = nullptr, *fll_ph = nullptr, *fll_fr = nullptr;
    float2* hist = nullptr;
    float2* hist_far = nullptr;     // [C][48]: the delay-line samples before hist's 80 (the fused kernel's long rows and the generic kernel)
    bool far_valid = true;          // false once the fused kernel has run since hist_far was written: it then reads as zeros
    float2 *g_xs = nullptr, *g_ys = nullptr;      // generic kernel's scratch: [C][128 + max_samples] FLL outputs, [C][7 + max_samples] RRC outputs
    float *d_g_be_a = nullptr, *d_g_be_b = nullptr, *d_g_rrc = nullptr;   // un-padded tap tables for it, [kGenMaxTaps] each
    float *mu = nullptr, *omega = nullptr, *cph = nullptr, *cfr = nullptr, *ph2 = nullptr;
    int *offset = nullptr, *prev = nullptr;
    int n_wide = 0;             // channels [0, n_wide) run in 32-channel workgroups, [n_wide, C) in 16-channel ones ...
    bool small = false;         // ... or, when they are at most 4 per CU (or the flag forces it), in 4-channel ones
    bool force_small = false;   // TETRA_FLAG_SMALL_WORKGROUPS
    bool force_generic = false; // TETRA_FLAG_GENERIC_KERNEL
    bool force_shape = false;   // TETRA_FLAG_WIDE_WORKGROUPS / _NARROW_: the caller chose
    int cus = 256;
    int* rrc_valid = nullptr;   // [C] delay-line samples the RRC may see (tetra_demod.h: tetra_demod_channel_state.rrc_valid)
    float2* y = nullptr;        // TETRA_FLAG_KEEP_RRC_OUT: time-major RRC output scratch [(7 + max_samples)][C]
    float2* ybuf = nullptr;     // COMPLEX_FD delay buffer [C][7]
    int* d_overruns = nullptr;  // [1] channels cut off at their row capacity, counted by the kernels since create
    long long overruns_seen = 0;   // ... and what the host entry points have already reported of it
    int* cut_flag = nullptr;    // set around an in-place call's launch: a cut-off channel also stores 1 here (mapped host memory)
    bool tn_disabled = false;   // the platform refused the mapped, coherent host blocks: short calls keep the copy-engine path
    float* q_ring = nullptr;    // TETRA_FLAG_QUALITY: [C][4096] distance ring + per-channel state (k_quality)
    int *q_ptr = nullptr, *q_disp = nullptr, *q_sync = nullptr;
    float* q_err = nullptr;
    float2* q_sym = nullptr;    // [C][q_sym_stride] symbols of the last launch when the caller did not ask for them
    float2 *cd_blk = nullptr, *cd_part = nullptr;   // TETRA_FLAG_CONSTELLATION: [C][1024] last complete / partial block (k_constellation)
    int *cd_fill = nullptr, *cd_blocks = nullptr;   // [C] symbols in the partial block, blocks completed
    bool taps_sym() const { return q_ring || cd_blk; }      // a post-launch kernel reads the launch's symbols
    long long q_sym_stride = 0;
    bool user_rrc = false, user_be = false;   // caller-supplied FIR tables (cfg.rrc_taps / cfg.bandedge_taps)
    bool quirks = false;        // TETRA_FLAG_REFERENCE_QUIRKS
    bool keep_y = false;        // y scratch allocated
    float* d_bank = nullptr;
    float *d_be_re80 = nullptr, *d_be_im80 = nullptr, *d_rrc_ext = nullptr;   // band-edge taps zero-padded (old end) to 80, RRC zero-extended
    // host-path staging
    float* st_iq = nullptr;
    uint8_t* st_bits = nullptr;
    int* st_nbits = nullptr;
    float* st_sym = nullptr;
    size_t st_iq_bytes = 0, st_bits_bytes = 0, st_sym_bytes = 0;
    // small synchronous calls (the single-channel drop-in's 180-sample chunks): page-locked host staging, one packed output
    uint8_t *pk_dev = nullptr, *pk_host = nullptr, *pk_in = nullptr;
    size_t pk_bytes = 0, pk_in_bytes = 0;
    // the smallest synchronous calls: page-locked, mapped, coherent blocks the kernels read / write in place (no copy engine)
    uint8_t *tn_out = nullptr, *tn_in = nullptr, *tn_out_dev = nullptr, *tn_in_dev = nullptr;
    // ring of HIP-event pairs (before / after the call's launches), one slot per process call
    static constexpr int kEvSlots = 64;
    hipEvent_t ev[kEvSlots][2] = {};
    long long n_calls = 0;      // process calls that launched kernels
    hipStream_t own_stream = nullptr;   // tetra_demod_process_resident: the handle's own (non-blocking) stream
    long long* d_prof = nullptr;   // TETRA_DEMOD_PROFILE scratch
    int last_n = 0;
    // tetra_demod_process_async: three streams, time chunks double-buffered in HBM (see the function)
    struct Async {
        bool ready = false;
        hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
        hipEvent_t ev_in[2] = {}, ev_free[2] = {}, ev_done[2] = {}, ev_out[2] = {};
        void* d_raw[2] = {};        // int16 input only: the chunk as it came
        float* d_iq[2] = {};        // the chunk as complex float [C][chunk] (or [chunk][C])
        uint8_t* d_cbits[2] = {};   // the chunk's bits [C][chunk_stride]
        int* d_cnb[2] = {};
        uint8_t* d_out[2] = {};     // a call's bits [C][bits_stride]; two calls may be in flight
        int* d_onb[2] = {};
        size_t raw_bytes = 0, iq_bytes = 0, cbits_bytes = 0, out_bytes = 0;
        long long chunks = 0, calls = 0;
    } as;
};

#define HIP_TRY(h, expr)                                  \
    do {                                                  \
        hipError_t e__ = (expr);                          \
        if (e__ != hipSuccess) {                          \
            (h)->last_hip = (int)e__;                     \
            return TETRA_ERR_HIP;                         \
        }                                                 \
    } while (0)

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int upload_tables(tetra_demod* h) {
    HIP_TRY(h, hipMemcpy(h->d_bank, h->design.bank.data(), sizeof(float) * kInterpPhases * kInterpTaps,
                         hipMemcpyHostToDevice));
    {   // the generic kernel's un-padded tables (any accepted tap count)
        std::vector<float> a(kGenMaxTaps, 0.f), b(kGenMaxTaps, 0.f), r(kGenMaxTaps, 0.f);
        for (int k = 0; k < h->design.ntaps_be; k++) { a[k] = h->design.be_re[k]; b[k] = h->design.be_im[k]; }
        for (int k = 0; k < h->design.ntaps; k++) r[k] = h->design.rrc[k];
        HIP_TRY(h, hipMemcpy(h->d_g_be_a, a.data(), sizeof(float) * kGenMaxTaps, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_g_be_b, b.data(), sizeof(float) * kGenMaxTaps, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_g_rrc, r.data(), sizeof(float) * kGenMaxTaps, hipMemcpyHostToDevice));
    }
    if (h->design.ntaps <= kF8Pad && h->design.ntaps_be <= kF8Pad) {
        std::vector<float> re72(kBePad, 0.f), im72(kBePad, 0.f), rrx(kRrcExt, 0.f);
        const int o72 = kBePad - h->design.ntaps_be;
        const int rpad = (8 - ((h->design.ntaps - 1) & 7)) & 7;     // RRC windows start on a multiple of 8, see kernel_fused.hpp
        for (int k = 0; k < h->design.ntaps_be; k++) {
            re72[o72 + k] = h->design.be_re[k];
            im72[o72 + k] = h->design.be_im[k];
        }
        for (int k = 0; k < h->design.ntaps; k++) rrx[7 + rpad + k] = h->design.rrc[k];
        HIP_TRY(h, hipMemcpy(h->d_be_re80, re72.data(), sizeof(float) * kBePad, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_be_im80, im72.data(), sizeof(float) * kBePad, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_rrc_ext, rrx.data(), sizeof(float) * kRrcExt, hipMemcpyHostToDevice));
    } else {
        // filters of 73 .. 129 taps: the same tables in the long rows' sizes (kernel_fused.hpp: LONG)
        std::vector<float> re(kBePadLong, 0.f), im(kBePadLong, 0.f), rrx(kRrcExtLong, 0.f);
        const int o = kBePadLong - h->design.ntaps_be;
        const int rpad = (8 - ((h->design.ntaps - 1) & 7)) & 7;
        for (int k = 0; k < h->design.ntaps_be; k++) {
            re[o + k] = h->design.be_re[k];
            im[o + k] = h->design.be_im[k];
        }
        for (int k = 0; k < h->design.ntaps; k++) rrx[7 + rpad + k] = h->design.rrc[k];
        HIP_TRY(h, hipMemcpy(h->d_be_re80, re.data(), sizeof(float) * kBePadLong, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_be_im80, im.data(), sizeof(float) * kBePadLong, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_rrc_ext, rrx.data(), sizeof(float) * kRrcExtLong, hipMemcpyHostToDevice));
    }
    return TETRA_OK;
}

int fill(tetra_demod* h, float* p, float v, int first, int count) {
    if (count <= 0) return TETRA_OK;
    hipLaunchKernelGGL(k_fill_f32, dim3((count + 255) / 256), dim3(256), 0, 0, p + first, v, count);
    HIP_TRY(h, hipGetLastError());
    return TETRA_OK;
}

// reset of the timing-recovery loop only (COMPLEX_FD::reset / setOmega, complex_fd.cpp:30-41,78-87)
int reset_timing(tetra_demod* h, int first, int count) {
    int rc;
    if ((rc = fill(h, h->mu, 0.0f, first, count))) return rc;
    if ((rc = fill(h, h->omega, h->design.tr_omega, first, count))) return rc;
    HIP_TRY(h, hipMemsetAsync(h->offset + first, 0, sizeof(int) * count, 0));
    return TETRA_OK;
}

// PI4DQPSK::reset (pi4dqpsk.cpp:120-130) for channels [first, first+count); `fresh` = also everything the reference's
// reset leaves alone (a new handle, and every reset without TETRA_FLAG_REFERENCE_QUIRKS).
int reset_range(tetra_demod* h, int first, int count, bool fresh) {
    int rc;
    // FastAGC::reset -> initGain 1.0; FLL::reset fll.cpp:120-127; FIR::reset clears the delay line (ONE delay line for the
    // three FIRs here, see tetra_demod.h); PLL::reset; COMPLEX_FD::reset complex_fd.cpp:78-87
    if ((rc = fill(h, h->agc_g, 1.0f, first, count))) return rc;
    HIP_TRY(h, hipMemsetAsync(h->fll_ph + first, 0, sizeof(float) * count, 0));
    HIP_TRY(h, hipMemsetAsync(h->fll_fr + first, 0, sizeof(float) * count, 0));
    // FIR::reset: the reference clears the RRC's delay line only; FLL::reset leaves the band-edge FIRs' lines alone.  To the
    // letter (quirks, fused pipeline) the shared line therefore stays and the RRC is told to see none of it.
    if (fresh) {
        HIP_TRY(h, hipMemsetAsync(h->hist + (size_t)first * kHist, 0, sizeof(float2) * kHist * (size_t)count, 0));
        HIP_TRY(h, hipMemsetAsync(h->hist_far + (size_t)first * (kGenHist - kHist), 0, sizeof(float2) * (kGenHist - kHist) * (size_t)count, 0));
        // "all visible" = the longest delay line any kernel keeps (128); the regular rows saturate the count at their own 80
        hipLaunchKernelGGL(k_fill_i32, dim3((count + 255) / 256), dim3(256), 0, 0, h->rrc_valid + first, (int)kGenHist, count);
        HIP_TRY(h, hipGetLastError());
    } else {
        HIP_TRY(h, hipMemsetAsync(h->rrc_valid + first, 0, sizeof(int) * count, 0));
    }
    if ((rc = reset_timing(h, first, count))) return rc;
    HIP_TRY(h, hipMemsetAsync(h->cph + first, 0, sizeof(float) * count, 0));
    HIP_TRY(h, hipMemsetAsync(h->cfr + first, 0, sizeof(float) * count, 0));
    if (fresh) {
        // not touched by the reference's reset: ph2 (a plain member, pi4dqpsk_costas.h:32), the slicer's previous symbol
        // and statistic (DQPSKSymbolExtractor is another block), COMPLEX_FD's delay buffer
        HIP_TRY(h, hipMemsetAsync(h->ph2 + first, 0, sizeof(float) * count, 0));
        HIP_TRY(h, hipMemsetAsync(h->prev + first, 0, sizeof(int) * count, 0));
        if (h->q_ring) {
            HIP_TRY(h, hipMemsetAsync(h->q_ring + (size_t)first * 4096, 0, sizeof(float) * 4096 * (size_t)count, 0));
            HIP_TRY(h, hipMemsetAsync(h->q_ptr + first, 0, sizeof(int) * count, 0));
            HIP_TRY(h, hipMemsetAsync(h->q_disp + first, 0, sizeof(int) * count, 0));
            HIP_TRY(h, hipMemsetAsync(h->q_sync + first, 0, sizeof(int) * count, 0));
            HIP_TRY(h, hipMemsetAsync(h->q_err + first, 0, sizeof(float) * count, 0));
        }
        if (h->cd_blk) {                // (the Reshaper behind the symbol stream is another block too)
            HIP_TRY(h, hipMemsetAsync(h->cd_blk + (size_t)first * kCdSyms, 0, sizeof(float2) * kCdSyms * (size_t)count, 0));
            HIP_TRY(h, hipMemsetAsync(h->cd_fill + first, 0, sizeof(int) * count, 0));
            HIP_TRY(h, hipMemsetAsync(h->cd_blocks + first, 0, sizeof(int) * count, 0));
        }
        HIP_TRY(h, hipMemsetAsync(h->ybuf + (size_t)first * kYHist, 0, sizeof(float2) * kYHist * (size_t)count, 0));
    }
    HIP_TRY(h, hipStreamSynchronize(0));
    return TETRA_OK;
}

void free_all(tetra_demod* h) {
    void* gen[] = { h->hist_far, h->g_xs, h->g_ys, h->d_g_be_a, h->d_g_be_b, h->d_g_rrc };
    for (void* p : gen)
        if (p) (void)hipFree(p);
    void* ptrs[] = { h->agc_g, h->fll_ph, h->fll_fr, h->hist, h->mu, h->omega, h->cph, h->cfr, h->ph2, h->offset,
                     h->prev, h->rrc_valid, h->y, h->ybuf, h->q_ring, h->q_sym, h->q_ptr, h->q_disp, h->q_sync, h->q_err, h->cd_blk, h->cd_part, h->cd_fill, h->cd_blocks, h->d_overruns, h->d_bank, h->d_be_re80, h->d_be_im80,
                     h->d_rrc_ext, h->st_iq, h->st_bits, h->st_nbits, h->st_sym, h->d_prof };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto& slot : h->ev)
        for (auto& e : slot)
            if (e) (void)hipEventDestroy(e);
    auto& a = h->as;
    for (int i = 0; i < 2; i++) {
        void* bufs[] = { a.d_raw[i], a.d_iq[i], a.d_cbits[i], a.d_cnb[i], a.d_out[i], a.d_onb[i] };
        for (void* p : bufs)
            if (p) (void)hipFree(p);
        hipEvent_t evs[] = { a.ev_in[i], a.ev_free[i], a.ev_done[i], a.ev_out[i] };
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
    }
    if (h->pk_dev) (void)hipFree(h->pk_dev);
    if (h->pk_host) (void)hipHostFree(h->pk_host);
    if (h->pk_in) (void)hipHostFree(h->pk_in);
    if (h->tn_out) (void)hipHostFree(h->tn_out);
    if (h->tn_in) (void)hipHostFree(h->tn_in);
    hipStream_t ss[] = { a.s_in, a.s_k, a.s_out, h->own_stream };
    for (hipStream_t st : ss)
        if (st) (void)hipStreamDestroy(st);
}

// Reads the kernels' overrun counter (the device must be idle for this handle's work) and tells whether it moved since the
// last report: > 0 = channels newly cut off, 0 = none, < 0 = a TETRA_ERR_* status.
int new_overruns(tetra_demod* h) {
    int total = 0;
    HIP_TRY(h, hipMemcpy(&total, h->d_overruns, sizeof(int), hipMemcpyDeviceToHost));
    const long long fresh = (long long)total - h->overruns_seen;
    h->overruns_seen = total;
    return fresh > 0 ? (int)(fresh > 0x7fffffff ? 0x7fffffff : fresh) : 0;
}

template <class T> int dalloc(tetra_demod* h, T** p, size_t count) {
    HIP_TRY(h, hipMalloc((void**)p, sizeof(T) * count));
    return TETRA_OK;
}

// Can a launch of this handle take the generic kernel (kernel_generic.hpp)?  Its parameters, or TETRA_FLAG_GENERIC_KERNEL, decide.
bool generic_applies(const tetra_demod* h, const host::Design& d) {
    return host::needs_generic(d) || ((host::needs_long(d) || host::deep_level(d) == 2) && h->force_generic);
}
bool generic_applies(const tetra_demod* h) { return generic_applies(h, h->design); }
// The generic kernel's HBM scratch -- 2 x 8 B x C x (max_samples + 128), the delay lines of a whole call -- is held exactly while
// generic_applies(): allocated by create / the setter that moves the handle there (so that the stream-asynchronous process entry
// point never allocates or synchronises), released by the setter that moves it away.  The device is idle when this runs.
int sync_generic_scratch(tetra_demod* h, const host::Design& d) {
    if (!generic_applies(h, d)) {
        if (h->g_xs) (void)hipFree(h->g_xs);
        if (h->g_ys) (void)hipFree(h->g_ys);
        h->g_xs = h->g_ys = nullptr;
        return TETRA_OK;
    }
    const size_t xs_stride = (size_t)kGenHist + (size_t)h->max_samples, ys_stride = (size_t)kYHist + (size_t)h->max_samples;
    hipError_t e = hipSuccess;
    if (!h->g_xs) e = hipMalloc((void**)&h->g_xs, sizeof(float2) * xs_stride * (size_t)h->C);
    if (e == hipSuccess && !h->g_ys) e = hipMalloc((void**)&h->g_ys, sizeof(float2) * ys_stride * (size_t)h->C);
    if (e != hipSuccess) {
        h->last_hip = (int)e;
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? TETRA_ERR_NOMEM : TETRA_ERR_HIP;
    }
    return TETRA_OK;
}
int sync_generic_scratch(tetra_demod* h) { return sync_generic_scratch(h, h->design); }

}  // namespace

extern "C" {

int tetra_demod_abi_version(void) { return TETRA_DEMOD_ABI_VERSION; }

// sha256 of the sources + flags this library was compiled from (build.py: source_hash() -> -DTETRA_BUILD_ID); the marker in
// front lets build.py read it from the file without loading it.
#ifndef TETRA_BUILD_ID
#define TETRA_BUILD_ID "0000000000000000000000000000000000000000000000000000000000000000"
#endif
const char* tetra_demod_build_id(void) {
    static const char id[] = "TETRA_BUILD_ID=" TETRA_BUILD_ID;
    return id + 15;
}

int tetra_demod_device_info(int device, int* clock_khz, int* compute_units) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return TETRA_ERR_NO_DEVICE;
    int v = 0;
    if (clock_khz) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeClockRate, device) != hipSuccess) return TETRA_ERR_HIP;
        *clock_khz = v;
    }
    if (compute_units) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return TETRA_ERR_HIP;
        *compute_units = v;
    }
    return TETRA_OK;
}

const char* tetra_demod_strerror(int status) {
    switch (status) {
    case TETRA_OK: return "ok";
    case TETRA_ERR_ARG: return "invalid argument";
    case TETRA_ERR_UNSUPPORTED: return "unsupported parameter";
    case TETRA_ERR_NO_DEVICE: return "no usable HIP device";
    case TETRA_ERR_HIP: return "HIP runtime error";
    case TETRA_ERR_NOMEM: return "out of memory";
    case TETRA_ERR_SIZE: return "size out of range";
    case TETRA_ERR_ALIGN: return "misaligned output buffer";
    case TETRA_ERR_OVERRUN: return "a channel filled its output row and was cut off (outputs delivered)";
    default: return "unknown status";
    }
}

int tetra_demod_default_config(tetra_demod_config_t* cfg) {
    if (!cfg) return TETRA_ERR_ARG;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->n_channels = 1;
    cfg->max_samples = 65536;
    cfg->layout = TETRA_LAYOUT_CHANNEL_MAJOR;
    cfg->device = -1;
    cfg->symbolrate = 18000;
    cfg->samplerate = 36000;
    cfg->rrc_tap_count = 65;
    cfg->rrc_beta = 0.35f;
    cfg->agc_rate = 0.02f;
    cfg->costas_bandwidth = 0.01f;
    cfg->fll_bandwidth = 0.006f;
    host::default_timing_gains(cfg->omega_gain, cfg->mu_gain);
    cfg->omega_rel_limit = 0.02f;
    return TETRA_OK;
}

int tetra_demod_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tetra_demod_bits_stride(int n_samples) {
    if (n_samples < 0) return TETRA_ERR_ARG;
    // bits = 2 * symbols; n / 0.95 + 16 covers every timing loop whose slowest step is >= 1.9 samples per symbol
    // (tetra_demod_bits_stride_for: the reference plugin's parameters give 1.9424)
    long long s = (long long)((double)n_samples / 0.95) + 16;
    s = (s + 15) / 16 * 16;
    return s > 0x7ffffff0ll ? TETRA_ERR_SIZE : (int)s;
}

namespace {
long long stride_for(const host::Design& d, long long n) { return host::bits_stride_for(d, n); }
}  // namespace

int tetra_demod_bits_stride_for(tetra_demod_t* h, int n_samples) {
    if (!h || n_samples < 0) return TETRA_ERR_ARG;
    const long long s = stride_for(h->design, n_samples);
    return s > 0x7ffffff0ll ? TETRA_ERR_SIZE : (int)s;
}

int tetra_demod_create(const tetra_demod_config_t* cfg, tetra_demod_t** out) {
    if (!cfg || !out) return TETRA_ERR_ARG;
    *out = nullptr;
    if (cfg->n_channels < 1 || cfg->max_samples < 1) return TETRA_ERR_ARG;
    if (cfg->layout != TETRA_LAYOUT_CHANNEL_MAJOR && cfg->layout != TETRA_LAYOUT_TIME_MAJOR) return TETRA_ERR_ARG;
    {
        const int shapes = cfg->flags & (TETRA_FLAG_WIDE_WORKGROUPS | TETRA_FLAG_NARROW_WORKGROUPS | TETRA_FLAG_SMALL_WORKGROUPS);
        if (shapes & (shapes - 1)) return TETRA_ERR_ARG;      // at most one shape can be forced
    }
    int ndev = tetra_demod_device_count();
    if (ndev <= 0) return TETRA_ERR_NO_DEVICE;
    int dev = cfg->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) return TETRA_ERR_NO_DEVICE;
    }
    if (dev >= ndev) return TETRA_ERR_NO_DEVICE;

    tetra_demod* h = new (std::nothrow) tetra_demod();
    if (!h) return TETRA_ERR_NOMEM;
    h->cfg = *cfg;
    h->cfg.rrc_taps = h->cfg.bandedge_taps = h->cfg.interp_bank = nullptr;
    h->device = dev;
    h->C = cfg->n_channels;
    h->max_samples = cfg->max_samples;
    h->dp.symbolrate = cfg->symbolrate;
    h->dp.samplerate = cfg->samplerate;
    h->dp.rrc_tap_count = cfg->rrc_tap_count;
    h->dp.rrc_beta = cfg->rrc_beta;
    h->dp.agc_rate = cfg->agc_rate;
    h->dp.costas_bandwidth = cfg->costas_bandwidth;
    h->dp.fll_bandwidth = cfg->fll_bandwidth;
    h->dp.omega_gain = cfg->omega_gain;
    h->dp.mu_gain = cfg->mu_gain;
    h->dp.omega_rel_limit = cfg->omega_rel_limit;
    if (!host::make_design(h->dp, cfg->rrc_taps, cfg->bandedge_taps, cfg->interp_bank, h->design) ||
        stride_for(h->design, cfg->max_samples) > 0x7ffffff0ll || (cfg->flags & TETRA_FLAG_RETIRED_TWO_KERNEL)) {
        delete h;
        return TETRA_ERR_UNSUPPORTED;
    }
    DeviceGuard g(dev);
    if (!g.ok) {
        delete h;
        return TETRA_ERR_NO_DEVICE;
    }
    const size_t C = (size_t)h->C;
    int rc = TETRA_OK;
    auto A = [&](int r) { if (rc == TETRA_OK) rc = r; };
    A(dalloc(h, &h->agc_g, C)); A(dalloc(h, &h->fll_ph, C)); A(dalloc(h, &h->fll_fr, C));
    A(dalloc(h, &h->hist, C * kHist));
    A(dalloc(h, &h->hist_far, C * (size_t)(kGenHist - kHist)));
    A(dalloc(h, &h->d_g_be_a, (size_t)kGenMaxTaps)); A(dalloc(h, &h->d_g_be_b, (size_t)kGenMaxTaps)); A(dalloc(h, &h->d_g_rrc, (size_t)kGenMaxTaps));
    A(dalloc(h, &h->mu, C)); A(dalloc(h, &h->omega, C)); A(dalloc(h, &h->cph, C)); A(dalloc(h, &h->cfr, C));
    A(dalloc(h, &h->ph2, C)); A(dalloc(h, &h->offset, C)); A(dalloc(h, &h->prev, C));
    A(dalloc(h, &h->rrc_valid, C));
    h->user_rrc = cfg->rrc_taps != nullptr;
    h->user_be = cfg->bandedge_taps != nullptr;
    h->quirks = (cfg->flags & TETRA_FLAG_REFERENCE_QUIRKS) != 0;
    h->keep_y = (cfg->flags & TETRA_FLAG_KEEP_RRC_OUT) != 0;
    {
        // Workgroup shapes.  16 channels per workgroup is the fastest way through ONE workgroup (kWg16 clocks per sample) and
        // right while there is at most one per CU; the 32-channel workgroup (FLL rows of 4 lanes per channel: the loop code
        // of an FLL wave serves twice the channels; kWg32 clocks per sample) gets a CU through 32 channels in 1.3x that time.
        // Plan: whole rounds of 32-channel workgroups, then the rest in whichever shape is through first (rounds of
        // workgroups per CU x clocks per round) -- at most two launches per call; the flags force one shape for everything.
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cus = cus > 0 ? cus : 256;
        // whole rounds of 32-channel workgroups first; what is left takes whichever shape gets it through in less time: one
        // round of 4-channel workgroups (each has a CU to itself and the shortest FLL step: kWg4 clocks per sample) if there
        // are at most 4 channels per CU, else rounds of 16-channel ones, or one more round of 32-channel ones
        h->cus = cus;
        const long long per_round32 = (long long)kFChWide * cus;
        const long long full = (h->C / per_round32) * per_round32, rest = h->C - full;
        const long long r16 = ((rest + kFCh - 1) / kFCh + cus - 1) / cus, r32 = ((rest + kFChWide - 1) / kFChWide + cus - 1) / cus;
        const long long t16 = r16 * kWg16ClocksPerSample, t32 = r32 * kWg32ClocksPerSample;
        const long long t4 = rest <= (long long)kFChSmall * cus ? (long long)kWg4ClocksPerSample : t16 + t32 + 1;
        const bool rest_wide = rest > 0 && t32 < t16 && t32 < t4;
        h->n_wide = (int)(rest_wide ? h->C : full);
        h->small = rest > 0 && !rest_wide && t4 < t16;
        if (cfg->flags & TETRA_FLAG_WIDE_WORKGROUPS) { h->n_wide = h->C; h->small = false; h->force_shape = true; }
        if (cfg->flags & TETRA_FLAG_NARROW_WORKGROUPS) { h->n_wide = 0; h->small = false; h->force_shape = true; }
        if (cfg->flags & TETRA_FLAG_SMALL_WORKGROUPS) { h->n_wide = 0; h->small = h->force_small = true; }
        h->force_generic = (cfg->flags & TETRA_FLAG_GENERIC_KERNEL) != 0;
    }
    if (h->keep_y) A(dalloc(h, &h->y, C * ((size_t)h->max_samples + kYHist)));
    A(dalloc(h, &h->ybuf, C * kYHist));
    if (cfg->flags & TETRA_FLAG_QUALITY) {
        A(dalloc(h, &h->q_ring, C * 4096)); A(dalloc(h, &h->q_ptr, C));
        A(dalloc(h, &h->q_disp, C)); A(dalloc(h, &h->q_sync, C)); A(dalloc(h, &h->q_err, C));
    }
    if (cfg->flags & TETRA_FLAG_CONSTELLATION) {
        A(dalloc(h, &h->cd_blk, C * (size_t)kCdSyms)); A(dalloc(h, &h->cd_part, C * (size_t)kCdSyms));
        A(dalloc(h, &h->cd_fill, C)); A(dalloc(h, &h->cd_blocks, C));
    }
    if (cfg->flags & (TETRA_FLAG_QUALITY | TETRA_FLAG_CONSTELLATION)) {
        h->q_sym_stride = stride_for(h->design, h->max_samples) / 2;
        A(dalloc(h, &h->q_sym, C * (size_t)h->q_sym_stride));
    }
    A(dalloc(h, &h->d_overruns, (size_t)1));
    A(dalloc(h, &h->d_be_re80, (size_t)kBePadLong)); A(dalloc(h, &h->d_be_im80, (size_t)kBePadLong));   // (sized for the long rows' tables)
    A(dalloc(h, &h->d_rrc_ext, (size_t)kRrcExtLong));
    A(dalloc(h, &h->d_bank, (size_t)kInterpPhases * kInterpTaps));
    if (rc == TETRA_OK && hipMemset(h->d_overruns, 0, sizeof(int)) != hipSuccess) rc = TETRA_ERR_HIP;
    for (auto& slot : h->ev)
        for (auto& e : slot)
            if (rc == TETRA_OK && hipEventCreate(&e) != hipSuccess) rc = TETRA_ERR_HIP;
    if (rc == TETRA_OK) rc = upload_tables(h);
    if (rc == TETRA_OK) rc = reset_range(h, 0, h->C, true);
    if (rc == TETRA_OK) rc = sync_generic_scratch(h);
    if (rc != TETRA_OK) {
        int st = (h->last_hip == (int)hipErrorOutOfMemory) ? TETRA_ERR_NOMEM : rc;
        free_all(h);
        delete h;
        return st;
    }
    *out = h;
    return TETRA_OK;
}

int tetra_demod_destroy(tetra_demod_t* h) {
    if (!h) return TETRA_ERR_ARG;
    DeviceGuard g(h->device);
    (void)hipDeviceSynchronize();
    free_all(h);
    delete h;
    return TETRA_OK;
}

int tetra_demod_process_device(tetra_demod_t* h, const float* d_iq, int n_samples, uint8_t* d_bits, int bits_stride,
                               int32_t* d_n_bits, float* d_sym, void* hip_stream) {
    if (!h || !d_iq || !d_bits || !d_n_bits) return TETRA_ERR_ARG;
    if (n_samples < 0 || n_samples > h->max_samples) return TETRA_ERR_SIZE;
    if (bits_stride < stride_for(h->design, n_samples)) return TETRA_ERR_SIZE;
    if ((bits_stride & 7) || (reinterpret_cast<uintptr_t>(d_bits) & 7) || (reinterpret_cast<uintptr_t>(d_sym) & 7))
        return TETRA_ERR_ALIGN;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)hip_stream;
    h->last_n = n_samples;
    if (n_samples == 0) {
        HIP_TRY(h, hipMemsetAsync(d_n_bits, 0, sizeof(int32_t) * (size_t)h->C, s));
        return TETRA_OK;
    }
    hipEvent_t* ev = h->ev[h->n_calls % tetra_demod::kEvSlots];
    const bool long_rows = host::needs_long(h->design) && !h->force_generic;
    if (generic_applies(h)) {
        // timing loops slower than 0.07 samples per symbol (and, with TETRA_FLAG_GENERIC_KERNEL, those below 0.27 and filters of more than 72 taps): one
        // lane per channel, delay lines in an HBM scratch (kernel_generic.hpp) that create / the setters hold ready (sync_generic_scratch)
        const size_t xs_stride = (size_t)kGenHist + (size_t)h->max_samples, ys_stride = (size_t)kYHist + (size_t)h->max_samples;
        if (!h->g_xs || !h->g_ys) return TETRA_ERR_NOMEM;
        GenericParams pg;
        pg.iq = reinterpret_cast<const float2*>(d_iq);
        if (h->cfg.layout == TETRA_LAYOUT_CHANNEL_MAJOR) { pg.in_ch_stride = n_samples; pg.in_t_stride = 1; }
        else { pg.in_ch_stride = 1; pg.in_t_stride = h->C; }
        pg.n = n_samples; pg.n_channels = h->C;
        pg.agc_g = h->agc_g; pg.fll_ph = h->fll_ph; pg.fll_fr = h->fll_fr; pg.hist = h->hist; pg.hist_far = h->hist_far;
        pg.far_valid = h->far_valid ? 1 : 0;
        pg.rrc_valid = h->rrc_valid; pg.mu = h->mu; pg.omega = h->omega; pg.offset = h->offset;
        pg.cph = h->cph; pg.cfr = h->cfr; pg.ph2 = h->ph2; pg.prev = h->prev; pg.ybuf = h->ybuf;
        pg.be_a = h->d_g_be_a; pg.be_b = h->d_g_be_b; pg.rrc = h->d_g_rrc; pg.ntaps = h->design.ntaps; pg.ntaps_be = h->design.ntaps_be;
        pg.bank = h->d_bank;
        pg.xs = h->g_xs; pg.ys = h->g_ys; pg.xs_stride = (long long)xs_stride; pg.ys_stride = (long long)ys_stride;
        pg.bits = d_bits; pg.bits_stride = bits_stride; pg.n_bits = d_n_bits;
        pg.sym = reinterpret_cast<float2*>(d_sym); pg.sym_stride = bits_stride / 2;
        if (h->taps_sym() && !pg.sym) { pg.sym = h->q_sym; pg.sym_stride = h->q_sym_stride; }
        pg.overruns = h->d_overruns; pg.cut_flag = h->cut_flag;
        pg.y_dbg = h->keep_y ? h->y : nullptr;
        pg.k1 = h->design.k1; pg.k2 = h->design.k2;
        HIP_TRY(h, hipEventRecord(ev[0], s));
        {   // about eight waves per CU when there are enough channels, never more than 64 channels per wave
            int lanes = h->C / (8 * h->cus);
            lanes = lanes < 1 ? 1 : lanes > 64 ? 64 : lanes;
            pg.lanes = lanes;
            hipLaunchKernelGGL(k_generic, dim3((h->C + lanes - 1) / lanes), dim3(64), 0, s, pg);
        }
        if (h->q_ring)
            hipLaunchKernelGGL(k_quality, dim3(h->C), dim3(64), 0, s, pg.sym, pg.sym_stride, d_n_bits, h->q_ring, h->q_ptr, h->q_disp,
                               h->q_err, h->q_sync);
        if (h->cd_blk)
            hipLaunchKernelGGL(k_constellation, dim3(h->C), dim3(256), 0, s, pg.sym, pg.sym_stride, d_n_bits, h->cd_blk, h->cd_part,
                               h->cd_fill, h->cd_blocks);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(ev[1], s));
        h->n_calls++;
        h->far_valid = true;
        return TETRA_OK;
    }
    const bool far_was_valid = h->far_valid;
    {
        FusedParams pf;
        pf.iq = reinterpret_cast<const float2*>(d_iq);
        if (h->cfg.layout == TETRA_LAYOUT_CHANNEL_MAJOR) { pf.in_ch_stride = n_samples; pf.in_t_stride = 1; }
        else { pf.in_ch_stride = 1; pf.in_t_stride = h->C; }
        pf.n = n_samples; pf.n_channels = h->C; pf.ch_base = 0;
        pf.agc_g = h->agc_g; pf.fll_ph = h->fll_ph; pf.fll_fr = h->fll_fr; pf.hist = h->hist;
        pf.rrc_valid = h->rrc_valid;
        pf.mu = h->mu; pf.omega = h->omega; pf.offset = h->offset;
        pf.cph = h->cph; pf.cfr = h->cfr; pf.ph2 = h->ph2; pf.prev = h->prev; pf.ybuf = h->ybuf;
        pf.be_re80 = h->d_be_re80; pf.be_im80 = h->d_be_im80; pf.rrc_ext = h->d_rrc_ext; pf.ntaps = h->design.ntaps;
        pf.bank = h->d_bank;
        pf.bits = d_bits; pf.bits_stride = bits_stride; pf.n_bits = d_n_bits; pf.sym = reinterpret_cast<float2*>(d_sym);
        pf.y_dbg = h->keep_y ? h->y : nullptr;
        pf.overruns = h->d_overruns;
        pf.sym_stride = bits_stride / 2;
        if (h->taps_sym() && !pf.sym) { pf.sym = h->q_sym; pf.sym_stride = h->q_sym_stride; }   // the statistic reads the symbols
        pf.k1 = h->design.k1; pf.k2 = h->design.k2;
        pf.prof = reinterpret_cast<long long*>(h->cut_flag);      // the non-instrumented 16- / 32-channel kernels read it as the cut flag (kernel_fused.hpp)
        // channels [0, n_wide) in 32-channel workgroups (see tetra_demod_create; their FLL rows hold 4 x 17 taps), the rest in
        // 16-channel ones: at most two launches, back to back on the stream
        // (band-edge filters of more than 68 taps do not fit the 32-channel shape's rows: then everything is "the rest")
        // (the same for a timing loop that may emit several symbols from one offset: the deep symbol ring exists for the 16-
        // and 4-channel shapes)
        const bool deep = host::needs_deep(h->design);
        const bool deeper = host::deep_level(h->design) == 2;      // more than 3.7 symbols per sample: the 4-channel shape's 1024-deep ring
        const int n_wide = h->design.ntaps_be <= kF4Pad && !deep && !long_rows ? h->n_wide : 0;
        // (long rows: 4-channel workgroups while every one of them has a CU to itself, 16-channel ones beyond -- or as the flags say)
        const bool rest_small = deeper || h->force_small || (long_rows ? !h->force_shape && h->C <= kFChSmall * h->cus
                                                             : h->small && h->C - n_wide <= kFChSmall * h->cus);
        const dim3 gw((n_wide + kFChWide - 1) / kFChWide), gf((h->C - n_wide + kFCh - 1) / kFCh),
            gs((h->C - n_wide + kFChSmall - 1) / kFChSmall);
        // the FLL's loop filter runs with alpha = 0 (fll.cpp:25; design.hpp never produces anything else): only those kernels exist
        if (pf.k1.fll_alpha != 0.0f) return TETRA_ERR_UNSUPPORTED;
        bool profiling = false;
        (void)profiling;
#ifdef TETRA_DEMOD_DEBUG
        // Debug builds only (profiles/build_debug.sh): TETRA_DEMOD_PROFILE=<file> appends the per-role busy clocks of every
        // launch to <file>.  The release library has neither the getenv nor the instrumented instantiation.
        const char* prof_path = std::getenv("TETRA_DEMOD_PROFILE");
        if (prof_path && n_wide == 0 && !rest_small && !long_rows) {      // (the instrumented instantiation exists for the regular rows only)
            const size_t nwg = (size_t)gf.x;
            if (!h->d_prof) HIP_TRY(h, hipMalloc((void**)&h->d_prof, sizeof(long long) * 8 * nwg));
            HIP_TRY(h, hipMemsetAsync(h->d_prof, 0, sizeof(long long) * 8 * nwg, s));
            pf.prof = h->d_prof;
            profiling = true;
        }
#endif
        HIP_TRY(h, hipEventRecord(ev[0], s));
#ifdef TETRA_DEMOD_DEBUG
        if (profiling && !deep) { FusedParamsT<kFCh> pp; static_cast<FusedParams&>(pp) = pf; hipLaunchKernelGGL((k_fused<true, true>), gf, dim3(kFThreads), 0, s, pp); }
        else
#endif
        {
            if (n_wide > 0) {
                const dim3 tw(fused_threads(kFChWide));
                pf.ch_base = 0;
                { FusedParamsT<kFChWide> pw; static_cast<FusedParams&>(pw) = pf; hipLaunchKernelGGL((k_fused<true, false, kFChWide>), gw, tw, 0, s, pw); }
            }
            if (n_wide < h->C && rest_small) {
                const dim3 ts(fused_threads(kFChSmall));
                pf.ch_base = n_wide;
                FusedParamsT<kFChSmall> ps;
                static_cast<FusedParams&>(ps) = pf;
                ps.cut_flag4 = h->cut_flag;
                if (long_rows) {
                    // filters of 73 .. 129 taps: FLL rows of 16 x 9 taps, 4 channels per workgroup whatever the channel count
                    FusedParamsLongT<kFChSmall> pl;
                    static_cast<FusedParamsT<kFChSmall>&>(pl) = ps;
                    pl.hist_far = h->hist_far;
                    pl.far_valid = far_was_valid ? 1 : 0;
                    if (deeper) hipLaunchKernelGGL((k_fused<true, false, kFChSmall, 2, true>), gs, ts, 0, s, pl);
                    else if (deep) hipLaunchKernelGGL((k_fused<true, false, kFChSmall, 1, true>), gs, ts, 0, s, pl);
                    else hipLaunchKernelGGL((k_fused<true, false, kFChSmall, 0, true>), gs, ts, 0, s, pl);
                } else if (deeper) hipLaunchKernelGGL((k_fused<true, false, kFChSmall, 2>), gs, ts, 0, s, ps);
                else if (deep) hipLaunchKernelGGL((k_fused<true, false, kFChSmall, 1>), gs, ts, 0, s, ps);
                else hipLaunchKernelGGL((k_fused<true, false, kFChSmall>), gs, ts, 0, s, ps);
            } else if (n_wide < h->C) {
                pf.ch_base = n_wide;
                FusedParamsT<kFCh> pn;
                static_cast<FusedParams&>(pn) = pf;
                if (long_rows) {
                    // filters of 73 .. 129 taps on more than 1024 channels: FLL rows of 8 x 17 taps
                    FusedParamsLongT<kFCh> pl;
                    static_cast<FusedParamsT<kFCh>&>(pl) = pn;
                    pl.hist_far = h->hist_far;
                    pl.far_valid = far_was_valid ? 1 : 0;
                    if (deep) hipLaunchKernelGGL((k_fused<true, false, kFCh, 1, true>), gf, dim3(kFThreads), 0, s, pl);
                    else hipLaunchKernelGGL((k_fused<true, false, kFCh, 0, true>), gf, dim3(kFThreads), 0, s, pl);
                } else if (deep) hipLaunchKernelGGL((k_fused<true, false, kFCh, 1>), gf, dim3(kFThreads), 0, s, pn);
                else hipLaunchKernelGGL((k_fused<true>), gf, dim3(kFThreads), 0, s, pn);
            }
        }
        if (h->q_ring)
            hipLaunchKernelGGL(k_quality, dim3(h->C), dim3(64), 0, s, pf.sym, pf.sym_stride, d_n_bits, h->q_ring, h->q_ptr, h->q_disp,
                               h->q_err, h->q_sync);
        if (h->cd_blk)
            hipLaunchKernelGGL(k_constellation, dim3(h->C), dim3(256), 0, s, pf.sym, pf.sym_stride, d_n_bits, h->cd_blk, h->cd_part,
                               h->cd_fill, h->cd_blocks);
        HIP_TRY(h, hipGetLastError());
        h->far_valid = long_rows;      // the fused kernel carries the newest 80 delay-line samples only -- its long rows all 128
        HIP_TRY(h, hipEventRecord(ev[1], s));
        h->n_calls++;
#ifdef TETRA_DEMOD_DEBUG
        if (profiling) {
            const size_t nwg = (size_t)gf.x;
            std::vector<long long> host(8 * nwg);
            HIP_TRY(h, hipStreamSynchronize(s));
            HIP_TRY(h, hipMemcpy(host.data(), h->d_prof, sizeof(long long) * host.size(), hipMemcpyDeviceToHost));
            if (FILE* f = std::fopen(prof_path, "a")) {
                double sum[8] = { 0 };
                for (size_t w = 0; w < nwg; w++)
                    for (int r = 0; r < 8; r++) sum[r] += (double)host[8 * w + r];
                std::fprintf(f, "{\"n\": %d, \"workgroups\": %zu, \"mean_busy_clocks\": {\"E\": %.0f, \"D\": %.0f, \"F0\": %.0f, \"F1\": %.0f, \"A\": %.0f, \"C\": %.0f}, \"mean_total_clocks\": %.0f}\n",
                             n_samples, nwg, sum[0] / nwg, sum[1] / nwg, sum[2] / nwg, sum[3] / nwg, sum[4] / nwg, sum[5] / nwg, sum[7] / nwg);
                std::fclose(f);
            }
        }
#endif
        return TETRA_OK;
    }
}

int tetra_demod_process_resident(tetra_demod_t* h, const float* d_iq, int n_samples, uint8_t* d_bits, int bits_stride,
                                 int32_t* d_n_bits, float* d_sym) {
    if (!h) return TETRA_ERR_ARG;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    if (h->as.ready) {      // like tetra_demod_process: asynchronous calls still in flight finish first (state order)
        HIP_TRY(h, hipStreamSynchronize(h->as.s_in));
        HIP_TRY(h, hipStreamSynchronize(h->as.s_k));
        HIP_TRY(h, hipStreamSynchronize(h->as.s_out));
    }
    if (!h->own_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    const int rc = tetra_demod_process_device(h, d_iq, n_samples, d_bits, bits_stride, d_n_bits, d_sym, h->own_stream);
    if (rc != TETRA_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    const int cut = new_overruns(h);
    return cut > 0 ? TETRA_ERR_OVERRUN : cut;
}

int tetra_demod_process(tetra_demod_t* h, const float* iq, int n_samples, uint8_t* bits, int bits_stride,
                        int32_t* n_bits, float* sym) {
    if (!h || !iq || !bits || !n_bits) return TETRA_ERR_ARG;
    if (n_samples < 0 || n_samples > h->max_samples) return TETRA_ERR_SIZE;
    if (bits_stride < stride_for(h->design, n_samples)) return TETRA_ERR_SIZE;
    if (bits_stride & 7) return TETRA_ERR_ALIGN;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    if (h->as.ready) {      // asynchronous calls still in flight run on their own streams: let them finish first (state order)
        HIP_TRY(h, hipStreamSynchronize(h->as.s_in));
        HIP_TRY(h, hipStreamSynchronize(h->as.s_k));
        HIP_TRY(h, hipStreamSynchronize(h->as.s_out));
    }
    const size_t C = (size_t)h->C;
    const size_t iq_bytes = sizeof(float) * 2 * C * (size_t)n_samples;
    const size_t bits_bytes = C * (size_t)bits_stride;
    const size_t sym_bytes = sym ? sizeof(float) * 2 * C * (size_t)(bits_stride / 2) : 0;
    // Small calls -- the single-channel drop-in hands over 180 samples at a time (SDR++'s stream chunks at 36 ksps) -- are
    // dominated by the four blocking copies around a ~40 us launch.  They take one asynchronous chain on the handle's own
    // stream instead: samples through a page-locked bounce buffer, ONE packed output [n_bits | bits | symbols] back into
    // page-locked memory, one synchronisation, then plain memcpys into the caller's arrays.
    constexpr size_t kSmallCall = 256 * 1024;
    const size_t nb_bytes = (sizeof(int) * C + 15) / 16 * 16;
    const size_t pack_bytes = nb_bytes + bits_bytes + sym_bytes + 16;      // + the overrun counter, so that it rides along
#ifdef TETRA_EXP_TINY_ENV      // experiment builds: the in-place limit from the environment (profiles/measure_tiny_calls.py)
    static const int kTiny = std::getenv("TETRA_TINY_SAMPLES") ? std::atoi(std::getenv("TETRA_TINY_SAMPLES")) : kTinyCallSamples;
#else
    constexpr int kTiny = kTinyCallSamples;
#endif
    if (n_samples > 0 && n_samples <= kTiny && iq_bytes <= kSmallCall && pack_bytes <= kSmallCall && !h->tn_disabled) {
        // The shortest calls use no copy engine at all: the CPU copies the samples into a page-locked, mapped, coherent block
        // that the AGC wave reads in place over PCIe (a tile ahead, as always), the kernels write n_bits | bits | symbols | a
        // "some channel was cut off" flag straight into a second such block, ONE synchronisation, plain memcpys out.  Measured
        // (profiles/r03/r03_ad_tiny_calls.json): 1 x 180 samples 67.7 -> 59.9 us per call, 16 x 180 77.6 -> 62.7, 64 x 180
        // 88.0 -> 65.9, 64 x 500 128.7 -> 109.4; the launch itself gets ~10 % slower per sample (the AGC wave's loads cross
        // PCIe), which is why calls of more than kTinyCallSamples keep the copy engines (1 x 1024: 144 vs 147 us).
        // The overrun COUNTER stays in device memory (an atomic across PCIe is not something every platform routes); a channel
        // that is cut off additionally leaves a plain store in the host block, and only then is the counter read back.
        // A platform that refuses mapped + coherent host memory (or its device pointer) loses nothing but this shortcut: the
        // blocks are only kept once BOTH calls succeeded, otherwise the path is switched off and the call continues below.
        if (!h->own_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        auto mapped_block = [&](uint8_t** host_p, uint8_t** dev_p) {
            if (*host_p) return true;
            uint8_t *hp = nullptr, *dp = nullptr;
            if (hipHostMalloc((void**)&hp, kSmallCall, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (hipHostGetDevicePointer((void**)&dp, hp, 0) != hipSuccess || !dp) { (void)hipGetLastError(); (void)hipHostFree(hp); return false; }
            std::memset(hp, 0, kSmallCall);
            *host_p = hp; *dev_p = dp;
            return true;
        };
        if (!mapped_block(&h->tn_out, &h->tn_out_dev) || !mapped_block(&h->tn_in, &h->tn_in_dev)) h->tn_disabled = true;
    }
    if (n_samples > 0 && n_samples <= kTiny && iq_bytes <= kSmallCall && pack_bytes <= kSmallCall && !h->tn_disabled) {
        std::memcpy(h->tn_in, iq, iq_bytes);
        volatile int* flag = reinterpret_cast<volatile int*>(h->tn_out + pack_bytes - 16);
        *flag = 0;
        uint8_t* d_bits = h->tn_out_dev + nb_bytes;
        h->cut_flag = reinterpret_cast<int*>(h->tn_out_dev + pack_bytes - 16);
        const int rc = tetra_demod_process_device(h, reinterpret_cast<const float*>(h->tn_in_dev), n_samples, d_bits, bits_stride,
                                                  reinterpret_cast<int32_t*>(h->tn_out_dev),
                                                  sym ? reinterpret_cast<float*>(d_bits + bits_bytes) : nullptr, h->own_stream);
        h->cut_flag = nullptr;
        if (rc != TETRA_OK) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->own_stream));
        std::memcpy(n_bits, h->tn_out, sizeof(int) * C);
        std::memcpy(bits, h->tn_out + nb_bytes, bits_bytes);
        if (sym) std::memcpy(sym, h->tn_out + nb_bytes + bits_bytes, sym_bytes);
        if (*flag == 0) return TETRA_OK;
        const int cut = new_overruns(h);          // rare: a poisoned channel filled its row
        return cut < 0 ? cut : TETRA_ERR_OVERRUN;
    }
    if (n_samples > 0 && iq_bytes <= kSmallCall && pack_bytes <= kSmallCall) {
        if (!h->own_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        if (pack_bytes > h->pk_bytes) {
            if (h->pk_dev) (void)hipFree(h->pk_dev);
            if (h->pk_host) (void)hipHostFree(h->pk_host);
            h->pk_dev = h->pk_host = nullptr; h->pk_bytes = 0;
            HIP_TRY(h, hipMalloc((void**)&h->pk_dev, kSmallCall));
            HIP_TRY(h, hipMemset(h->pk_dev, 0, kSmallCall));
            HIP_TRY(h, hipHostMalloc((void**)&h->pk_host, kSmallCall, hipHostMallocDefault));
            h->pk_bytes = kSmallCall;
        }
        if (iq_bytes > h->pk_in_bytes) {
            if (h->pk_in) (void)hipHostFree(h->pk_in);
            h->pk_in = nullptr; h->pk_in_bytes = 0;
            HIP_TRY(h, hipHostMalloc((void**)&h->pk_in, kSmallCall, hipHostMallocDefault));
            h->pk_in_bytes = kSmallCall;
        }
        if (iq_bytes > h->st_iq_bytes) {
            if (h->st_iq) (void)hipFree(h->st_iq);
            h->st_iq = nullptr; h->st_iq_bytes = 0;
            HIP_TRY(h, hipMalloc((void**)&h->st_iq, iq_bytes));
            h->st_iq_bytes = iq_bytes;
        }
        std::memcpy(h->pk_in, iq, iq_bytes);
        HIP_TRY(h, hipMemcpyAsync(h->st_iq, h->pk_in, iq_bytes, hipMemcpyHostToDevice, h->own_stream));
        uint8_t* d_nb = h->pk_dev;
        uint8_t* d_bits = h->pk_dev + nb_bytes;
        uint8_t* d_sym = d_bits + bits_bytes;
        int rc = tetra_demod_process_device(h, h->st_iq, n_samples, d_bits, bits_stride, reinterpret_cast<int32_t*>(d_nb),
                                            sym ? reinterpret_cast<float*>(d_sym) : nullptr, h->own_stream);
        if (rc != TETRA_OK) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->pk_dev + pack_bytes - 16, h->d_overruns, sizeof(int), hipMemcpyDeviceToDevice, h->own_stream));
        HIP_TRY(h, hipMemcpyAsync(h->pk_host, h->pk_dev, pack_bytes, hipMemcpyDeviceToHost, h->own_stream));
        HIP_TRY(h, hipStreamSynchronize(h->own_stream));
        std::memcpy(n_bits, h->pk_host, sizeof(int) * C);
        std::memcpy(bits, h->pk_host + nb_bytes, bits_bytes);
        if (sym) std::memcpy(sym, h->pk_host + nb_bytes + bits_bytes, sym_bytes);
        int total = 0;
        std::memcpy(&total, h->pk_host + pack_bytes - 16, sizeof(int));
        const long long fresh = (long long)total - h->overruns_seen;
        h->overruns_seen = total;
        return fresh > 0 ? TETRA_ERR_OVERRUN : TETRA_OK;
    }
    if (iq_bytes > h->st_iq_bytes) {
        if (h->st_iq) (void)hipFree(h->st_iq);
        h->st_iq = nullptr; h->st_iq_bytes = 0;
        HIP_TRY(h, hipMalloc((void**)&h->st_iq, iq_bytes));
        h->st_iq_bytes = iq_bytes;
    }
    if (bits_bytes > h->st_bits_bytes) {
        if (h->st_bits) (void)hipFree(h->st_bits);
        h->st_bits = nullptr; h->st_bits_bytes = 0;
        HIP_TRY(h, hipMalloc((void**)&h->st_bits, bits_bytes));
        HIP_TRY(h, hipMemset(h->st_bits, 0, bits_bytes));      // once: the kernels define bits[c][0 .. n_bits[c]) per call, the rest stays as it is
        h->st_bits_bytes = bits_bytes;
    }
    if (!h->st_nbits) HIP_TRY(h, hipMalloc((void**)&h->st_nbits, sizeof(int) * C));
    if (sym_bytes > h->st_sym_bytes) {
        if (h->st_sym) (void)hipFree(h->st_sym);
        h->st_sym = nullptr; h->st_sym_bytes = 0;
        HIP_TRY(h, hipMalloc((void**)&h->st_sym, sym_bytes));
        h->st_sym_bytes = sym_bytes;
    }
    if (iq_bytes) HIP_TRY(h, hipMemcpy(h->st_iq, iq, iq_bytes, hipMemcpyHostToDevice));
#ifdef TETRA_DEMOD_DEBUG
    HIP_TRY(h, hipMemset(h->st_bits, 0, bits_bytes));      // release builds: only bits[c][0 .. n_bits[c]) are defined
#endif
    int rc = tetra_demod_process_device(h, h->st_iq ? h->st_iq : reinterpret_cast<const float*>(h->agc_g), n_samples,
                                        h->st_bits, bits_stride, h->st_nbits, sym ? h->st_sym : nullptr, nullptr);
    if (rc != TETRA_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(0));
    HIP_TRY(h, hipMemcpy(bits, h->st_bits, bits_bytes, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(n_bits, h->st_nbits, sizeof(int) * C, hipMemcpyDeviceToHost));
    if (sym) HIP_TRY(h, hipMemcpy(sym, h->st_sym, sym_bytes, hipMemcpyDeviceToHost));
    const int cut = new_overruns(h);
    return cut > 0 ? TETRA_ERR_OVERRUN : cut;
}

// ------------------------------------------------------------------------------------------------
// Asynchronous host entry point.  The call is cut along the TIME axis into chunks (state carries from chunk to chunk like
// from call to call, so the bits are those of one call); chunk k+1 crosses PCIe while chunk k is demodulated:
//   s_in : H2D of chunk k into slot k%2                      (waits until the kernel that last read that slot is done)
//   s_k  : [int16 -> float] -> k_fused -> k_append_bits        (waits for the copy)
//   s_out: D2H of the call's bits and counts                   (waits for the last append)
// Two calls may be in flight (two output slots), so call j+1's input copy overlaps call j's output copy.
// ------------------------------------------------------------------------------------------------
namespace {
int grow(tetra_demod* h, void** p, size_t* have, size_t want) {
    if (want <= *have && *p) return TETRA_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    HIP_TRY(h, hipMalloc(p, want));
    return TETRA_OK;
}

int async_chunk_len(int n_samples) {
    int k = n_samples / 4096;
    k = k < 1 ? 1 : (k > 8 ? 8 : k);
    const int len = (n_samples + k - 1) / k;
    return (len + 31) & ~31;
}
}  // namespace

namespace {
// The body of tetra_demod_process_async after its resources exist; any failure leaves work enqueued on the three streams.
int async_enqueue(tetra_demod* h, const void* iq, int iq_format, int n_samples, uint8_t* bits, int bits_stride, int32_t* n_bits) {
    auto& a = h->as;
    const size_t C = (size_t)h->C;
    const int chunk = async_chunk_len(n_samples);
    const int cstride = (int)stride_for(h->design, chunk);
    const size_t in_elem = iq_format == TETRA_IQ_CS16 ? sizeof(short) * 2 : iq_format == TETRA_IQ_CS8 ? 2 : sizeof(float) * 2;
    const size_t want_iq = sizeof(float) * 2 * C * (size_t)chunk, want_raw = iq_format != TETRA_IQ_CF32 ? in_elem * C * (size_t)chunk : 0;
    const size_t want_cbits = C * (size_t)cstride, want_out = C * (size_t)bits_stride;
    if (want_iq > a.iq_bytes || want_raw > a.raw_bytes || want_cbits > a.cbits_bytes || want_out > a.out_bytes) {
        HIP_TRY(h, hipDeviceSynchronize());          // buffers may be in use by calls still in flight
        for (int i = 0; i < 2; i++) {
            int rc = TETRA_OK;
            size_t t;
            t = a.iq_bytes; rc = grow(h, (void**)&a.d_iq[i], &t, want_iq);
            if (rc == TETRA_OK && want_raw) { t = a.raw_bytes; rc = grow(h, &a.d_raw[i], &t, want_raw); }
            if (rc == TETRA_OK) { t = a.cbits_bytes; rc = grow(h, (void**)&a.d_cbits[i], &t, want_cbits); }
            if (rc == TETRA_OK) { t = a.out_bytes; rc = grow(h, (void**)&a.d_out[i], &t, want_out); }
            if (rc == TETRA_OK && want_out > a.out_bytes) {      // once; a call defines bits[c][0 .. n_bits[c])
                const hipError_t e = hipMemset(a.d_out[i], 0, want_out);
                if (e != hipSuccess) { h->last_hip = (int)e; rc = TETRA_ERR_HIP; }
            }
            if (rc != TETRA_OK) {      // a buffer may be gone: forget every size, the next call allocates all of them again
                a.iq_bytes = a.raw_bytes = a.cbits_bytes = a.out_bytes = 0;
                return rc;
            }
        }
        a.iq_bytes = a.iq_bytes > want_iq ? a.iq_bytes : want_iq;
        if (want_raw) a.raw_bytes = a.raw_bytes > want_raw ? a.raw_bytes : want_raw;
        a.cbits_bytes = a.cbits_bytes > want_cbits ? a.cbits_bytes : want_cbits;
        a.out_bytes = a.out_bytes > want_out ? a.out_bytes : want_out;
    }
    const int os = (int)(a.calls & 1);               // output slot of this call
    if (a.calls >= 2) HIP_TRY(h, hipStreamWaitEvent(a.s_k, a.ev_out[os], 0));     // its previous user's D2H is done
    HIP_TRY(h, hipMemsetAsync(a.d_onb[os], 0, sizeof(int) * C, a.s_k));
    const bool time_major = h->cfg.layout == TETRA_LAYOUT_TIME_MAJOR;
    const uint8_t* src = static_cast<const uint8_t*>(iq);
    for (int pos = 0; pos < n_samples; pos += chunk) {
        const int len = n_samples - pos < chunk ? n_samples - pos : chunk;
        const int sl = (int)(a.chunks & 1);
        if (a.chunks >= 2) HIP_TRY(h, hipStreamWaitEvent(a.s_in, a.ev_free[sl], 0));
        void* dst = iq_format != TETRA_IQ_CF32 ? a.d_raw[sl] : (void*)a.d_iq[sl];
        if (time_major) {      // iq[n][c]: a time chunk is contiguous
            HIP_TRY(h, hipMemcpyAsync(dst, src + in_elem * C * (size_t)pos, in_elem * C * (size_t)len, hipMemcpyHostToDevice, a.s_in));
        } else {               // iq[c][n]: one row piece per channel, packed to [C][len] on the device
            HIP_TRY(h, hipMemcpy2DAsync(dst, in_elem * (size_t)len, src + in_elem * (size_t)pos, in_elem * (size_t)n_samples,
                                        in_elem * (size_t)len, C, hipMemcpyHostToDevice, a.s_in));
        }
        HIP_TRY(h, hipEventRecord(a.ev_in[sl], a.s_in));
        HIP_TRY(h, hipStreamWaitEvent(a.s_k, a.ev_in[sl], 0));
        if (iq_format == TETRA_IQ_CS16) {
            const long long n = (long long)C * len;
            hipLaunchKernelGGL(k_cs16_to_cf32, dim3(2048), dim3(256), 0, a.s_k, static_cast<const short2*>(a.d_raw[sl]),
                               reinterpret_cast<float2*>(a.d_iq[sl]), n);
            HIP_TRY(h, hipGetLastError());
        } else if (iq_format == TETRA_IQ_CS8) {
            const long long n = (long long)C * len;
            hipLaunchKernelGGL(k_cs8_to_cf32, dim3(2048), dim3(256), 0, a.s_k, static_cast<const char2*>(a.d_raw[sl]),
                               reinterpret_cast<float2*>(a.d_iq[sl]), n);
            HIP_TRY(h, hipGetLastError());
        }
        const int rc = tetra_demod_process_device(h, a.d_iq[sl], len, a.d_cbits[sl], cstride, a.d_cnb[sl], nullptr, a.s_k);
        if (rc != TETRA_OK) return rc;
        hipLaunchKernelGGL(k_append_bits, dim3((unsigned)C), dim3(64), 0, a.s_k, a.d_cbits[sl], cstride, a.d_cnb[sl], a.d_out[os],
                           bits_stride, a.d_onb[os]);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(a.ev_free[sl], a.s_k));
        a.chunks++;
    }
    HIP_TRY(h, hipEventRecord(a.ev_done[os], a.s_k));
    HIP_TRY(h, hipStreamWaitEvent(a.s_out, a.ev_done[os], 0));
    HIP_TRY(h, hipMemcpyAsync(bits, a.d_out[os], want_out, hipMemcpyDeviceToHost, a.s_out));
    HIP_TRY(h, hipMemcpyAsync(n_bits, a.d_onb[os], sizeof(int) * C, hipMemcpyDeviceToHost, a.s_out));
    HIP_TRY(h, hipEventRecord(a.ev_out[os], a.s_out));
    a.calls++;
    return TETRA_OK;
}
}  // namespace

int tetra_demod_process_async(tetra_demod_t* h, const void* iq, int iq_format, int n_samples, uint8_t* bits, int bits_stride,
                              int32_t* n_bits) {
    if (!h || !iq || !bits || !n_bits) return TETRA_ERR_ARG;
    if (iq_format != TETRA_IQ_CF32 && iq_format != TETRA_IQ_CS16 && iq_format != TETRA_IQ_CS8) return TETRA_ERR_ARG;
    if (n_samples < 0 || n_samples > h->max_samples) return TETRA_ERR_SIZE;
    if (bits_stride < stride_for(h->design, n_samples)) return TETRA_ERR_SIZE;
    if (bits_stride & 7) return TETRA_ERR_ALIGN;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    auto& a = h->as;
    const size_t C = (size_t)h->C;
    if (n_samples == 0) {      // nothing to enqueue: the counts are final right away
        for (size_t c = 0; c < C; c++) n_bits[c] = 0;
        return TETRA_OK;
    }
    if (!a.ready) {
        // every resource is created only where it is still missing, so a call after a failed set-up neither leaks nor
        // re-creates what already exists
        if (!a.s_in) HIP_TRY(h, hipStreamCreateWithFlags(&a.s_in, hipStreamNonBlocking));
        if (!a.s_k) HIP_TRY(h, hipStreamCreateWithFlags(&a.s_k, hipStreamNonBlocking));
        if (!a.s_out) HIP_TRY(h, hipStreamCreateWithFlags(&a.s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            if (!a.ev_in[i]) HIP_TRY(h, hipEventCreateWithFlags(&a.ev_in[i], hipEventDisableTiming));
            if (!a.ev_free[i]) HIP_TRY(h, hipEventCreateWithFlags(&a.ev_free[i], hipEventDisableTiming));
            if (!a.ev_done[i]) HIP_TRY(h, hipEventCreateWithFlags(&a.ev_done[i], hipEventDisableTiming));
            if (!a.ev_out[i]) HIP_TRY(h, hipEventCreateWithFlags(&a.ev_out[i], hipEventDisableTiming));
            if (!a.d_cnb[i]) HIP_TRY(h, hipMalloc((void**)&a.d_cnb[i], sizeof(int) * C));
            if (!a.d_onb[i]) HIP_TRY(h, hipMalloc((void**)&a.d_onb[i], sizeof(int) * C));
        }
        a.ready = true;
    }
    const int rc = async_enqueue(h, iq, iq_format, n_samples, bits, bits_stride, n_bits);
    if (rc != TETRA_OK) {
        // part of the call may be enqueued: let it drain, then start the slot / event bookkeeping afresh (no wait of a later
        // call refers to an event this call did not get to record)
        (void)hipStreamSynchronize(a.s_in);
        (void)hipStreamSynchronize(a.s_k);
        (void)hipStreamSynchronize(a.s_out);
        a.chunks = 0;
        a.calls = 0;
    }
    return rc;
}

int tetra_demod_wait(tetra_demod_t* h) {
    if (!h) return TETRA_ERR_ARG;
    if (!h->as.ready) return TETRA_OK;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    HIP_TRY(h, hipStreamSynchronize(h->as.s_in));
    HIP_TRY(h, hipStreamSynchronize(h->as.s_k));
    HIP_TRY(h, hipStreamSynchronize(h->as.s_out));
    const int cut = new_overruns(h);
    return cut > 0 ? TETRA_ERR_OVERRUN : cut;
}

int tetra_demod_get_overruns(tetra_demod_t* h, long long* total) {
    if (!h || !total) return TETRA_ERR_ARG;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    HIP_TRY(h, hipDeviceSynchronize());
    int v = 0;
    HIP_TRY(h, hipMemcpy(&v, h->d_overruns, sizeof(int), hipMemcpyDeviceToHost));
    *total = (long long)v;
    return TETRA_OK;
}

void* tetra_demod_host_alloc(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void tetra_demod_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int tetra_demod_reset(tetra_demod_t* h, int channel) {
    if (!h) return TETRA_ERR_ARG;
    if (channel < -1 || channel >= h->C) return TETRA_ERR_ARG;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    HIP_TRY(h, hipDeviceSynchronize());
    return channel < 0 ? reset_range(h, 0, h->C, !h->quirks) : reset_range(h, channel, 1, !h->quirks);
}

namespace {
// What every setter ends in: validate the new parameter set, re-design what `tables` / `with_tap_count` ask for, then commit
// (device tables, the RRC's view of a longer delay line, the timing loop, the statistic's symbol scratch).
int apply_params(tetra_demod* h, const host::DesignParams& np, bool tables, bool new_tap_count, bool timing_reset) {
    host::Design nd = h->design;          // caller-supplied tables and everything a setter does not own are carried over
    if (!host::params_ok(np)) return TETRA_ERR_UNSUPPORTED;
    if (tables) {
        // A caller-supplied RRC table is the caller's: the rate setters leave it alone and do the rest of their work (timing loop:
        // COMPLEX_FD::setOmega); the caller follows with tetra_demod_set_tables (what the SDR++ build of the C++ mirror does with
        // taps::rootRaisedCosine's output, pi4dqpsk.cpp:36-40).  A setter that ONLY re-designs the RRC has nothing left to do.
        if (h->user_rrc && !timing_reset) return TETRA_ERR_UNSUPPORTED;
        if (!h->user_rrc) host::design_rrc(np, nd);
        if (new_tap_count && !h->quirks && np.rrc_tap_count != nd.ntaps_be) {
            if (h->user_be) return TETRA_ERR_UNSUPPORTED;
            host::design_bandedge(np, nd, np.rrc_tap_count);           // documented deviation: one length for the three FIRs
        }
        if (nd.ntaps > kGenMaxTaps || nd.ntaps_be > kGenMaxTaps) return TETRA_ERR_UNSUPPORTED;   // (params_ok has said so already)
    }
    host::design_loops(np, nd);
    host::design_timing_limits(np, nd);
    if (stride_for(nd, h->max_samples) > 0x7ffffff0ll) return TETRA_ERR_UNSUPPORTED;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    HIP_TRY(h, hipDeviceSynchronize());
    if (h->taps_sym() && stride_for(nd, h->max_samples) / 2 > h->q_sym_stride) {
        // a slower timing loop emits more symbols per sample: the statistic's symbol scratch grows with it
        const long long want = stride_for(nd, h->max_samples) / 2;
        float2* q = nullptr;
        HIP_TRY(h, hipMalloc((void**)&q, sizeof(float2) * (size_t)h->C * (size_t)want));
        (void)hipFree(h->q_sym);
        h->q_sym = q;
        h->q_sym_stride = want;
    }
    if (generic_applies(h, nd)) {      // the generic kernel's scratch first: a failure leaves the handle as it was
        const int rc = sync_generic_scratch(h, nd);
        if (rc != TETRA_OK) return rc;
    }
    const int old_ntaps = h->design.ntaps;
    h->dp = np;
    h->design = nd;
    (void)sync_generic_scratch(h);        // (releases it when the parameters have left the generic kernel's domain)
    if (tables) {
        int rc = upload_tables(h);
        if (rc != TETRA_OK) return rc;
        if (h->quirks && nd.ntaps > old_ntaps) {
            // FIR::setTaps with more taps keeps the RRC's old taps-1 history samples and zero-fills the newly visible part
            hipLaunchKernelGGL(k_min_i32, dim3((h->C + 255) / 256), dim3(256), 0, 0, h->rrc_valid, old_ntaps - 1, h->C);
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipStreamSynchronize(0));
        }
    }
    if (timing_reset) {
        int rc = reset_timing(h, 0, h->C);
        if (rc != TETRA_OK) return rc;
        HIP_TRY(h, hipStreamSynchronize(0));
    }
    return TETRA_OK;
}
}  // namespace

int tetra_demod_set_param(tetra_demod_t* h, int param_id, double value) {
    if (!h) return TETRA_ERR_ARG;
    host::DesignParams np = h->dp;
    bool timing_reset = false, tables = false;
    switch (param_id) {
    // loop setters (pi4dqpsk.cpp:76-118): loop constants only
    case TETRA_PARAM_AGC_RATE: np.agc_rate = value; break;
    case TETRA_PARAM_COSTAS_BANDWIDTH: np.costas_bandwidth = value; break;
    case TETRA_PARAM_FLL_BANDWIDTH: np.fll_bandwidth = value; break;
    case TETRA_PARAM_OMEGA_GAIN: np.omega_gain = value; break;
    case TETRA_PARAM_MU_GAIN: np.mu_gain = value; break;
    case TETRA_PARAM_OMEGA_REL_LIMIT: np.omega_rel_limit = value; break;
    // rate setters (pi4dqpsk.cpp:32-54): RRC taps + COMPLEX_FD::setOmega; the FLL's filters are not touched
    case TETRA_PARAM_SYMBOLRATE:
    case TETRA_PARAM_SAMPLERATE:
        if (param_id == TETRA_PARAM_SYMBOLRATE) np.symbolrate = value; else np.samplerate = value;
        timing_reset = tables = true;
        break;
    // setRRCTapCount / the beta half of setRRCParams (pi4dqpsk.cpp:56-74)
    case TETRA_PARAM_RRC_TAP_COUNT: np.rrc_tap_count = (int)value; tables = true; break;
    case TETRA_PARAM_RRC_BETA: np.rrc_beta = value; tables = true; break;
    default: return TETRA_ERR_ARG;
    }
    return apply_params(h, np, tables, param_id == TETRA_PARAM_RRC_TAP_COUNT, timing_reset);
}

// PI4DQPSK::setRRCParams (pi4dqpsk.cpp:56-66): tap count and roll-off in ONE re-design of the RRC.
int tetra_demod_set_rrc_params(tetra_demod_t* h, int rrc_tap_count, double rrc_beta) {
    if (!h) return TETRA_ERR_ARG;
    host::DesignParams np = h->dp;
    np.rrc_tap_count = rrc_tap_count;
    np.rrc_beta = rrc_beta;
    return apply_params(h, np, true, true, false);
}

// FIR::setTaps with tables the CALLER designed (ABI 5): what PI4DQPSK::init / setSymbolrate / setSamplerate / setRRCParams do with
// the output of SDR++'s own generators (pi4dqpsk.cpp:18-19,38-39,50-51,63-64), FLL::createBandedgeFilters (fll.cpp:61-95) and
// COMPLEX_FD::generateInterpTaps (complex_fd.cpp:153-158).
int tetra_demod_set_tables(tetra_demod_t* h, const float* rrc_taps, int n_rrc, const float* bandedge_taps, int n_be,
                           const float* interp_bank) {
    if (!h) return TETRA_ERR_ARG;
    if (!rrc_taps && !bandedge_taps && !interp_bank) return TETRA_ERR_ARG;
    if (rrc_taps && (n_rrc < 2 || n_rrc > kGenMaxTaps)) return TETRA_ERR_UNSUPPORTED;
    if (bandedge_taps && (n_be < 2 || n_be > kGenMaxTaps)) return TETRA_ERR_UNSUPPORTED;
    host::Design nd = h->design;
    host::DesignParams np = h->dp;
    if (rrc_taps) {
        nd.ntaps = n_rrc;
        nd.rrc.assign(rrc_taps, rrc_taps + n_rrc);
        np.rrc_tap_count = n_rrc;
    }
    if (bandedge_taps) {
        nd.ntaps_be = n_be;
        nd.be_re.assign(bandedge_taps, bandedge_taps + n_be);
        nd.be_im.assign(bandedge_taps + n_be, bandedge_taps + 2 * (size_t)n_be);
    }
    if (interp_bank) nd.bank.assign(interp_bank, interp_bank + kInterpPhases * kInterpTaps);
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    HIP_TRY(h, hipDeviceSynchronize());
    if (generic_applies(h, nd)) {
        const int rc = sync_generic_scratch(h, nd);
        if (rc != TETRA_OK) return rc;
    }
    const int old_ntaps = h->design.ntaps;
    // commit, upload -- and take the commit back if the upload fails: the handle never stays half-updated (its host-side design is
    // what the next setter re-designs from and what get_tables reports)
    const host::Design old_design = h->design;
    const host::DesignParams old_dp = h->dp;
    const bool old_user_rrc = h->user_rrc, old_user_be = h->user_be;
    h->dp = np;
    h->design = nd;
    if (rrc_taps) h->user_rrc = true;
    if (bandedge_taps) h->user_be = true;
    (void)sync_generic_scratch(h);
    int rc = upload_tables(h);
    if (rc != TETRA_OK) {
        h->dp = old_dp;
        h->design = old_design;
        h->user_rrc = old_user_rrc;
        h->user_be = old_user_be;
        (void)sync_generic_scratch(h);
        (void)upload_tables(h);
        return rc;
    }
    if (h->quirks && nd.ntaps > old_ntaps) {
        // FIR::setTaps with more taps keeps the RRC's old taps-1 history samples and zero-fills the newly visible part
        hipLaunchKernelGGL(k_min_i32, dim3((h->C + 255) / 256), dim3(256), 0, 0, h->rrc_valid, old_ntaps - 1, h->C);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize(0));
    }
    return TETRA_OK;
}

int tetra_demod_get_state(tetra_demod_t* h, int channel, tetra_demod_channel_state_t* out) {
    if (!h || !out || channel < 0 || channel >= h->C) return TETRA_ERR_ARG;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    HIP_TRY(h, hipDeviceSynchronize());
    const int c = channel;
#define GET1(dst, src) HIP_TRY(h, hipMemcpy(&(dst), (src) + c, sizeof(dst), hipMemcpyDeviceToHost))
    GET1(out->agc_gain, h->agc_g); GET1(out->fll_phase, h->fll_ph); GET1(out->fll_freq, h->fll_fr);
    GET1(out->mu, h->mu); GET1(out->omega, h->omega); GET1(out->offset, h->offset);
    GET1(out->costas_phase, h->cph); GET1(out->costas_freq, h->cfr); GET1(out->ph2, h->ph2); GET1(out->prev, h->prev);
    GET1(out->rrc_valid, h->rrc_valid);
#undef GET1
    HIP_TRY(h, hipMemcpy(out->hist, h->hist + (size_t)c * kHist, sizeof(float2) * kHist, hipMemcpyDeviceToHost));
    if (h->far_valid)
        HIP_TRY(h, hipMemcpy(out->hist_far, h->hist_far + (size_t)c * (kGenHist - kHist), sizeof(float2) * (kGenHist - kHist), hipMemcpyDeviceToHost));
    else
        std::memset(out->hist_far, 0, sizeof(out->hist_far));
    HIP_TRY(h, hipMemcpy(out->ybuf, h->ybuf + (size_t)c * kYHist, sizeof(float2) * kYHist, hipMemcpyDeviceToHost));
    return TETRA_OK;
}

int tetra_demod_set_state(tetra_demod_t* h, int channel, const tetra_demod_channel_state_t* in) {
    if (!h || !in || channel < 0 || channel >= h->C) return TETRA_ERR_ARG;
    // Phases outside what the loops can produce are refused (every pcl.advance wraps to [-pi, pi], ph2 to (-2 pi, 2 pi),
    // pi4dqpsk_costas.cpp:10-15): the kernel's phasor evaluation relies on those ranges.  NaN (a poisoned channel) passes.
    if (std::fabs(in->fll_phase) > kFlPi || std::fabs(in->costas_phase) > kFlPi || std::fabs(in->ph2) >= 2 * kFlPi) return TETRA_ERR_ARG;
    // The carried read position of the timing loop is >= 0 by construction (complex_fd.cpp:141-148: offset -= count only after
    // the loop left with offset >= count) and mu is a fraction in [0, 1) -- or NaN on a poisoned channel; a negative offset or an
    // infinite mu would index the kernels' delay lines out of range.
    if (in->offset < 0 || std::isinf(in->mu)) return TETRA_ERR_ARG;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    HIP_TRY(h, hipDeviceSynchronize());
    const int c = channel;
#define SET1(dst, src) HIP_TRY(h, hipMemcpy((dst) + c, &(src), sizeof(src), hipMemcpyHostToDevice))
    // a phase of -0 is stored as +0: the loops never produce one (a sum is -0 only for two -0 operands, and the chain starts at
    // +0), and the FLL blocks' rint-based phase wrap (gen_fll_asm.py) would turn it into +0 one step later than the reference
    const float fll_phase = in->fll_phase == 0.0f ? 0.0f : in->fll_phase;
    const float costas_phase = in->costas_phase == 0.0f ? 0.0f : in->costas_phase;      // (the Costas wave wraps the same way)
    SET1(h->agc_g, in->agc_gain); SET1(h->fll_ph, fll_phase); SET1(h->fll_fr, in->fll_freq);
    SET1(h->mu, in->mu); SET1(h->omega, in->omega); SET1(h->offset, in->offset);
    SET1(h->cph, costas_phase); SET1(h->cfr, in->costas_freq); SET1(h->ph2, in->ph2); SET1(h->prev, in->prev);
    const int rv = in->rrc_valid < 0 ? 0 : in->rrc_valid > (int)kGenHist ? (int)kGenHist : in->rrc_valid;
    SET1(h->rrc_valid, rv);
#undef SET1
    HIP_TRY(h, hipMemcpy(h->hist + (size_t)c * kHist, in->hist, sizeof(float2) * kHist, hipMemcpyHostToDevice));
    if (!h->far_valid) {      // the far delay line of EVERY channel reads as zeros right now: make that literal, then this channel's is current
        HIP_TRY(h, hipMemset(h->hist_far, 0, sizeof(float2) * (kGenHist - kHist) * (size_t)h->C));
        h->far_valid = true;
    }
    HIP_TRY(h, hipMemcpy(h->hist_far + (size_t)c * (kGenHist - kHist), in->hist_far, sizeof(float2) * (kGenHist - kHist), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->ybuf + (size_t)c * kYHist, in->ybuf, sizeof(float2) * kYHist, hipMemcpyHostToDevice));
    return TETRA_OK;
}

int tetra_demod_get_tables(tetra_demod_t* h, int* taps, float* rrc, int* be_taps, float* be_re, float* be_im, float* bank) {
    if (!h) return TETRA_ERR_ARG;
    const int nt = h->design.ntaps;
    if (taps) *taps = nt;
    if (be_taps) *be_taps = h->design.ntaps_be;
    if (rrc) std::memcpy(rrc, h->design.rrc.data(), sizeof(float) * nt);
    if (be_re) std::memcpy(be_re, h->design.be_re.data(), sizeof(float) * h->design.ntaps_be);
    if (be_im) std::memcpy(be_im, h->design.be_im.data(), sizeof(float) * h->design.ntaps_be);
    if (bank) std::memcpy(bank, h->design.bank.data(), sizeof(float) * kInterpPhases * kInterpTaps);
    return TETRA_OK;
}

int tetra_demod_bandedge_tap_count(tetra_demod_t* h) { return h ? h->design.ntaps_be : TETRA_ERR_ARG; }

int tetra_demod_debug_read_rrc_out(tetra_demod_t* h, float* y, int n_samples) {
    if (!h || !y || n_samples < 0 || n_samples > h->max_samples) return TETRA_ERR_ARG;
    if (!h->y) return TETRA_ERR_UNSUPPORTED;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    HIP_TRY(h, hipDeviceSynchronize());
    // rows kYHist.. of the time-major scratch hold this call's y
    const size_t C = (size_t)h->C;
    std::vector<float2> tm((size_t)n_samples * C);
    if (n_samples)
        HIP_TRY(h, hipMemcpy(tm.data(), h->y + (size_t)kYHist * C, sizeof(float2) * tm.size(), hipMemcpyDeviceToHost));
    float2* out = reinterpret_cast<float2*>(y);
    for (size_t c = 0; c < C; c++)
        for (size_t i = 0; i < (size_t)n_samples; i++) out[c * (size_t)n_samples + i] = tm[i * C + c];
    return TETRA_OK;
}

int tetra_demod_kernel_ms_history(tetra_demod_t* h, int n, float* ms) {
    if (!h || !ms || n < 1 || n > tetra_demod::kEvSlots || (long long)n > h->n_calls) return TETRA_ERR_ARG;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    for (int i = 0; i < n; i++) {
        hipEvent_t* ev = h->ev[(h->n_calls - n + i) % tetra_demod::kEvSlots];
        HIP_TRY(h, hipEventSynchronize(ev[1]));
        HIP_TRY(h, hipEventElapsedTime(&ms[i], ev[0], ev[1]));
    }
    return TETRA_OK;
}

int tetra_demod_last_kernel_ms(tetra_demod_t* h, float* ms) { return tetra_demod_kernel_ms_history(h, 1, ms); }

int tetra_demod_get_quality(tetra_demod_t* h, float* standarderr, uint8_t* sync) {
    if (!h) return TETRA_ERR_ARG;
    if (!h->q_ring) return TETRA_ERR_UNSUPPORTED;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    HIP_TRY(h, hipDeviceSynchronize());
    if (standarderr) HIP_TRY(h, hipMemcpy(standarderr, h->q_err, sizeof(float) * (size_t)h->C, hipMemcpyDeviceToHost));
    if (sync) {
        std::vector<int> tmp((size_t)h->C);
        HIP_TRY(h, hipMemcpy(tmp.data(), h->q_sync, sizeof(int) * (size_t)h->C, hipMemcpyDeviceToHost));
        for (int c = 0; c < h->C; c++) sync[c] = (uint8_t)(tmp[c] != 0);
    }
    return TETRA_OK;
}

int tetra_demod_get_constellation(tetra_demod_t* h, int first, int count, float* symbols, int32_t* n_blocks) {
    if (!h) return TETRA_ERR_ARG;
    if (!h->cd_blk) return TETRA_ERR_UNSUPPORTED;
    if (first < 0 || count < 0 || first > h->C || count > h->C - first) return TETRA_ERR_ARG;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    HIP_TRY(h, hipDeviceSynchronize());
    if (count == 0) return TETRA_OK;
    if (symbols)
        HIP_TRY(h, hipMemcpy(symbols, h->cd_blk + (size_t)first * kCdSyms, sizeof(float2) * kCdSyms * (size_t)count, hipMemcpyDeviceToHost));
    if (n_blocks) HIP_TRY(h, hipMemcpy(n_blocks, h->cd_blocks + first, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost));
    return TETRA_OK;
}

int tetra_demod_last_hip_error(tetra_demod_t* h) { return h ? h->last_hip : 0; }

int tetra_demod_debug_selftest(tetra_demod_t* h, const float* in128, float* out320) {
    if (!h || !in128 || !out320) return TETRA_ERR_ARG;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    struct Tmp {                      // freed on every return path
        float* p = nullptr;
        ~Tmp() { if (p) (void)hipFree(p); }
    } din, dout;
    HIP_TRY(h, hipMalloc((void**)&din.p, sizeof(float) * 128));
    HIP_TRY(h, hipMalloc((void**)&dout.p, sizeof(float) * 320));
    HIP_TRY(h, hipMemcpy(din.p, in128, sizeof(float) * 128, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, 0, din.p, dout.p);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpy(out320, dout.p, sizeof(float) * 320, hipMemcpyDeviceToHost));
    return TETRA_OK;
}

int tetra_demod_debug_mfma_selftest(tetra_demod_t* h, int shape, int k, const float* a, const float* b, float* d) {
    if (!h || !a || !b || !d || (shape != 16 && shape != 32) || k < 4 || (k & 3) || k > 4096) return TETRA_ERR_ARG;
    DeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    struct Tmp {
        float* p = nullptr;
        ~Tmp() { if (p) (void)hipFree(p); }
    } da, db, dd;
    const size_t na = (size_t)shape * k, nd = (size_t)shape * shape;
    HIP_TRY(h, hipMalloc((void**)&da.p, sizeof(float) * na));
    HIP_TRY(h, hipMalloc((void**)&db.p, sizeof(float) * na));
    HIP_TRY(h, hipMalloc((void**)&dd.p, sizeof(float) * nd));
    HIP_TRY(h, hipMemcpy(da.p, a, sizeof(float) * na, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(db.p, b, sizeof(float) * na, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma_selftest, dim3(1), dim3(64), 0, 0, shape, k, da.p, db.p, dd.p);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpy(d, dd.p, sizeof(float) * nd, hipMemcpyDeviceToHost));
    return TETRA_OK;
}

}  // extern "C"
