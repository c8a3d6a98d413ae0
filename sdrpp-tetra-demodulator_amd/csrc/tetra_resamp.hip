// tetra_resamp.hip -- rational resampler on time-major frames (include/tetra_chan.h, round 6): HIP kernels + C ABI.
//
// Sits between the channeliser (50 ksps per channel) and the demodulator, which then runs at the reference instance's own rate:
// VFO_SAMPLERATE 36000, 2 samples per symbol (/root/reference/src/main.cpp:35,75,84).  Bound by its HBM traffic: per input frame of C
// channels 8 C bytes are read and 8 C I / DN written (config 5, a quarter second: 80 MB + 57.6 MB); the arithmetic (2 T flop per
// output float) is a tenth of what the vector pipes do in that time.  Thread-level code: resamp_core.hpp.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/tetra_chan.h"
#include "resamp_core.hpp"

namespace {

constexpr int kThreads = 256;

// Workgroups are dealt to the 8 XCDs round-robin (each XCD has its own L2) and consecutive groups share T - 1 of their DN + T - 1
// rows: the XCD a workgroup lands on (blockIdx.x mod 8) takes a CONTIGUOUS range of the launch's thread blocks, so that a row's
// second reader finds it in the L2 its first reader filled.
__device__ __forceinline__ long long remapped_block(int span) {
    return (long long)(blockIdx.x & 7) * span + (long long)(blockIdx.x >> 3);
}

template <int I, int DN, int T, int W> __global__ __launch_bounds__(kThreads) void k_resample(resamp::Ctx c, int span, long long blocks) {
    const long long b = remapped_block(span);
    if (b >= blocks) return;
    resamp::thread_fixed<I, DN, T, W>(c, b * kThreads + threadIdx.x);
}

template <int W> __global__ __launch_bounds__(kThreads) void k_resample_generic(resamp::Ctx c, int span, long long blocks) {
    const long long b = remapped_block(span);
    if (b >= blocks) return;
    resamp::thread_generic<W>(c, b * kThreads + threadIdx.x);
}

typedef void (*fixed_kernel_t)(resamp::Ctx, int, long long);

template <int I, int DN, int T> fixed_kernel_t pick_w(int W) {
    return W == 4 ? (fixed_kernel_t)k_resample<I, DN, T, 4> : (fixed_kernel_t)k_resample<I, DN, T, 2>;
}

fixed_kernel_t pick_fixed(int I, int DN, int T, int W) {
    if (I == 18 && DN == 25) {
        if (T == 8) return pick_w<18, 25, 8>(W);
        if (T == 12) return pick_w<18, 25, 12>(W);
        if (T == 16) return pick_w<18, 25, 16>(W);
        if (T == 24) return pick_w<18, 25, 24>(W);
    }
    if (T == 8) {
        if (I == 2 && DN == 3) return pick_w<2, 3, 8>(W);
        if (I == 3 && DN == 2) return pick_w<3, 2, 8>(W);
        if (I == 1 && DN == 2) return pick_w<1, 2, 8>(W);
    }
    return nullptr;
}

double bessel_i0(double x) {
    double s = 1.0, t = 1.0;
    for (int k = 1; k < 60; k++) {
        t *= (x / (2.0 * k)) * (x / (2.0 * k));
        s += t;
        if (t < 1e-18 * s) break;
    }
    return s;
}

// Kaiser-windowed sinc at the zero-stuffed rate, cutoff fc = cutoff_rel / (2 max(I, DN)) cycles per sample there (= cutoff_rel x the
// narrower of the input and output Nyquist bands), DC gain I.
void design_prototype(int I, int DN, int T, double cutoff_rel, double beta, std::vector<float>& h) {
    const int L = I * T;
    const double pi = 3.14159265358979323846, fc = cutoff_rel / (2.0 * (double)(I > DN ? I : DN));
    std::vector<double> t(L);
    double sum = 0.0;
    for (int l = 0; l < L; l++) {
        const double u = (double)l - 0.5 * (double)(L - 1);
        const double sinc = (u == 0.0) ? 2.0 * fc : std::sin(2.0 * pi * fc * u) / (pi * u);
        const double r = L > 1 ? 2.0 * u / (double)(L - 1) : 0.0;
        const double w = bessel_i0(beta * std::sqrt(1.0 - r * r > 0 ? 1.0 - r * r : 0.0)) / bessel_i0(beta);
        t[l] = sinc * w;
        sum += t[l];
    }
    h.resize(L);
    for (int l = 0; l < L; l++) h[l] = (float)(t[l] / sum * (double)I);
}

struct Guard {
    int prev = -1;
    bool ok;
    explicit Guard(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; ok = hipSetDevice(d) == hipSuccess; }
    ~Guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct tetra_resamp {
    tetra_resamp_config_t cfg;
    int device = 0, last_hip = 0;
    int C = 0, I = 0, DN = 0, T = 0, W = 4, units = 0, max_in = 0;
    std::vector<float> proto;
    fixed_kernel_t fixed = nullptr;
    float* d_coef = nullptr;     // [I][T] phase table (fixed kernel) or the prototype (generic)
    float* hist = nullptr;       // [T - 1][2 C]: the frames before the next call's first
    float* halt = nullptr;       // same size: receives the next call's delay line, then the two swap roles
    long long n_total = 0;       // frames consumed so far
    long long m_next = 0;        // outputs emitted so far
    float* st_in = nullptr;      // host-path staging
    float* st_out = nullptr;
    size_t st_in_frames = 0, st_out_frames = 0;
    hipEvent_t ev[2] = { nullptr, nullptr };
    bool ev_valid = false;
};

#define RS_TRY(h, expr)                                   \
    do {                                                  \
        hipError_t e__ = (expr);                          \
        if (e__ != hipSuccess) {                          \
            (h)->last_hip = (int)e__;                     \
            return TETRA_ERR_HIP;                         \
        }                                                 \
    } while (0)

namespace {

void free_all(tetra_resamp* h) {
    void* ptrs[] = { h->d_coef, h->hist, h->halt, h->st_in, h->st_out };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
}

size_t hist_bytes(const tetra_resamp* h) { return sizeof(float) * 2 * (size_t)h->C * (size_t)(h->T - 1); }

}  // namespace

extern "C" {

int tetra_resamp_default_config(tetra_resamp_config_t* cfg) {
    if (!cfg) return TETRA_ERR_ARG;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->n_channels = 800;
    cfg->interp = 18;
    cfg->decim = 25;
    cfg->taps_per_phase = 16;
    cfg->max_in = 1 << 16;
    cfg->device = -1;
    cfg->cutoff_rel = 1.0;
    cfg->kaiser_beta = 6.0;
    return TETRA_OK;
}

int tetra_resamp_create(const tetra_resamp_config_t* cfg, tetra_resamp_t** out) {
    if (!cfg || !out) return TETRA_ERR_ARG;
    *out = nullptr;
    if (cfg->n_channels < 1 || cfg->interp < 1 || cfg->decim < 1 || cfg->taps_per_phase < 2 || cfg->taps_per_phase > 64 ||
        cfg->interp > 4096 || cfg->decim > 4096 || cfg->max_in < 1 || cfg->reserved != 0 ||
        (cfg->flags & ~(TETRA_RESAMP_FLAG_GENERIC | TETRA_RESAMP_FLAG_NARROW_UNITS)) || !(cfg->cutoff_rel > 0) || !(cfg->kaiser_beta >= 0))
        return TETRA_ERR_ARG;
    // the outputs of one call are addressed with 32-bit frame counts
    if ((long long)cfg->max_in * cfg->interp / cfg->decim + 2 > 0x7fffffffLL) return TETRA_ERR_SIZE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TETRA_ERR_NO_DEVICE;
    int dev = cfg->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return TETRA_ERR_NO_DEVICE;
    if (dev >= ndev) return TETRA_ERR_NO_DEVICE;
    tetra_resamp* h = new (std::nothrow) tetra_resamp();
    if (!h) return TETRA_ERR_NOMEM;
    h->cfg = *cfg;
    h->cfg.prototype = nullptr;
    h->device = dev;
    h->C = cfg->n_channels; h->I = cfg->interp; h->DN = cfg->decim; h->T = cfg->taps_per_phase; h->max_in = cfg->max_in;
    h->W = (h->C % 2 == 0 && !(cfg->flags & TETRA_RESAMP_FLAG_NARROW_UNITS)) ? 4 : 2;                     // 16-byte lane units when a row divides into them
    h->units = 2 * h->C / h->W;
    if (cfg->prototype) h->proto.assign(cfg->prototype, cfg->prototype + (size_t)h->I * h->T);
    else design_prototype(h->I, h->DN, h->T, cfg->cutoff_rel, cfg->kaiser_beta, h->proto);
    h->fixed = (cfg->flags & TETRA_RESAMP_FLAG_GENERIC) ? nullptr : pick_fixed(h->I, h->DN, h->T, h->W);
    Guard g(dev);
    if (!g.ok) { delete h; return TETRA_ERR_NO_DEVICE; }
    std::vector<float> coef((size_t)h->I * h->T);
    if (h->fixed) resamp::phase_table(h->proto.data(), h->I, h->DN, h->T, coef.data());
    else coef = h->proto;
    bool ok = hipMalloc((void**)&h->d_coef, sizeof(float) * coef.size()) == hipSuccess &&
              hipMalloc((void**)&h->hist, hist_bytes(h)) == hipSuccess && hipMalloc((void**)&h->halt, hist_bytes(h)) == hipSuccess &&
              hipEventCreate(&h->ev[0]) == hipSuccess && hipEventCreate(&h->ev[1]) == hipSuccess;
    int rc = ok ? TETRA_OK : TETRA_ERR_NOMEM;
    if (rc == TETRA_OK && (hipMemcpy(h->d_coef, coef.data(), sizeof(float) * coef.size(), hipMemcpyHostToDevice) != hipSuccess ||
                           hipMemset(h->hist, 0, hist_bytes(h)) != hipSuccess))
        rc = TETRA_ERR_HIP;
    if (rc != TETRA_OK) { free_all(h); delete h; return rc; }
    *out = h;
    return TETRA_OK;
}

int tetra_resamp_destroy(tetra_resamp_t* h) {
    if (!h) return TETRA_ERR_ARG;
    Guard g(h->device);
    (void)hipDeviceSynchronize();
    free_all(h);
    delete h;
    return TETRA_OK;
}

int tetra_resamp_frames_for(tetra_resamp_t* h, int n_in) {
    if (!h || n_in < 0) return TETRA_ERR_ARG;
    return (int)(resamp::outputs_after(h->n_total + n_in, h->I, h->DN) - h->m_next);
}

int tetra_resamp_process_device(tetra_resamp_t* h, const float* d_in, int n_in, float* d_out, int* n_out, void* hip_stream) {
    if (!h || (!d_in && n_in > 0) || !d_out || !n_out) return TETRA_ERR_ARG;
    if (n_in < 0 || n_in > h->max_in) return TETRA_ERR_SIZE;
    const size_t amask = h->W == 4 ? 15 : 7;
    if (((uintptr_t)d_in & amask) || ((uintptr_t)d_out & amask)) return TETRA_ERR_ALIGN;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)hip_stream;
    const long long m1 = resamp::outputs_after(h->n_total + n_in, h->I, h->DN);
    const long long n_new = m1 - h->m_next;
    *n_out = (int)n_new;
    RS_TRY(h, hipEventRecord(h->ev[0], s));
    if (n_new > 0) {
        resamp::Ctx c;
        c.x = d_in; c.hist = h->hist; c.out = d_out; c.coef = h->d_coef;
        c.n0 = h->n_total; c.m0 = h->m_next; c.m1 = m1; c.n_in = n_in; c.units = h->units;
        c.I = h->I; c.DN = h->DN; c.T = h->T;
        long long threads;
        if (h->fixed) threads = ((m1 + h->I - 1) / h->I - h->m_next / h->I) * (long long)h->units;
        else threads = n_new * (long long)h->units;
        const long long blocks = (threads + kThreads - 1) / kThreads;
        const int span = (int)((blocks + 7) / 8);
        if (h->fixed) hipLaunchKernelGGL(h->fixed, dim3(8 * span), dim3(kThreads), 0, s, c, span, blocks);
        else if (h->W == 4) hipLaunchKernelGGL(k_resample_generic<4>, dim3(8 * span), dim3(kThreads), 0, s, c, span, blocks);
        else hipLaunchKernelGGL(k_resample_generic<2>, dim3(8 * span), dim3(kThreads), 0, s, c, span, blocks);
        RS_TRY(h, hipGetLastError());
    }
    RS_TRY(h, hipEventRecord(h->ev[1], s));
    h->ev_valid = true;
    // carry: the last T - 1 frames of [delay line | new] become the next call's delay line -- into the OTHER buffer (an in-place
    // move would overlap for n_in < T - 1), then the two swap roles.  Stream order keeps the kernel ahead of the copies.
    if (n_in > 0) {
        const size_t row = sizeof(float) * 2 * (size_t)h->C, hist = (size_t)h->T - 1;
        const size_t from_x = (size_t)n_in < hist ? (size_t)n_in : hist, keep = hist - from_x;
        if (keep) RS_TRY(h, hipMemcpyAsync(h->halt, (const char*)h->hist + row * (size_t)n_in, row * keep, hipMemcpyDeviceToDevice, s));
        RS_TRY(h, hipMemcpyAsync((char*)h->halt + row * keep, (const char*)d_in + row * ((size_t)n_in - from_x), row * from_x,
                                 hipMemcpyDeviceToDevice, s));
        float* t = h->hist; h->hist = h->halt; h->halt = t;
    }
    h->n_total += n_in;
    h->m_next = m1;
    return TETRA_OK;
}

int tetra_resamp_process(tetra_resamp_t* h, const float* in, int n_in, float* out, int* n_out) {
    if (!h || (!in && n_in > 0) || !out || !n_out) return TETRA_ERR_ARG;
    if (n_in < 0 || n_in > h->max_in) return TETRA_ERR_SIZE;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    const size_t row = sizeof(float) * 2 * (size_t)h->C;
    const size_t frames = (size_t)(resamp::outputs_after(h->n_total + n_in, h->I, h->DN) - h->m_next);
    auto grow = [&](float*& p, size_t& have, size_t want) -> int {
        if (want <= have && p) return TETRA_OK;
        if (p) (void)hipFree(p);
        p = nullptr; have = 0;
        RS_TRY(h, hipMalloc((void**)&p, row * (want ? want : 1)));
        have = want ? want : 1;
        return TETRA_OK;
    };
    int rc = grow(h->st_in, h->st_in_frames, (size_t)n_in);
    if (rc == TETRA_OK) rc = grow(h->st_out, h->st_out_frames, frames);
    if (rc != TETRA_OK) return rc;
    if (n_in > 0) RS_TRY(h, hipMemcpy(h->st_in, in, row * (size_t)n_in, hipMemcpyHostToDevice));
    rc = tetra_resamp_process_device(h, h->st_in, n_in, h->st_out, n_out, nullptr);
    if (rc == TETRA_OK) {
        hipError_t e = hipStreamSynchronize(0);
        if (e == hipSuccess && frames) e = hipMemcpy(out, h->st_out, row * frames, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { h->last_hip = (int)e; rc = TETRA_ERR_HIP; }
    }
    return rc;
}

int tetra_resamp_reset(tetra_resamp_t* h) {
    if (!h) return TETRA_ERR_ARG;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    RS_TRY(h, hipDeviceSynchronize());
    RS_TRY(h, hipMemset(h->hist, 0, hist_bytes(h)));
    h->n_total = 0;
    h->m_next = 0;
    return TETRA_OK;
}

int tetra_resamp_get_prototype(tetra_resamp_t* h, float* proto) {
    if (!h || !proto) return TETRA_ERR_ARG;
    std::memcpy(proto, h->proto.data(), sizeof(float) * h->proto.size());
    return TETRA_OK;
}

int tetra_resamp_last_kernel_ms(tetra_resamp_t* h, float* ms) {
    if (!h || !ms || !h->ev_valid) return TETRA_ERR_ARG;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    RS_TRY(h, hipEventSynchronize(h->ev[1]));
    RS_TRY(h, hipEventElapsedTime(ms, h->ev[0], h->ev[1]));
    return TETRA_OK;
}

}  // extern "C"
