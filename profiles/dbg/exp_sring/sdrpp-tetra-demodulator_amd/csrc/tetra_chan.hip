// tetra_chan.hip -- polyphase channeliser front-end (see include/tetra_chan.h): HIP kernel + C ABI.
//
// One workgroup (256 threads) per output frame.  Weighted overlap-add: the L = P*M newest samples are weighted by the
// prototype and folded onto M bins indexed by ABSOLUTE sample time mod M (so the DFT needs no per-frame phase
// correction), then an M-point DFT, M = N1*N2, runs in LDS as N2 column DFTs of length N1, a twiddle, and N1 row DFTs of
// length N2 (direct sums: N1, N2 <= 64; for M = 800 that is 25 + 32 complex MACs per output instead of 800).  Frames go
// out time-major, out[m][k] -- exactly the TETRA_LAYOUT_TIME_MAJOR input of the demodulator.  Per second of a 20 MHz
// capture this is ~2.3 G complex MACs and 160 MB in / 320 MB out: a small fraction of the demodulator's time.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/tetra_chan.h"

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
    return TETRA_OK;
}

void free_all(tetra_chan* h) {
    void* ptrs[] = { h->xbuf, h->xalt, h->d_h, h->d_w1, h->d_w2, h->d_wm, h->st_out };
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
    if (cfg->prototype) h->proto.assign(cfg->prototype, cfg->prototype + h->L);
    else design_prototype(h->M, h->P, cfg->cutoff_rel, h->proto);
    Guard g(dev);
    if (!g.ok) { delete h; return TETRA_ERR_NO_DEVICE; }
    bool ok = hipMalloc((void**)&h->xbuf, sizeof(float2) * ((size_t)h->L - 1 + h->max_in)) == hipSuccess &&
              hipMalloc((void**)&h->xalt, sizeof(float2) * ((size_t)h->L - 1 + h->max_in)) == hipSuccess &&
              hipMalloc((void**)&h->d_h, sizeof(float) * h->L) == hipSuccess &&
              hipMalloc((void**)&h->d_w1, sizeof(float2) * h->N1) == hipSuccess &&
              hipMalloc((void**)&h->d_w2, sizeof(float2) * h->N2) == hipSuccess &&
              hipMalloc((void**)&h->d_wm, sizeof(float2) * h->M) == hipSuccess &&
              hipEventCreate(&h->ev[0]) == hipSuccess && hipEventCreate(&h->ev[1]) == hipSuccess;
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

int tetra_chan_process_device(tetra_chan_t* h, const float* d_x, int n_in, float* d_out, int* n_frames, void* hip_stream) {
    if (!h || (!d_x && n_in > 0) || !d_out || !n_frames) return TETRA_ERR_ARG;
    if (n_in < 0 || n_in > h->max_in) return TETRA_ERR_SIZE;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)hip_stream;
    const int frames = (h->phase + n_in) / h->D;
    *n_frames = frames;
    const size_t hist = (size_t)h->L - 1;
    if (n_in > 0) CH_TRY(h, hipMemcpyAsync(h->xbuf + hist, d_x, sizeof(float2) * (size_t)n_in, hipMemcpyDeviceToDevice, s));
    CH_TRY(h, hipEventRecord(h->ev[0], s));
    if (frames > 0) {
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
    if (n_in > 0) {
        CH_TRY(h, hipMemcpyAsync(h->xalt, h->xbuf + n_in, sizeof(float2) * hist, hipMemcpyDeviceToDevice, s));
        float2* t = h->xbuf; h->xbuf = h->xalt; h->xalt = t;
    }
    h->phase = (h->phase + n_in) % h->D;
    h->consumed += n_in;
    return TETRA_OK;
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
