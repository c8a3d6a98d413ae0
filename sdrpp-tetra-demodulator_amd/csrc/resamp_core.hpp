// resamp_core.hpp -- rational resampler I / DN on time-major frames [frame][channel] (complex64), thread level.
//
// BASELINE config 5 at the plugin's operating point: the channeliser's 800 x 25 kHz channels leave it at 50 ksps (2 x oversampled
// bank); the reference instance is created at VFO_SAMPLERATE 36000 with 2 samples per symbol (/root/reference/src/main.cpp:35,75,84),
// so the frames go through an 18 / 25 polyphase resampler before the demodulator (SURVEY.md section 8(f) #1 names this route).
//
//   y[m][c] = sum_{j < T} h[r_m + I j] x[q_m - j][c],      I q_m + r_m = DN m,  0 <= r_m < I          (include/tetra_chan.h)
//
// = zero-stuff by I, low-pass h (I T taps, gain I), keep every DN-th.  The filter is real and the same for every channel, so a frame
// row is 2 C independent floats: a lane owns a UNIT of W consecutive floats of the row (W = 4: two channels, 16-byte accesses) and one
// GROUP of I consecutive outputs m = I G + i -- a whole turn of the phase wheel, so that r_i = DN i mod I and q_i = floor(DN i / I)
// are compile-time constants of the unrolled loop: the T coefficients of output i are uniform over the wavefront (scalar loads, they
// sit in SGPRs), and of the DN + T - 1 input rows a group reads every one is loaded once and used from registers.  Consecutive lanes
// = consecutive units: every load / store instruction of a wavefront covers one contiguous 1 KB run of a row.
//
// Compiles for the host as well (tests/emul/resamp_emul.cpp runs every thread of a launch against the double-precision definition).
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#define RESAMP_HD __host__ __device__ __forceinline__
#else
#define RESAMP_HD inline
#endif
#if defined(__clang__)
#define RESAMP_FP_FAST _Pragma("clang fp contract(fast)")
#else
#define RESAMP_FP_FAST
#endif

namespace resamp {

struct alignas(16) f4 { float v[4]; };
struct alignas(8) f2 { float v[2]; };
template <int W> struct unit_of;
template <> struct unit_of<4> { using type = f4; };
template <> struct unit_of<2> { using type = f2; };

struct Ctx {
    const float* x;        // the call's n_in new frames [n_in][2 C] floats, where the caller left them
    const float* hist;     // the T - 1 frames before them, oldest first (the handle's delay line)
    float* out;            // [m1 - m0][2 C]
    const float* coef;     // [I][T]: coef[i][j] = h[r_i + I j], r_i = DN i mod I      (fixed-ratio kernel)
                           // [I T] : the prototype h itself                           (generic kernel)
    long long n0;          // absolute index of the first new frame
    long long m0, m1;      // absolute output indices this call emits: [m0, m1)
    int n_in;
    int units;             // lane units per row: 2 C / W
    int I, DN, T;          // (generic kernel only)
};

// Row `rel` of the stream counted from the call's first new frame.  Interior groups read only new frames (CAREFUL = false).  The
// groups at a call's two ends reach back into the delay line (rel < 0) or ask for rows the call does not hold -- rows that only
// feed outputs outside [m0, m1), which are not stored: those read the nearest row there is.
template <int W, bool CAREFUL> RESAMP_HD typename unit_of<W>::type row_at(const Ctx& c, long long rel, int u, int T) {
    using V = typename unit_of<W>::type;
    const long long stride = (long long)c.units;
    if (!CAREFUL) return reinterpret_cast<const V*>(c.x)[rel * stride + u];
    if (rel < 0) {
        const long long k = (T - 1) + rel;
        return reinterpret_cast<const V*>(c.hist)[(k < 0 ? 0 : k) * stride + u];
    }
    return reinterpret_cast<const V*>(c.x)[(rel < c.n_in ? rel : c.n_in - 1) * stride + u];
}

// One group of I outputs (absolute group G, i.e. outputs I G .. I G + I - 1) for lane unit u.
template <int I, int DN, int T, int W, bool CAREFUL> RESAMP_HD void group_t(const Ctx& c, long long G, int u) {
    RESAMP_FP_FAST
    using V = typename unit_of<W>::type;
    constexpr int kQmax = (DN * (I - 1)) / I;            // newest row the group's last output reads
    constexpr int kRows = kQmax + T;                      // rows -(T - 1) .. kQmax relative to row DN G
    const long long base = (long long)DN * G - c.n0;      // row DN G counted from the call's first new frame
    V rows[kRows];
#pragma unroll
    for (int k = 0; k < kRows; k++) rows[k] = row_at<W, CAREFUL>(c, base + (k - (T - 1)), u, T);
    V* const out = reinterpret_cast<V*>(c.out);
#pragma unroll
    for (int i = 0; i < I; i++) {
        const int q = (DN * i) / I;                       // compile-time after unrolling
        float acc[W];
#pragma unroll
        for (int w = 0; w < W; w++) acc[w] = 0.f;
#pragma unroll
        for (int j = T - 1; j >= 0; j--) {                 // oldest row first
            const float h = c.coef[i * T + j];
            const V xv = rows[q - j + (T - 1)];
#pragma unroll
            for (int w = 0; w < W; w++) acc[w] += h * xv.v[w];
        }
        const long long m = (long long)I * G + i;
        if (!CAREFUL || (m >= c.m0 && m < c.m1)) {
            V o;
#pragma unroll
            for (int w = 0; w < W; w++) o.v[w] = acc[w];
            out[(m - c.m0) * (long long)c.units + u] = o;
        }
    }
}

// is group G an interior one?  (all I outputs inside [m0, m1) and every row it reads among the new frames)
template <int I, int DN, int T> RESAMP_HD bool group_is_careful(const Ctx& c, long long G) {
    constexpr int kQmax = (DN * (I - 1)) / I;
    const long long base = (long long)DN * G - c.n0;
    return (long long)I * G < c.m0 || (long long)I * G + I > c.m1 || base - (T - 1) < 0 || base + kQmax > (long long)c.n_in - 1;
}

// Thread t of the launch: groups G0 .. of the call x units, flattened (a wavefront may straddle two groups: the coefficients do not
// depend on the group, so they stay wave-uniform).
template <int I, int DN, int T, int W> RESAMP_HD void thread_fixed(const Ctx& c, long long t) {
    const long long G0 = c.m0 / I, G1 = (c.m1 + I - 1) / I;         // groups [G0, G1) hold the call's outputs
    const long long g = t / c.units;
    const int u = (int)(t - g * c.units);
    const long long G = G0 + g;
    if (G >= G1) return;
    if (group_is_careful<I, DN, T>(c, G)) group_t<I, DN, T, W, true>(c, G, u);
    else group_t<I, DN, T, W, false>(c, G, u);
}

// Any ratio, any length (run-time I, DN, T): one output per (thread, unit); coefficients from the prototype in memory.
template <int W> RESAMP_HD void thread_generic(const Ctx& c, long long t) {
    RESAMP_FP_FAST
    using V = typename unit_of<W>::type;
    const long long k = t / c.units;
    const int u = (int)(t - k * c.units);
    const long long m = c.m0 + k;
    if (m >= c.m1) return;
    const long long q = ((long long)c.DN * m) / c.I;
    const int r = (int)((long long)c.DN * m - q * c.I);
    float acc[W];
#pragma unroll
    for (int w = 0; w < W; w++) acc[w] = 0.f;
    for (int j = c.T - 1; j >= 0; j--) {
        const float h = c.coef[r + c.I * j];
        const V xv = row_at<W, true>(c, q - j - c.n0, u, c.T);
#pragma unroll
        for (int w = 0; w < W; w++) acc[w] += h * xv.v[w];
    }
    V o;
#pragma unroll
    for (int w = 0; w < W; w++) o.v[w] = acc[w];
    reinterpret_cast<V*>(c.out)[k * (long long)c.units + u] = o;
}

// coef[i][j] = h[(DN i mod I) + I j]
inline void phase_table(const float* h, int I, int DN, int T, float* coef) {
    for (int i = 0; i < I; i++)
        for (int j = 0; j < T; j++) coef[i * T + j] = h[(DN * i) % I + I * j];
}

// outputs that exist once n frames have arrived: m with floor(DN m / I) <= n - 1
inline long long outputs_after(long long n, int I, int DN) { return (n * I + DN - 1) / DN; }

}  // namespace resamp
