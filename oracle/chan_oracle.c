/*
 * chan_oracle.c -- CPU restatement of the polyphase channeliser front-end (TEST INFRASTRUCTURE ONLY).
 *
 * The reference plugin has no channeliser: it asks SDR++ for one VFO per instance (src/main.cpp:75,
 * sigpath::vfoManager.createVFO(..., VFO_SAMPLERATE ...)), i.e. a per-channel DDC that lives in SDR++
 * core.  BASELINE.json config 5 / SURVEY.md section 8(f) #1 replace N such VFOs by one analysis filter
 * bank.  There is therefore no reference code to follow; this file states the DEFINITION the GPU kernel
 * is checked against, evaluated the slow, obvious way in double precision:
 *
 *   y_k[m] = sum_{l=0}^{L-1} h[l] * x[m*D - l] * exp(-j*2*pi*k*(m*D - l)/M),   k = 0..M-1
 *
 * = channel k is the band centred at k*Fs/M (k > M/2: negative frequencies), mixed to baseband, low-pass
 * filtered by the prototype h (L = P*M taps) and decimated by D.  Samples before the start of the stream
 * (index < 0) come from the carried history (zeros initially).  Frames are emitted time-major: out[m][k].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CH_PI 3.14159265358979323846

/* Prototype low-pass: Kaiser-windowed sinc (beta 9), cutoff = cutoff_rel * (Fs/M)/2 ... in cycles/sample
 * fc = cutoff_rel / (2*M); unity DC gain.  Same formula as csrc/chan_design.hpp. */
static double bessel_i0(double x) {
    double s = 1.0, t = 1.0;
    for (int k = 1; k < 60; k++) {
        t *= (x / (2.0 * k)) * (x / (2.0 * k));
        s += t;
        if (t < 1e-18 * s) break;
    }
    return s;
}

void chan_oracle_prototype(int M, int P, double cutoff_rel, float* h) {
    const int L = M * P;
    const double fc = cutoff_rel / (2.0 * (double)M);
    const double beta = 9.0;
    double sum = 0.0;
    double* t = (double*)malloc(sizeof(double) * (size_t)L);
    for (int l = 0; l < L; l++) {
        const double u = (double)l - 0.5 * (double)(L - 1);
        const double sinc = (u == 0.0) ? 2.0 * fc : sin(2.0 * CH_PI * fc * u) / (CH_PI * u);
        const double r = 2.0 * u / (double)(L - 1);
        const double w = bessel_i0(beta * sqrt(1.0 - r * r > 0 ? 1.0 - r * r : 0.0)) / bessel_i0(beta);
        t[l] = sinc * w;
        sum += t[l];
    }
    for (int l = 0; l < L; l++) h[l] = (float)(t[l] / sum);
    free(t);
}

/* x: n_in wideband samples (re,im); hist: the L-1 samples preceding x[0] (oldest first), updated on return.
 * Emits frames m = 0 .. n_frames-1 with n_frames = floor((phase + n_in) / D) where `phase` in [0, D) is the number of
 * samples already consumed towards the next frame (carried in *phase).  Frame j of this call is aligned so that its
 * newest sample is x[(j+1)*D - 1 - phase0].  out: [n_frames][M] complex (re,im).  Returns n_frames. */
/* The same, for the n_sel channels sel[] only (out: [n_frames][n_sel]); sel == NULL: all M channels (out: [n_frames][M]). */
int chan_oracle_process_channels(int M, int P, int D, const float* h, float* hist, int* phase, int64_t* frame_index,
                                 int n_in, const float* x, float* out, int n_sel, const int32_t* sel);

int chan_oracle_process(int M, int P, int D, const float* h, float* hist, int* phase, int64_t* frame_index,
                        int n_in, const float* x, float* out) {
    return chan_oracle_process_channels(M, P, D, h, hist, phase, frame_index, n_in, x, out, M, NULL);
}

int chan_oracle_process_channels(int M, int P, int D, const float* h, float* hist, int* phase, int64_t* frame_index,
                                 int n_in, const float* x, float* out, int n_sel, const int32_t* sel) {
    const int L = M * P;
    const int ph0 = *phase;
    const int n_frames = (ph0 + n_in) / D;
    /* linear buffer: hist (L-1) + x */
    double* br = (double*)malloc(sizeof(double) * (size_t)(L - 1 + n_in) * 2);
    for (int i = 0; i < L - 1; i++) { br[2 * i] = hist[2 * i]; br[2 * i + 1] = hist[2 * i + 1]; }
    for (int i = 0; i < n_in; i++) { br[2 * (L - 1 + i)] = x[2 * i]; br[2 * (L - 1 + i) + 1] = x[2 * i + 1]; }
    const int64_t f0 = *frame_index;
    /* exp(-j 2 pi q / M) for q = 0 .. M-1: the phasor of the definition depends on k n mod M only, so it is evaluated once per
     * residue instead of once per (frame, channel, tap) -- the same double-precision values, memoised */
    double* ct = (double*)malloc(sizeof(double) * (size_t)M * 2);
    for (int q = 0; q < M; q++) {
        const double ang = -2.0 * CH_PI * (double)q / (double)M;
        ct[2 * q] = cos(ang);
        ct[2 * q + 1] = sin(ang);
    }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int j = 0; j < n_frames; j++) {
        const int newest = (j + 1) * D - 1 - ph0;            /* index into x of this frame's newest sample */
        /* absolute sample time of the newest sample: n_abs = (f0 + j + 1)*D - 1  (stream starts at n = 0) */
        const int64_t n_abs = (f0 + j + 1) * (int64_t)D - 1;
        for (int ks = 0; ks < n_sel; ks++) {
            const int k = sel ? sel[ks] : ks;
            double ar = 0.0, ai = 0.0;
            for (int l = 0; l < L; l++) {
                const int bi = (L - 1) + newest - l;          /* >= 0 */
                const int64_t n = n_abs - l;
                const int64_t kn = ((int64_t)k * (n % M + M)) % M; /* k*n mod M */
                const double c = ct[2 * kn], s = ct[2 * kn + 1];
                const double xr = br[2 * bi], xi = br[2 * bi + 1];
                ar += (double)h[l] * (xr * c - xi * s);
                ai += (double)h[l] * (xr * s + xi * c);
            }
            out[((size_t)j * n_sel + ks) * 2] = (float)ar;
            out[((size_t)j * n_sel + ks) * 2 + 1] = (float)ai;
        }
    }
    /* carry */
    const int consumed = n_in;
    for (int i = 0; i < L - 1; i++) { hist[2 * i] = (float)br[2 * (consumed + i)]; hist[2 * i + 1] = (float)br[2 * (consumed + i) + 1]; }
    *phase = (ph0 + n_in) % D;
    *frame_index = f0 + n_frames;
    free(ct);
    free(br);
    return n_frames;
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * Rational resampler I / DN on time-major frames -- the DEFINITION the GPU kernel (csrc/tetra_resamp.hip) is checked against.
 * The reference has no resampler either: SDR++'s VFO hands every plugin instance a 36 ksps stream (src/main.cpp:35,75,
 * VFO_SAMPLERATE 36000; the demodulator is initialised with 2 samples per symbol, :84).  Config 5's bank emits 50 ksps per
 * channel; SURVEY.md section 8(f) #1 names the 18 / 25 rational resampler that brings the frames to the plugin's rate.
 * Textbook form, evaluated in double precision: zero-stuff by I, filter with h (I*T taps), keep every DN-th sample:
 *
 *   u[I k] = x[k], 0 elsewhere;   v[p] = sum_i h[i] u[p - i];   y[m] = v[DN m]
 *   =>  y[m] = sum_{j < T} h[r + I j] x[q - j],   DN m = I q + r,  0 <= r < I.
 * --------------------------------------------------------------------------------------------------------------------------- */
/* Kaiser(beta)-windowed sinc at the zero-stuffed rate, cutoff fc = cutoff_rel / (2 max(I, DN)) cycles per sample there, DC gain I.
 * Same formula as design_prototype in csrc/tetra_resamp.hip. */
void resamp_oracle_prototype(int I, int DN, int T, double cutoff_rel, double beta, float* h) {
    const int L = I * T;
    const double fc = cutoff_rel / (2.0 * (double)(I > DN ? I : DN));
    double sum = 0.0;
    double* t = (double*)malloc(sizeof(double) * (size_t)L);
    for (int l = 0; l < L; l++) {
        const double u = (double)l - 0.5 * (double)(L - 1);
        const double sinc = (u == 0.0) ? 2.0 * fc : sin(2.0 * CH_PI * fc * u) / (CH_PI * u);
        const double r = L > 1 ? 2.0 * u / (double)(L - 1) : 0.0;
        const double w = bessel_i0(beta * sqrt(1.0 - r * r > 0 ? 1.0 - r * r : 0.0)) / bessel_i0(beta);
        t[l] = sinc * w;
        sum += t[l];
    }
    for (int l = 0; l < L; l++) h[l] = (float)(t[l] / sum * (double)I);
    free(t);
}

/* x: [n_in][C] complex frames; hist: the T - 1 frames before x[0] (oldest first; [T-1][C] complex), updated on return;
 * *n_total = frames consumed before this call, *m_next = outputs emitted before it (both updated).  Emits every output m whose newest
 * input frame floor(DN m / I) has arrived; out: [..][C] complex.  Returns the number of output frames. */
int resamp_oracle_process(int I, int DN, int T, const float* h, int C, float* hist, int64_t* n_total, int64_t* m_next,
                          int n_in, const float* x, float* out) {
    const int64_t n0 = *n_total, n1 = n0 + n_in;
    const int64_t m0 = *m_next, m1 = (n1 * I + DN - 1) / DN;
    const int H = T - 1;
    const size_t row = (size_t)C * 2;
    /* linear buffer of rows: hist (H) + x */
    float* br = (float*)malloc(sizeof(float) * row * (size_t)(H + n_in + 1));
    memcpy(br, hist, sizeof(float) * row * (size_t)H);
    if (n_in > 0) memcpy(br + row * (size_t)H, x, sizeof(float) * row * (size_t)n_in);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t m = m0; m < m1; m++) {
        const int64_t q = ((int64_t)DN * m) / I;
        const int r = (int)((int64_t)DN * m - q * I);
        for (size_t f = 0; f < row; f++) {
            double acc = 0.0;
            for (int j = 0; j < T; j++) {
                const int64_t bi = (q - j) - n0 + H;      /* >= 0: the delay line holds what an output of this call reaches back to */
                acc += (double)h[r + I * j] * (double)br[row * (size_t)bi + f];
            }
            out[row * (size_t)(m - m0) + f] = (float)acc;
        }
    }
    memcpy(hist, br + row * (size_t)n_in, sizeof(float) * row * (size_t)H);
    *n_total = n1;
    *m_next = m1;
    free(br);
    return (int)(m1 - m0);
}
