/*
 * tetra_fast.c -- the same demodulator chain as tetra_oracle.c, written for CPU SPEED instead of bit-exactness.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline "port-fast" leg, tests/test_oracle.py's sanity check); never
 * linked into the product.  The oracle (tetra_oracle.c) pins one summation order with serial fmaf chains so that the GPU
 * can be compared bit for bit -- which makes it a slow CPU program (one dependent chain per FIR).  A CPU baseline should
 * be what a competent CPU port would look like: built -O3 -march=native -ffast-math, FIR sums split over independent
 * accumulators so that they vectorise (AVX2/AVX-512), the RRC evaluated as a blocked correlation over the whole call, libm-free
 * sine/cosine.  The algorithm, tables, state and call structure are the oracle's (same file:line anchors in the reference:
 * pi4dqpsk.cpp:132-140, fll.cpp:135-149, complex_fd.cpp:89-151, pi4dqpsk_costas.cpp:5-28, dqpsk_sym_extr.cpp:32-52,
 * bit_unpacker.cpp:4-10); floats differ from the oracle's in the last bits, decisions after lock do not.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "tetra_oracle.h"

#define FL_M_PI 3.1415926535f
#define H (TETRA_ORACLE_MAX_TAPS - 1)

static inline void fast_sincos(float x, float* s, float* c) {
    const float k = rintf(x * 0.318309886183790672f);
    float r = x - k * 3.140625f;
    r -= k * 9.67502593994140625e-4f;
    r -= k * 1.509957990978376432e-7f;
    const float z = r * r;
    float ps = ((2.597026877992903e-06f * z - 0.0001980524102691561f) * z + 0.008332998491823673f) * z - 0.16666656732559204f;
    float sr = ps * z * r + r;
    float cr = ((((-2.604826931928983e-07f * z + 2.476031113474164e-05f) * z - 0.0013888374669477344f) * z + 0.04166663810610771f) * z - 0.5f) * z + 1.0f;
    if (((int)k) & 1) { sr = -sr; cr = -cr; }
    *s = sr;
    *c = cr;
}

static inline float fast_amp(float re, float im) {
    const float r = fabsf(re), i = fabsf(im);
    return r > i ? r + 0.4f * i : i + 0.4f * r;
}

/* One call on one channel; scratch = caller-provided 4*(H+count)+... floats (see tetra_fast_scratch_floats). */
static int fast_process(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st, int count, const float* iq, float* scratch,
                        uint8_t* bits) {
    const int nt = tab->ntaps, nb = tab->ntaps_be;
    float* wr = scratch;
    float* wi = wr + (H + count + 16);
    float* yr = wi + (H + count + 16);
    float* yi = yr + (7 + count + 16);
    for (int k = 0; k < H; k++) { wr[k] = st->hist[2 * k]; wi[k] = st->hist[2 * k + 1]; }

    /* AGC + FLL: serial over samples, each band-edge sum vectorised over the taps */
    float g = st->agc_gain, ph = st->fll_phase, fr = st->fll_freq;
    const float* ba = tab->be_a;
    const float* bb = tab->be_b;
    for (int i = 0; i < count; i++) {
        const float ar = iq[2 * i] * g, ai = iq[2 * i + 1] * g;
        g += (tab->agc_set_point - sqrtf(ar * ar + ai * ai)) * tab->agc_rate;
        if (g > tab->agc_max_gain) g = tab->agc_max_gain;
        float s, c;
        fast_sincos(-ph, &s, &c);
        const float xr = ar * c - ai * s, xi = ai * c + ar * s;
        wr[H + i] = xr;
        wi[H + i] = xi;
        const float* pr = wr + i + (H - (nb - 1));
        const float* pi = wi + i + (H - (nb - 1));
        float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma omp simd reduction(+ : s1, s2, s3, s4)
        for (int k = 0; k < nb; k++) {
            s1 += pr[k] * ba[k];
            s2 += pi[k] * bb[k];
            s3 += pr[k] * bb[k];
            s4 += pi[k] * ba[k];
        }
        const float err = fast_amp(s1 + s2, s4 - s3) - fast_amp(s1 - s2, s4 + s3);
        fr += tab->fll_beta * err;
        fr = fr > tab->fll_max_freq ? tab->fll_max_freq : (fr < tab->fll_min_freq ? tab->fll_min_freq : fr);
        ph += fr + tab->fll_alpha * err;
        while (ph > FL_M_PI) ph -= 2 * FL_M_PI;
        while (ph < -FL_M_PI) ph += 2 * FL_M_PI;
    }
    st->agc_gain = g; st->fll_phase = ph; st->fll_freq = fr;

    /* RRC as a blocked correlation: for each block of outputs, taps outermost -> the inner loop is a unit-stride axpy */
    for (int k = 0; k < 7; k++) { yr[k] = st->ybuf[2 * k]; yi[k] = st->ybuf[2 * k + 1]; }
    {
        const float* h = tab->rrc;
        const int base = H - (nt - 1);
        /* samples the RRC must not see are handled by a (rare) masked prefix; the bulk runs unmasked */
        const int hidden = st->rrc_valid < nt - 1 ? (nt - 1) - st->rrc_valid : 0;   /* outputs 0..hidden-1 touch hidden samples */
        for (int i = 0; i < count && i < hidden; i++) {
            float ar = 0.f, ai = 0.f;
            const int k0 = nt - (st->rrc_valid + i + 1);
            for (int k = k0 > 0 ? k0 : 0; k < nt; k++) { ar += wr[base + i + k] * h[k]; ai += wi[base + i + k] * h[k]; }
            yr[7 + i] = ar; yi[7 + i] = ai;
        }
        enum { B = 256 };
        for (int i0 = hidden < count ? hidden : count; i0 < count; i0 += B) {
            const int n = count - i0 < B ? count - i0 : B;
            float ar[B], ai[B];
            memset(ar, 0, sizeof(float) * (size_t)n);
            memset(ai, 0, sizeof(float) * (size_t)n);
            for (int k = 0; k < nt; k++) {
                const float hk = h[k];
                const float* pr = wr + base + i0 + k;
                const float* pi = wi + base + i0 + k;
#pragma omp simd
                for (int i = 0; i < n; i++) { ar[i] += pr[i] * hk; ai[i] += pi[i] * hk; }
            }
            memcpy(yr + 7 + i0, ar, sizeof(float) * (size_t)n);
            memcpy(yi + 7 + i0, ai, sizeof(float) * (size_t)n);
        }
    }
    for (int k = 0; k < H; k++) { st->hist[2 * k] = wr[count + k]; st->hist[2 * k + 1] = wi[count + k]; }
    st->rrc_valid = (long long)st->rrc_valid + count >= H ? H : st->rrc_valid + count;

    /* timing recovery + Costas + slicer, symbol-serial */
    float mu = st->mu, om = st->omega, cph = st->costas_phase, cfr = st->costas_freq, ph2 = st->ph2;
    int offset = st->offset, S = 0;
    uint8_t prev = st->prev;
    while (offset < count) {
        int p = (int)floorf(mu * (float)TETRA_ORACLE_INTERP_PHASES);
        p = p < 0 ? 0 : (p > TETRA_ORACLE_INTERP_PHASES - 1 ? TETRA_ORACLE_INTERP_PHASES - 1 : p);
        const int p1 = p == TETRA_ORACLE_INTERP_PHASES - 1 ? p : p + 1, p0 = p == 0 ? p : p - 1;
        const float *br = yr + offset, *bi = yi + offset, *t = tab->bank[p], *t1 = tab->bank[p1], *t0 = tab->bank[p0];
        float vr = 0.f, vi = 0.f, ur = 0.f, ui = 0.f, lr = 0.f, li = 0.f;
        for (int k = 0; k < 8; k++) {
            vr += br[k] * t[k]; vi += bi[k] * t[k];
            ur += br[k] * t1[k]; ui += bi[k] * t1[k];
            lr += br[k] * t0[k]; li += bi[k] * t0[k];
        }
        const float sc = (p1 != p && p0 != p) ? 0.5f : 1.0f;
        const float dr = (ur - lr) * sc, di = (ui - li) * sc;
        float terr = (vr > 0 ? dr : -dr) + (vi > 0 ? di : -di);
        terr = terr > 1.f ? 1.f : (terr < -1.f ? -1.f : terr);
        om += tab->tr_beta * terr;
        om = om > tab->tr_max_freq ? tab->tr_max_freq : (om < tab->tr_min_freq ? tab->tr_min_freq : om);
        mu += om + tab->tr_alpha * terr;
        const float delta = floorf(mu);
        offset += (int)delta;
        mu -= delta;

        float s, c;
        fast_sincos(-cph, &s, &c);
        const float xr = vr * c - vi * s, xi = vi * c + vr * s;
        ph2 += -FL_M_PI / 4.0f;
        if (ph2 >= 2 * FL_M_PI) ph2 -= 2 * FL_M_PI;
        else if (ph2 <= -2 * FL_M_PI) ph2 += 2 * FL_M_PI;
        fast_sincos(ph2, &s, &c);
        const float zr = xr * c - xi * s, zi = xi * c + xr * s;
        float cerr = (zr > 0 ? zi : -zi) - (zi > 0 ? zr : -zr);
        cerr = cerr > 1.f ? 1.f : (cerr < -1.f ? -1.f : cerr);
        cfr += tab->costas_beta * cerr;
        cfr = cfr > tab->costas_max_freq ? tab->costas_max_freq : (cfr < tab->costas_min_freq ? tab->costas_min_freq : cfr);
        cph += cfr + tab->costas_alpha * cerr;
        while (cph > FL_M_PI) cph -= 2 * FL_M_PI;
        while (cph < -FL_M_PI) cph += 2 * FL_M_PI;

        const int a = zi < 0, b = zr < 0;
        const uint8_t sym = (uint8_t)((a << 1) | (a != b));
        static const uint8_t remap[4] = { 0, 1, 3, 2 };
        const uint8_t d = remap[(sym - prev + 4) & 3];
        prev = sym;
        bits[2 * S] = (uint8_t)(d >> 1);
        bits[2 * S + 1] = (uint8_t)(d & 1);
        S++;
    }
    offset -= count;
    for (int k = 0; k < 7; k++) { st->ybuf[2 * k] = yr[count + k]; st->ybuf[2 * k + 1] = yi[count + k]; }
    st->mu = mu; st->omega = om; st->offset = offset;
    st->costas_phase = cph; st->costas_freq = cfr; st->ph2 = ph2; st->prev = prev;
    return S;
}

/* Same contract as tetra_oracle_process_batch (without the symbol output). */
int tetra_fast_process_batch(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* states, int n_channels, int n_samples,
                             int chunk, int threads, const float* iq, uint8_t* bits, int bits_stride, int32_t* n_bits) {
    if (chunk <= 0 || chunk > n_samples) chunk = n_samples;
    int fail = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
    {
        float* scratch = (float*)malloc(sizeof(float) * (size_t)(4 * (H + chunk + 16) + 64));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int c = 0; c < n_channels; c++) {
            int nb = 0;
            for (int pos = 0; pos < n_samples; pos += chunk) {
                const int n = n_samples - pos < chunk ? n_samples - pos : chunk;
                if (nb + 2 * ((int)(n / 1.9) + 2) > bits_stride) { fail = 1; break; }   /* S <= count/1.9 + 2 like the oracle */
                nb += 2 * fast_process(tab, &states[c], n, iq + 2 * ((size_t)c * n_samples + pos), scratch,
                                       bits + (size_t)c * bits_stride + nb);
            }
            n_bits[c] = nb;
        }
        free(scratch);
    }
    return fail ? -1 : 0;
}
