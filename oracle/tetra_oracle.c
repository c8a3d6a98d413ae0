/*
 * tetra_oracle.c -- CPU restatement of the reference demodulator chain.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see tetra_oracle.h for both notes
 * and for the arithmetic contract this file defines.
 *
 * All file:line citations are relative to the reference repository
 * cropinghigh/sdrpp-tetra-demodulator.  "SDR++ core" marks semantics of the
 * un-vendored AlexandreRouma/SDRPlusPlus core/src/dsp headers (unpinned, master;
 * SURVEY.md Appendix A) that the reference calls into.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -fopenmp -shared -fPIC (see Makefile).
 */
#include "tetra_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* SDR++ core dsp/types.h: FL_M_PI */
#define FL_M_PI 3.1415926535f
#define DB_M_PI 3.14159265358979323846

/* ------------------------------------------------------------------------- */
/* Run-time phasor: fixed-polynomial sin/cos (replaces the libm cosf/sinf of      */
/* SDR++ core math::phasor, called at fll.cpp:137, pi4dqpsk_costas.cpp:7,16).   */
/* Three-term Cody-Waite reduction by pi (k = rint(x/pi), r in [-pi/2, pi/2]),   */
/* minimax polynomials in r^2 (degree-9 sine, degree-10 cosine, fitted for this */
/* project; |error| <= 1.6e-7 over [-2pi, 2pi]), sign = (-1)^k for both.         */
/* ------------------------------------------------------------------------- */
void tetra_oracle_sincosf(float x, float* s, float* c) {
    float k = rintf(x * 0.318309886183790672f);
    float r = fmaf(-k, 3.140625f, x);
    r = fmaf(-k, 9.67502593994140625e-4f, r);
    r = fmaf(-k, 1.509957990978376432e-7f, r);
    float z = r * r;
    float ps = fmaf(2.597026877992903e-06f, z, -0.0001980524102691561f);
    ps = fmaf(ps, z, 0.008332998491823673f);
    ps = fmaf(ps, z, -0.16666656732559204f);
    ps = ps * z;
    float sr = fmaf(ps, r, r);
    float pc = fmaf(-2.604826931928983e-07f, z, 2.476031113474164e-05f);
    pc = fmaf(pc, z, -0.0013888374669477344f);
    pc = fmaf(pc, z, 0.04166663810610771f);
    pc = fmaf(pc, z, -0.5f);
    float cr = fmaf(pc, z, 1.0f);
    if (((int)k) & 1) { sr = -sr; cr = -cr; }
    *s = sr;
    *c = cr;
}

/* ------------------------------------------------------------------------- */
/* Tap / coefficient design (host, once).                                       */
/* ------------------------------------------------------------------------- */

/* SDR++ core dsp/math/sinc.h: unnormalised sinc in double. */
static double sdr_sinc(double x) { return (x == 0.0) ? 1.0 : (sin(x) / x); }

/* SDR++ core PhaseControlLoop<float>::criticallyDamped;
 * called from FLL::init fll.cpp:23-24 and loop::PLL::init (pi4dqpsk.cpp:21). */
static void critically_damped(float bandwidth, float* alpha, float* beta) {
    float damp = (float)(sqrt(2.0) / 2.0);
    float denom = (float)(1.0 + 2.0 * (double)damp * (double)bandwidth + (double)(bandwidth * bandwidth));
    *alpha = (4 * damp * bandwidth) / denom;
    *beta = (4 * bandwidth * bandwidth) / denom;
}

/* SDR++ core taps::rootRaisedCosine<float>(count, beta, symbolrate, samplerate);
 * called at pi4dqpsk.cpp:18. */
static void design_rrc(int count, double beta, double symbolrate, double samplerate, float* taps) {
    double Ts = samplerate / symbolrate;
    double limit = Ts / (4.0 * beta);
    for (int i = 0; i < count; i++) {
        double t = (double)i - (double)count / 2.0 + 0.5;
        double v;
        if (t == 0.0) {
            v = (1.0 + beta * (4.0 / DB_M_PI - 1.0)) / Ts;
        } else if (t == limit || t == -limit) {
            v = ((1.0 + 2.0 / DB_M_PI) * sin(DB_M_PI / (4.0 * beta)) +
                 (1.0 - 2.0 / DB_M_PI) * cos(DB_M_PI / (4.0 * beta))) * beta / (Ts * sqrt(2.0));
        } else {
            double a = 4.0 * beta * t / Ts;
            v = ((sin((1.0 - beta) * DB_M_PI * t / Ts) + cos((1.0 + beta) * DB_M_PI * t / Ts) * a) /
                 ((1.0 - a * a) * DB_M_PI * t / Ts)) / Ts;
        }
        taps[i] = (float)v;
    }
}

/* FLL::createBandedgeFilters, fll.cpp:61-95.  Stores the conjugate tap pair as
 * a[] (common real part) and b[] (imaginary part of the LOWER filter; the upper
 * filter's is -b), already in the reversed FIR order of fll.cpp:92-93. */
static void design_bandedge(int filt_size, float filt_a, double symbolrate, double samplerate,
                            float* a, float* b) {
    float sps = (float)(samplerate / symbolrate);                 /* fll.cpp:62 */
    const int M = (int)(filt_size / sps);                         /* fll.cpp:64 */
    float power = 0;
    float bb[TETRA_ORACLE_MAX_TAPS];
    for (int i = 0; i < filt_size; i++) {                         /* fll.cpp:69-75 */
        float k = -M + i * 2.0f / sps;
        float tap = (float)(sdr_sinc(filt_a * k - 0.5f) + sdr_sinc(filt_a * k + 0.5f));
        power += tap;
        bb[i] = tap;
    }
    int N = (int)((filt_size - 1.0f) / 2.0f);                     /* fll.cpp:83 */
    for (int i = 0; i < filt_size; i++) {                         /* fll.cpp:84-94 */
        float tap = bb[i] / power;
        float k = (-N + (int)i) / (2.0f * sps);
        float arg = -2.0f * FL_M_PI * (1.0f + filt_a) * k;        /* lower: phasor(-2pi(1+a)k) */
        float re = cosf(arg) * tap;                               /* math::phasor = {cosf, sinf} */
        float im = sinf(arg) * tap;
        a[filt_size - i - 1] = re;
        b[filt_size - i - 1] = im;   /* upper filter: phasor(+...) = conj -> (re, -im) */
    }
}

/* COMPLEX_FD::generateInterpTaps complex_fd.cpp:153-158 =
 * windowedSinc<float>(P*T, hzToRads(0.5/P,1), nuttall, P) + buildPolyphaseBank(P)
 * (SDR++ core). */
static double win_nuttall(double n, double N) {
    static const double coefs[4] = { 0.355768, 0.487396, 0.144232, 0.012604 };
    double win = 0.0, sign = 1.0;
    for (int i = 0; i < 4; i++) {
        win += sign * coefs[i] * cos((double)i * 2.0 * DB_M_PI * n / N);
        sign = -sign;
    }
    return win;
}

static void design_interp_bank(float bank[TETRA_ORACLE_INTERP_PHASES][TETRA_ORACLE_INTERP_TAPS]) {
    const int P = TETRA_ORACLE_INTERP_PHASES, T = TETRA_ORACLE_INTERP_TAPS;
    const int count = P * T;
    double bw = 0.5 / (double)P;
    double omega = 2.0 * DB_M_PI * bw / 1.0;            /* math::hzToRads(bw, 1.0) */
    double half = (double)count / 2.0;
    double corr = (double)P * omega / DB_M_PI;          /* norm * omega / pi */
    for (int i = 0; i < count; i++) {
        double t = (double)i - half + 0.5;
        float tap = (float)(sdr_sinc(t * omega) * win_nuttall(t - half, (double)count) * corr);
        bank[(P - 1) - (i % P)][i / P] = tap;           /* buildPolyphaseBank */
    }
}

void tetra_oracle_default_cfg(tetra_oracle_cfg_t* cfg) {
    /* src/main.cpp:35-44 and the gain computation of src/main.cpp:78-82 (float arithmetic
     * with the double literal 2.0 in the denominator, as written there). */
    float bw = 0.00628f, damp = 0.707f;
    float denom = (float)(1.0f + 2.0 * damp * bw + bw * bw);
    float mu = (4.0f * damp * bw) / denom;
    float om = (4.0f * bw * bw) / denom;
    cfg->symbolrate = 18000;
    cfg->samplerate = 36000;
    cfg->rrc_tap_count = 65;
    cfg->rrc_beta = 0.35f;
    cfg->agc_rate = 0.02f;
    cfg->costas_bandwidth = 0.01f;
    cfg->fll_bandwidth = 0.006f;
    cfg->omega_gain = om;
    cfg->mu_gain = mu;
    cfg->omega_rel_limit = 0.02f;
}

int tetra_oracle_design(const tetra_oracle_cfg_t* cfg, tetra_oracle_tables_t* tab) {
    if (!cfg || !tab) return -1;
    if (cfg->rrc_tap_count < 2 || cfg->rrc_tap_count > TETRA_ORACLE_MAX_TAPS) return -2;
    if (!(cfg->symbolrate > 0) || !(cfg->samplerate > 0)) return -3;
    memset(tab, 0, sizeof(*tab));
    tab->cfg = *cfg;
    tab->ntaps = cfg->rrc_tap_count;
    tab->ntaps_be = cfg->rrc_tap_count;

    /* PI4DQPSK::init pi4dqpsk.cpp:17: FLL::init(NULL, fllBandwidth, (int)sym, (int)samp, taps,
     * (float)beta, 0, -pi/2, +pi/2); rates pass through int parameters (fll.h:33). */
    int sym_i = (int)cfg->symbolrate, samp_i = (int)cfg->samplerate;
    design_bandedge(tab->ntaps, (float)cfg->rrc_beta, (double)sym_i, (double)samp_i, tab->be_a, tab->be_b);
    float a_unused;
    critically_damped((float)cfg->fll_bandwidth, &a_unused, &tab->fll_beta);
    tab->fll_alpha = 0.0f;                                     /* fll.cpp:25 */
    tab->fll_min_freq = (float)(double)(-FL_M_PI / 2.0f);      /* pi4dqpsk.cpp:17 */
    tab->fll_max_freq = (float)(double)(FL_M_PI / 2.0f);

    design_rrc(tab->ntaps, cfg->rrc_beta, cfg->symbolrate, cfg->samplerate, tab->rrc); /* pi4dqpsk.cpp:18 */

    tab->agc_set_point = (float)1.0;                           /* pi4dqpsk.cpp:20 */
    tab->agc_max_gain = (float)10e6;
    tab->agc_rate = (float)cfg->agc_rate;

    critically_damped((float)cfg->costas_bandwidth, &tab->costas_alpha, &tab->costas_beta); /* pi4dqpsk.cpp:21 */
    tab->costas_min_freq = (float)(double)(-FL_M_PI / 10.0f);
    tab->costas_max_freq = (float)(double)(FL_M_PI / 10.0f);

    /* COMPLEX_FD::init complex_fd.cpp:12-28, called at pi4dqpsk.cpp:22 */
    double omega = cfg->samplerate / cfg->symbolrate;
    tab->tr_alpha = (float)cfg->mu_gain;
    tab->tr_beta = (float)cfg->omega_gain;
    tab->tr_omega = (float)omega;
    tab->tr_min_freq = (float)(omega * (1.0 - cfg->omega_rel_limit));
    tab->tr_max_freq = (float)(omega * (1.0 + cfg->omega_rel_limit));
    design_interp_bank(tab->bank);
    return 0;
}

static void design_timing_limits(tetra_oracle_tables_t* tab) {
    double omega = tab->cfg.samplerate / tab->cfg.symbolrate;
    tab->tr_omega = (float)omega;
    tab->tr_min_freq = (float)(omega * (1.0 - tab->cfg.omega_rel_limit));
    tab->tr_max_freq = (float)(omega * (1.0 + tab->cfg.omega_rel_limit));
}

int tetra_oracle_set_param(tetra_oracle_tables_t* tab, int param_id, double value, int quirks) {
    if (!tab) return -1;
    tetra_oracle_cfg_t* c = &tab->cfg;
    float unused;
    switch (param_id) {
    case 0: /* setSymbolrate pi4dqpsk.cpp:32-42 */
    case 1: /* setSamplerate pi4dqpsk.cpp:44-54 */
        if (!(value > 0)) return -3;
        if (param_id == 0) c->symbolrate = value; else c->samplerate = value;
        design_rrc(tab->ntaps, c->rrc_beta, c->symbolrate, c->samplerate, tab->rrc);
        design_timing_limits(tab);          /* recov.setOmega: complex_fd.cpp:30-41 */
        return 0;
    case 2: { /* setRRCTapCount -> setRRCParams pi4dqpsk.cpp:56-70 */
        int n = (int)value;
        if (n < 2 || n > TETRA_ORACLE_MAX_TAPS) return -2;
        c->rrc_tap_count = n;
        tab->ntaps = n;
        memset(tab->rrc, 0, sizeof(tab->rrc));
        design_rrc(n, c->rrc_beta, c->symbolrate, c->samplerate, tab->rrc);
        if (!quirks && n != tab->ntaps_be) {
            tab->ntaps_be = n;
            memset(tab->be_a, 0, sizeof(tab->be_a));
            memset(tab->be_b, 0, sizeof(tab->be_b));
            design_bandedge(n, (float)c->rrc_beta, (double)(int)c->symbolrate, (double)(int)c->samplerate, tab->be_a, tab->be_b);
        }
        return 0;
    }
    case 3: /* setRRCBeta(int) pi4dqpsk.cpp:72-74 */
        c->rrc_beta = quirks ? (double)(int)value : value;
        design_rrc(tab->ntaps, c->rrc_beta, c->symbolrate, c->samplerate, tab->rrc);
        return 0;
    case 4: c->agc_rate = value; tab->agc_rate = (float)value; return 0;                       /* pi4dqpsk.cpp:76-80 */
    case 5: c->costas_bandwidth = value;                                                      /* pi4dqpsk.cpp:82-86 */
        critically_damped((float)value, &tab->costas_alpha, &tab->costas_beta); return 0;
    case 6: c->fll_bandwidth = value; critically_damped((float)value, &unused, &tab->fll_beta); /* fll.cpp:63-72 */
        tab->fll_alpha = 0.0f; return 0;
    case 7: c->omega_gain = value; tab->tr_beta = (float)value; return 0;                      /* complex_fd.cpp:43-48 */
    case 8: c->mu_gain = value; tab->tr_alpha = (float)value; return 0;                        /* complex_fd.cpp:50-55 */
    case 9: c->omega_rel_limit = value; design_timing_limits(tab); return 0;                   /* complex_fd.cpp:57-62 */
    default: return -1;
    }
}

void tetra_oracle_reset_timing(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st) {
    st->mu = 0.0f;                  /* complex_fd.cpp:35-38 */
    st->omega = tab->tr_omega;
    st->offset = 0;
}

/* FIR::setTaps with MORE taps (SDR++ core, as restated in SURVEY.md Appendix A / tests/refshim): the RRC's delay line keeps
 * its old taps-1 samples and the newly visible older part is zero-filled.  Reference-quirks mode only; without the quirks
 * the single shared delay line simply becomes visible further back. */
void tetra_oracle_rrc_taps_grown(tetra_oracle_state_t* st, int old_ntaps) {
    if (st->rrc_valid > old_ntaps - 1) st->rrc_valid = old_ntaps - 1;
}

void tetra_oracle_reset_reference(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st) {
    st->agc_gain = 1.0f;            /* FastAGC::reset -> initGain */
    st->fll_phase = 0.0f;           /* FLL::reset fll.cpp:120-127 */
    st->fll_freq = 0.0f;
    st->rrc_valid = 0;              /* rrc.reset() pi4dqpsk.cpp:125: the RRC's delay line is cleared; the band-edge FIRs'
                                     * lines are NOT (FLL::reset touches only the loop) -- the shared line stays */
    st->costas_phase = 0.0f;        /* PLL::reset */
    st->costas_freq = 0.0f;
    tetra_oracle_reset_timing(tab, st);      /* COMPLEX_FD::reset complex_fd.cpp:78-87 */
}

void tetra_oracle_reset(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st) {
    memset(st, 0, sizeof(*st));
    st->agc_gain = 1.0f;            /* FastAGC initGain (SDR++ core) */
    st->fll_phase = 0.0f;           /* fll.cpp:26 */
    st->fll_freq = 0.0f;
    st->mu = 0.0f;                  /* complex_fd.cpp:22 */
    st->omega = tab->tr_omega;
    st->offset = 0;
    st->costas_phase = 0.0f;
    st->costas_freq = 0.0f;
    st->ph2 = 0.0f;
    st->prev = 0;
    st->rrc_valid = TETRA_ORACLE_MAX_TAPS - 1;
}

/* ------------------------------------------------------------------------- */
/* Run-time chain.                                                              */
/* ------------------------------------------------------------------------- */

/* SDR++ core complex_t::fastAmplitude */
static inline float fast_amplitude(float re, float im) {
    float r = fabsf(re), i = fabsf(im);
    return (r > i) ? (r + 0.4f * i) : (i + 0.4f * r);
}

/* Run-time phasor math::phasor(x) = { cosf(x), sinf(x) } (SDR++ core): the contract's polynomial, or -- reference-float
 * mode -- the host libm the reference itself calls. */
static inline __attribute__((always_inline)) void phasor_m(const int mode, float x, float* s, float* c) {
    if (mode == TETRA_ORACLE_REFERENCE_FLOATS) { *c = cosf(x); *s = sinf(x); }
    else tetra_oracle_sincosf(x, s, c);
}
/* One step of a dot product: the contract's fmaf chain, or -- reference-float mode -- the plain `acc += a * b` of a scalar
 * VOLK kernel (two roundings). */
static inline __attribute__((always_inline)) float mac_m(const int mode, float a, float b, float acc) {
    return mode == TETRA_ORACLE_REFERENCE_FLOATS ? acc + a * b : fmaf(a, b, acc);
}

static inline __attribute__((always_inline)) int process_impl(const int mode, const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st,
                         int count, const float* iq,
                         float* x_out, float* y_out, float* sym_out,
                         uint8_t* dibits, uint8_t* bits) {
    const int nt = tab->ntaps;                 /* RRC FIR length */
    const int nb = tab->ntaps_be;              /* band-edge FIR length (== nt unless a quirks-mode setRRCTapCount changed nt) */
    const int H = TETRA_ORACLE_MAX_TAPS - 1;   /* one delay line serves the three FIRs; the state always keeps the last
                                                * H = 128 FLL outputs (newest last), whatever the filter lengths */
    (void)nb;
    if (count <= 0) return 0;

    /* delay line + new samples, split re/im: w[0..H-1] = history, w[H+i] = x_i */
    float* wr = (float*)malloc(sizeof(float) * (size_t)(H + count) * 2);
    float* wi = wr + (H + count);
    float* yr = (float*)malloc(sizeof(float) * (size_t)(7 + count) * 2);
    float* yi = yr + (7 + count);
    for (int k = 0; k < H; k++) { wr[k] = st->hist[2 * k]; wi[k] = st->hist[2 * k + 1]; }

    /* --- FastAGC::process (SDR++ core; pi4dqpsk.cpp:134) fused sample-wise with
     * --- FLL::process fll.cpp:135-149 (no feedback between the stages). */
    float g = st->agc_gain;
    float ph = st->fll_phase, fr = st->fll_freq;
    const float pmax = FL_M_PI, pmin = -FL_M_PI, pdelta = pmax - pmin;
    for (int i = 0; i < count; i++) {
        /* AGC */
        float ar = iq[2 * i] * g, ai = iq[2 * i + 1] * g;
        float amp = sqrtf(ar * ar + ai * ai);
        g += (tab->agc_set_point - amp) * tab->agc_rate;
        if (g > tab->agc_max_gain) g = tab->agc_max_gain;
        /* FLL: x = in * phasor(-phase)  fll.cpp:137-138 */
        float s, c;
        phasor_m(mode, -ph, &s, &c);
        float xr = ar * c - ai * s;
        float xi = ai * c + ar * s;
        wr[H + i] = xr;
        wi[H + i] = xi;
        /* two band-edge FIRs over the same delay line, fll.cpp:141-142 */
        const float* pr = wr + i + (H - (nb - 1));
        const float* pi = wi + i + (H - (nb - 1));
        float lre, lim, hre, him;
        if (mode == TETRA_ORACLE_REFERENCE_FLOATS) {
            /* FIR<complex_t, complex_t>::process(1, ..) x 2 -> volk_32fc_x2_dot_prod_32fc in ascending tap order, each product
             * a full complex multiply added to the running sum: re += x.re t.re - x.im t.im, im += x.im t.re + x.re t.im.
             * Lower taps t = (a, b), upper taps t = (a, -b) (fll.cpp:89-93); y * (-b) == -(y * b) exactly. */
            lre = lim = hre = him = 0.0f;
            for (int k = 0; k < nb; k++) {
                const float a = tab->be_a[k], b = tab->be_b[k];
                lre += pr[k] * a - pi[k] * b;
                lim += pi[k] * a + pr[k] * b;
                hre += pr[k] * a - pi[k] * -b;
                him += pi[k] * a + pr[k] * -b;
            }
        } else {
            float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s4 = 0.0f;
            for (int k = 0; k < nb; k++) {
                s1 = fmaf(pr[k], tab->be_a[k], s1);
                s2 = fmaf(pi[k], tab->be_b[k], s2);
                s3 = fmaf(pr[k], tab->be_b[k], s3);
                s4 = fmaf(pi[k], tab->be_a[k], s4);
            }
            lre = s1 - s2; lim = s4 + s3;     /* x * (a + jb) */
            hre = s1 + s2; him = s4 - s3;     /* x * (a - jb) */
        }
        float err = fast_amplitude(hre, him) - fast_amplitude(lre, lim); /* fll.cpp:143 */
        /* pcl.advance(err) fll.cpp:145 (PhaseControlLoop<float>, alpha forced 0) */
        fr += tab->fll_beta * err;
        if (fr > tab->fll_max_freq) fr = tab->fll_max_freq;
        else if (fr < tab->fll_min_freq) fr = tab->fll_min_freq;
        ph += fr + tab->fll_alpha * err;
        while (ph > pmax) ph -= pdelta;
        while (ph < pmin) ph += pdelta;
    }
    st->agc_gain = g;
    st->fll_phase = ph;
    st->fll_freq = fr;
    if (x_out) for (int i = 0; i < count; i++) { x_out[2 * i] = wr[H + i]; x_out[2 * i + 1] = wi[H + i]; }

    /* --- FIR<complex_t,float>::process (SDR++ core; pi4dqpsk.cpp:136): RRC --- */
    for (int k = 0; k < 7; k++) { yr[k] = st->ybuf[2 * k]; yi[k] = st->ybuf[2 * k + 1]; }
    const int valid0 = st->rrc_valid;          /* delay-line samples the RRC may see (all of them unless a reference-style
                                                * reset or a tap-count growth happened less than nt-1 samples ago) */
    for (int i = 0; i < count; i++) {
        float ar = 0.0f, ai = 0.0f;
        const float* pr = wr + i + (H - (nt - 1));
        const float* pi = wi + i + (H - (nt - 1));
        const long long have = (long long)valid0 + i + 1;   /* window samples that exist for this filter, newest first */
        const int k0 = have >= nt ? 0 : (int)(nt - have);   /* older ones are zeros in the reference's RRC delay line;
                                                             * fmaf(0, tap, acc) == acc, so they are skipped */
        for (int k = k0; k < nt; k++) {
            ar = mac_m(mode, pr[k], tab->rrc[k], ar);
            ai = mac_m(mode, pi[k], tab->rrc[k], ai);
        }
        yr[7 + i] = ar;
        yi[7 + i] = ai;
    }
    if (y_out) for (int i = 0; i < count; i++) { y_out[2 * i] = yr[7 + i]; y_out[2 * i + 1] = yi[7 + i]; }
    /* FIR delay line update (memmove in SDR++ core FIR::process) */
    for (int k = 0; k < H; k++) { st->hist[2 * k] = wr[count + k]; st->hist[2 * k + 1] = wi[count + k]; }
    st->rrc_valid = (long long)valid0 + count >= H ? H : valid0 + count;

    /* --- COMPLEX_FD::process complex_fd.cpp:89-151, PI4DQPSK_COSTAS::process
     * --- pi4dqpsk_costas.cpp:5-21, DQPSKSymbolExtractor::process dqpsk_sym_extr.cpp:4-55
     * --- (decisions only; the sync statistic is not on the bit path),
     * --- BitUnpacker::process bit_unpacker.cpp:4-10; fused symbol-wise. */
    float mu = st->mu, om = st->omega;
    int offset = st->offset;
    float cph = st->costas_phase, cfr = st->costas_freq, ph2 = st->ph2;
    uint8_t prev = st->prev;
    int S = 0;
    while (offset < count) {
        /* complex_fd.cpp:101-103 */
        int phase = (int)floorf(mu * (float)TETRA_ORACLE_INTERP_PHASES);
        if (phase < 0) phase = 0;
        if (phase > TETRA_ORACLE_INTERP_PHASES - 1) phase = TETRA_ORACLE_INTERP_PHASES - 1;
        const float* br = yr + offset;
        const float* bi = yi + offset;
        const float* tp = tab->bank[phase];
        float vr = 0.0f, vi = 0.0f;
        for (int k = 0; k < 8; k++) { vr = mac_m(mode, br[k], tp[k], vr); vi = mac_m(mode, bi[k], tp[k], vi); }
        /* derivative, complex_fd.cpp:107-123 (_outSps == 1: every output is a symbol) */
        float dr, di;
        if (phase == 0) {
            const float* t1 = tab->bank[phase + 1];
            float fr1 = 0.0f, fi1 = 0.0f;
            for (int k = 0; k < 8; k++) { fr1 = mac_m(mode, br[k], t1[k], fr1); fi1 = mac_m(mode, bi[k], t1[k], fi1); }
            dr = fr1 - vr; di = fi1 - vi;
        } else if (phase == TETRA_ORACLE_INTERP_PHASES - 1) {
            const float* t0 = tab->bank[phase - 1];
            float fr0 = 0.0f, fi0 = 0.0f;
            for (int k = 0; k < 8; k++) { fr0 = mac_m(mode, br[k], t0[k], fr0); fi0 = mac_m(mode, bi[k], t0[k], fi0); }
            dr = vr - fr0; di = vi - fi0;
        } else {
            const float* t1 = tab->bank[phase + 1];
            const float* t0 = tab->bank[phase - 1];
            float fr1 = 0.0f, fi1 = 0.0f, fr0 = 0.0f, fi0 = 0.0f;
            for (int k = 0; k < 8; k++) {
                fr1 = mac_m(mode, br[k], t1[k], fr1); fi1 = mac_m(mode, bi[k], t1[k], fi1);
                fr0 = mac_m(mode, br[k], t0[k], fr0); fi0 = mac_m(mode, bi[k], t0[k], fi0);
            }
            dr = (fr1 - fr0) * 0.5f; di = (fi1 - fi0) * 0.5f;
        }
        /* complex_fd.cpp:126, 136-137 */
        float terr = ((vr > 0 ? 1.0f : -1.0f) * dr) + ((vi > 0 ? 1.0f : -1.0f) * di);
        if (terr > 1.0f) terr = 1.0f;
        if (terr < -1.0f) terr = -1.0f;
        /* pcl.advance (PhaseControlLoop<float,false>) complex_fd.cpp:140-143 */
        om += tab->tr_beta * terr;
        if (om > tab->tr_max_freq) om = tab->tr_max_freq;
        else if (om < tab->tr_min_freq) om = tab->tr_min_freq;
        mu += om + tab->tr_alpha * terr;
        float delta = floorf(mu);
        offset += (int)delta;
        mu -= delta;

        /* Costas, pi4dqpsk_costas.cpp:7-19 */
        float s, c;
        phasor_m(mode, -cph, &s, &c);
        float xr = vr * c - vi * s;
        float xi = vi * c + vr * s;
        ph2 += -FL_M_PI / 4.0f;
        if (ph2 >= 2 * FL_M_PI) ph2 -= 2 * FL_M_PI;
        else if (ph2 <= -2 * FL_M_PI) ph2 += 2 * FL_M_PI;
        phasor_m(mode, ph2, &s, &c);
        float zr = xr * c - xi * s;
        float zi = xi * c + xr * s;
        /* errorFunction pi4dqpsk_costas.cpp:23-28 (math::step: x>0 ? 1 : -1) */
        float cerr = ((zr > 0 ? 1.0f : -1.0f) * zi) - ((zi > 0 ? 1.0f : -1.0f) * zr);
        if (cerr < -1.0f) cerr = -1.0f;
        if (cerr > 1.0f) cerr = 1.0f;
        cfr += tab->costas_beta * cerr;
        if (cfr > tab->costas_max_freq) cfr = tab->costas_max_freq;
        else if (cfr < tab->costas_min_freq) cfr = tab->costas_min_freq;
        cph += cfr + tab->costas_alpha * cerr;
        while (cph > pmax) cph -= pdelta;
        while (cph < pmin) cph += pdelta;
        if (sym_out) { sym_out[2 * S] = zr; sym_out[2 * S + 1] = zi; }

        /* slicer + differential decoder, dqpsk_sym_extr.cpp:6-7, 32-52 */
        int a = zi < 0, b = zr < 0;
        {
            /* sync/quality statistic, dqpsk_sym_extr.cpp:8-31 (complex_t::phase() = atan2f(im, re)) */
            float ideal_re = b ? -0.7071f : 0.7071f, ideal_im = a ? -0.7071f : 0.7071f;
            float dist = fabsf(atan2f(ideal_im, ideal_re) - atan2f(zi, zr));
            st->errorbuf[st->errorptr] = dist;
            st->errorptr++;
            if (st->errorptr >= 4096) st->errorptr = 0;
            st->errordisplayptr++;
            if (st->errordisplayptr >= 256) {
                float xerr = 0;
                for (int q = 0; q < 4096; q++) xerr += st->errorbuf[q];
                xerr /= (float)4096;
                st->standarderr = xerr;
                st->sync = (xerr >= 0.35f) ? 0 : 1;
                st->errordisplayptr = 0;
            }
        }
        uint8_t sym = (uint8_t)((a << 1) | (a != b));
        uint8_t pd = (uint8_t)((sym - prev + 4) % 4);
        static const uint8_t remap[4] = { 0, 1, 3, 2 };
        uint8_t d = remap[pd];
        prev = sym;
        if (dibits) dibits[S] = d;
        /* bit_unpacker.cpp:6-7 */
        if (bits) { bits[2 * S] = (uint8_t)((d & 2) >> 1); bits[2 * S + 1] = (uint8_t)(d & 1); }
        S++;
    }
    offset -= count;                                            /* complex_fd.cpp:145 */
    for (int k = 0; k < 7; k++) { st->ybuf[2 * k] = yr[count + k]; st->ybuf[2 * k + 1] = yi[count + k]; } /* :148 */
    st->mu = mu; st->omega = om; st->offset = offset;
    st->costas_phase = cph; st->costas_freq = cfr; st->ph2 = ph2; st->prev = prev;

    free(wr);
    free(yr);
    return S;
}

int tetra_oracle_process(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st, int count, const float* iq,
                         float* x_out, float* y_out, float* sym_out, uint8_t* dibits, uint8_t* bits) {
    return process_impl(TETRA_ORACLE_CONTRACT, tab, st, count, iq, x_out, y_out, sym_out, dibits, bits);
}

int tetra_oracle_process_mode(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st, int mode, int count, const float* iq,
                              float* x_out, float* y_out, float* sym_out, uint8_t* dibits, uint8_t* bits) {
    if (mode == TETRA_ORACLE_REFERENCE_FLOATS)
        return process_impl(TETRA_ORACLE_REFERENCE_FLOATS, tab, st, count, iq, x_out, y_out, sym_out, dibits, bits);
    if (mode != TETRA_ORACLE_CONTRACT) return -1;
    return process_impl(TETRA_ORACLE_CONTRACT, tab, st, count, iq, x_out, y_out, sym_out, dibits, bits);
}

int tetra_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int tetra_oracle_process_batch(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* states,
                               int n_channels, int n_samples, int chunk, int threads,
                               const float* iq, uint8_t* bits, int bits_stride,
                               int32_t* n_bits, float* sym) {
    if (chunk <= 0 || chunk > n_samples) chunk = n_samples;
    int rc = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int c = 0; c < n_channels; c++) {
        const float* in = iq + (size_t)c * (size_t)n_samples * 2;
        uint8_t* bo = bits + (size_t)c * (size_t)bits_stride;
        float* so = sym ? sym + (size_t)c * (size_t)(bits_stride / 2) * 2 : NULL;
        int nb = 0;
        /* symbols one call can emit: every symbol moves mu by at least tr_min_freq - |tr_alpha| (complex_fd.cpp:136-143) */
        double step = (double)tab->tr_min_freq - fabs((double)tab->tr_alpha);
        if (!(step > 0.01)) step = 0.01;
        if (step > 1.0) step = 1.0;
        int maxs = (int)((double)(chunk + 1) / step) + 16;
        uint8_t* tb = (uint8_t*)malloc((size_t)maxs * 2);
        float* ts = (float*)malloc(sizeof(float) * (size_t)maxs * 2);
        for (int pos = 0; pos < n_samples; pos += chunk) {
            int cnt = (n_samples - pos < chunk) ? (n_samples - pos) : chunk;
            int S = tetra_oracle_process(tab, &states[c], cnt, in + (size_t)pos * 2, NULL, NULL, ts, NULL, tb);
            if (nb + 2 * S > bits_stride) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                rc = -1;
                break;
            }
            memcpy(bo + nb, tb, (size_t)2 * S);
            if (so) memcpy(so + nb, ts, sizeof(float) * (size_t)2 * S);
            nb += 2 * S;
        }
        n_bits[c] = nb;
        free(tb);
        free(ts);
    }
    return rc;
}

/* d[i][j] = the fmaf chain of the arithmetic contract over ascending k from +0 (test helper for the GPU's matrix-pipe
 * self-test: every FIR sum of the chain is such a chain). a [m][k], b [k][n], d [m][n], row-major. */
void tetra_oracle_fmaf_chain_matmul(const float* a, const float* b, int m, int n, int k, float* d) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            float acc = 0.0f;
            for (int q = 0; q < k; q++) acc = fmaf(a[(size_t)i * k + q], b[(size_t)q * n + j], acc);
            d[(size_t)i * n + j] = acc;
        }
}
