// design.hpp -- host-side (init-time) filter and loop design for the batched demodulator.
//
// Mirrors what the reference computes once in PI4DQPSK::init (src/dsp/pi4dqpsk.cpp:11-30):
//   FLL band-edge taps        FLL::createBandedgeFilters, src/dsp/fll.cpp:61-95
//   FLL / Costas loop gains   SDR++ core PhaseControlLoop::criticallyDamped via fll.cpp:23-26 and loop::PLL::init
//   RRC taps                  SDR++ core taps::rootRaisedCosine<float>, called at pi4dqpsk.cpp:18
//   interpolator bank         COMPLEX_FD::generateInterpTaps, src/dsp/complex_fd.cpp:153-158
//                             (SDR++ core windowedSinc + window::nuttall + buildPolyphaseBank)
//   timing loop limits        COMPLEX_FD::init, src/dsp/complex_fd.cpp:12-28
// SDR++ core is not vendored by the reference; its formulas are the ones recorded in SURVEY.md
// Appendix A.  Init-time math uses the host libm exactly like the reference does.
#pragma once

#include <cmath>
#include <cstring>
#include <vector>

#include "demod_core.hpp"

namespace tdm {
namespace host {

constexpr double kDbPi = 3.14159265358979323846;

struct DesignParams {
    double symbolrate = 18000, samplerate = 36000;
    int rrc_tap_count = 65;
    double rrc_beta = 0.35f, agc_rate = 0.02f, costas_bandwidth = 0.01f, fll_bandwidth = 0.006f;
    double omega_gain = 0, mu_gain = 0, omega_rel_limit = 0.02f;
};

struct Design {
    int ntaps = 0;                          // RRC FIR length
    int ntaps_be = 0;                       // band-edge FIR length (== ntaps unless TETRA_FLAG_REFERENCE_QUIRKS kept the FLL's
                                            // construction-time filters across a setRRCTapCount, pi4dqpsk.cpp:56-70)
    std::vector<float> rrc, be_re, be_im;   // [ntaps] / [ntaps_be]; be_* = lower band-edge filter, upper = conj
    std::vector<float> bank;                // [128*8]
    K1Consts k1;
    K2Consts k2;
    float tr_omega = 2.0f;
};

inline double sinc_unnorm(double x) { return (x == 0.0) ? 1.0 : (std::sin(x) / x); }

inline void loop_gains(float bandwidth, float& alpha, float& beta) {
    float damp = (float)(std::sqrt(2.0) / 2.0);
    float denom = (float)(1.0 + 2.0 * (double)damp * (double)bandwidth + (double)(bandwidth * bandwidth));
    alpha = (4 * damp * bandwidth) / denom;
    beta = (4 * bandwidth * bandwidth) / denom;
}

// The plugin's timing-loop gains, src/main.cpp:78-82.
inline void default_timing_gains(double& omega_gain, double& mu_gain) {
    float bw = 0.00628f, damp = 0.707f;
    float denom = (float)(1.0f + 2.0 * damp * bw + bw * bw);
    mu_gain = (4.0f * damp * bw) / denom;
    omega_gain = (4.0f * bw * bw) / denom;
}

inline void root_raised_cosine(int count, double beta, double symbolrate, double samplerate, float* taps) {
    const double Ts = samplerate / symbolrate;
    const double limit = Ts / (4.0 * beta);
    for (int i = 0; i < count; i++) {
        const double t = (double)i - (double)count / 2.0 + 0.5;
        double v;
        if (t == 0.0) {
            v = (1.0 + beta * (4.0 / kDbPi - 1.0)) / Ts;
        } else if (t == limit || t == -limit) {
            v = ((1.0 + 2.0 / kDbPi) * std::sin(kDbPi / (4.0 * beta)) +
                 (1.0 - 2.0 / kDbPi) * std::cos(kDbPi / (4.0 * beta))) * beta / (Ts * std::sqrt(2.0));
        } else {
            const double a = 4.0 * beta * t / Ts;
            v = ((std::sin((1.0 - beta) * kDbPi * t / Ts) + std::cos((1.0 + beta) * kDbPi * t / Ts) * a) /
                 ((1.0 - a * a) * kDbPi * t / Ts)) / Ts;
        }
        taps[i] = (float)v;
    }
}

inline void bandedge_filters(int filt_size, float filt_a, double symbolrate, double samplerate, float* re, float* im) {
    const float sps = (float)(samplerate / symbolrate);
    const int M = (int)(filt_size / sps);
    float power = 0;
    std::vector<float> bb(filt_size);
    for (int i = 0; i < filt_size; i++) {
        float k = -M + i * 2.0f / sps;
        float tap = (float)(sinc_unnorm(filt_a * k - 0.5f) + sinc_unnorm(filt_a * k + 0.5f));
        power += tap;
        bb[i] = tap;
    }
    const int N = (int)((filt_size - 1.0f) / 2.0f);
    for (int i = 0; i < filt_size; i++) {
        float tap = bb[i] / power;
        float k = (-N + (int)i) / (2.0f * sps);
        float arg = -2.0f * kFlPi * (1.0f + filt_a) * k;
        re[filt_size - i - 1] = cosf(arg) * tap;
        im[filt_size - i - 1] = sinf(arg) * tap;
    }
}

inline double nuttall(double n, double N) {
    static const double coefs[4] = { 0.355768, 0.487396, 0.144232, 0.012604 };
    double win = 0.0, sign = 1.0;
    for (int i = 0; i < 4; i++) {
        win += sign * coefs[i] * std::cos((double)i * 2.0 * kDbPi * n / N);
        sign = -sign;
    }
    return win;
}

inline void interp_bank(float* bank /* [128][8] */) {
    const int P = kInterpPhases, T = kInterpTaps, count = P * T;
    const double bw = 0.5 / (double)P;
    const double omega = 2.0 * kDbPi * bw / 1.0;
    const double half = (double)count / 2.0;
    const double corr = (double)P * omega / kDbPi;
    for (int i = 0; i < count; i++) {
        const double t = (double)i - half + 0.5;
        const float tap = (float)(sinc_unnorm(t * omega) * nuttall(t - half, (double)count) * corr);
        bank[((P - 1) - (i % P)) * T + (i / P)] = tap;
    }
}

// The pieces of the design, one per group of PI4DQPSK setters (src/dsp/pi4dqpsk.cpp:32-118): the loop setters change
// loop constants only, the rate / RRC setters re-design only the RRC taps (and the timing loop's nominal omega and limits);
// nothing but init ever designs the FLL's band-edge filters.
inline void design_loops(const DesignParams& p, Design& d) {
    float unused;
    loop_gains((float)p.fll_bandwidth, unused, d.k1.fll_beta);
    d.k1.fll_alpha = 0.0f;  // fll.cpp:25
    d.k1.fll_min_freq = (float)(double)(-kFlPi / 2.0f);
    d.k1.fll_max_freq = (float)(double)(kFlPi / 2.0f);
    d.k1.agc_set_point = (float)1.0;
    d.k1.agc_max_gain = (float)10e6;
    d.k1.agc_rate = (float)p.agc_rate;
    loop_gains((float)p.costas_bandwidth, d.k2.costas_alpha, d.k2.costas_beta);
    d.k2.costas_min_freq = (float)(double)(-kFlPi / 10.0f);
    d.k2.costas_max_freq = (float)(double)(kFlPi / 10.0f);
    d.k2.tr_alpha = (float)p.mu_gain;
    d.k2.tr_beta = (float)p.omega_gain;
}
inline void design_timing_limits(const DesignParams& p, Design& d) {
    const double omega = p.samplerate / p.symbolrate;
    d.tr_omega = (float)omega;
    d.k2.tr_min_freq = (float)(omega * (1.0 - p.omega_rel_limit));
    d.k2.tr_max_freq = (float)(omega * (1.0 + p.omega_rel_limit));
}
inline void design_rrc(const DesignParams& p, Design& d) {
    d.ntaps = p.rrc_tap_count;
    d.rrc.assign(d.ntaps, 0.f);
    root_raised_cosine(d.ntaps, p.rrc_beta, p.symbolrate, p.samplerate, d.rrc.data());
}
inline void design_bandedge(const DesignParams& p, Design& d, int count) {
    d.ntaps_be = count;
    d.be_re.assign(count, 0.f);
    d.be_im.assign(count, 0.f);
    // FLL::init takes the rates through int parameters (src/dsp/fll.h:33)
    bandedge_filters(count, (float)p.rrc_beta, (double)(int)p.symbolrate, (double)(int)p.samplerate, d.be_re.data(), d.be_im.data());
}
// What the kernels cover.  The timing loop moves floor(mu) samples per symbol with mu = frac + freq + alpha * err,
// freq >= omega (1 - rel_limit), |err| <= 1 (complex_fd.cpp:136-143): min_step = omega (1 - rel_limit) - |mu_gain| samples per
// symbol at least.  While min_step >= 1 every symbol advances by at least one sample (the kernel's forward-progress clamp is
// then neutral).  Below that the reference emits several symbols from one offset (floor(mu) = 0, complex_fd.cpp:141-143): the
// kernels' "deep" variant (kernel_fused.hpp: DEEP) does the same, with a symbol ring sized for min_step >= kMinStepDeep; below
// that (more than 3.7 symbols per sample) its second level (4-channel workgroups, ring sized for min_step >= kMinStepDeeper), and
// below THAT the generic kernel (kernel_generic.hpp: one lane per channel, HBM scratch instead of LDS rings); filters of 73 .. 129
// taps take the fused kernel's long rows (needs_long below).  Refused: min_step <= 0 -- the reference's own loop
// may then never leave process() or walk backwards out of its buffer -- and more than kMaxTaps taps (the delay line this
// library and its checker keep).  Output rows are sized from min_step (tetra_demod_bits_stride_for), so any accepted parameter
// set fits its rows.
constexpr double kMinStepDeep = 0.27;
constexpr double kMinStepDeeper = 0.07;        // the 4-channel shape's second ring level (kernel_fused.hpp: kFSDeeper)
constexpr int kMaxTaps = 129;          // = kernel_generic.hpp kGenMaxTaps = TETRA_DEMOD_MAX_TAPS = the oracle's TETRA_ORACLE_MAX_TAPS
inline bool params_ok(const DesignParams& p) {
    if (p.rrc_tap_count < 2 || p.rrc_tap_count > kMaxTaps) return false;
    if (!(p.symbolrate > 0) || !(p.samplerate > 0)) return false;
    if (!(p.omega_rel_limit >= 0.0) || !(p.omega_rel_limit < 1.0)) return false;
    const float omega_min = (float)(p.samplerate / p.symbolrate * (1.0 - p.omega_rel_limit));
    if (!((double)omega_min - std::fabs((double)(float)p.mu_gain) > 0.0)) return false;
    return true;
}

// Output row length (bytes = bits) that holds any call of n samples under design d (tetra_demod_bits_stride_for).
// Smallest advance of the timing loop per symbol, in samples: every step adds freq + alpha * err to mu with
// freq >= omega (1 - rel_limit) and |err| <= 1 (complex_fd.cpp:136-143), and floor(mu) of it moves the offset.
inline double min_step(const Design& d) { return (double)d.k2.tr_min_freq - std::fabs((double)d.k2.tr_alpha); }
// several symbols may share an offset: the launch takes the kernels' DEEP variant
inline bool needs_deep(const Design& d) { return min_step(d) < 1.0; }
// beyond the fused kernel's symbol ring: the generic kernel (kernel_generic.hpp)
inline bool needs_generic(const Design& d) { return min_step(d) < kMinStepDeeper; }
// 0: every symbol advances at least one sample; 1: several symbols may share an offset (deep symbol ring, 16- and 4-channel shapes);
// 2: more than 3.7 of them per sample (the 4-channel shape's 1024-deep ring)
inline int deep_level(const Design& d) { return min_step(d) < kMinStepDeep ? 2 : min_step(d) < 1.0 ? 1 : 0; }
// filters beyond the 72 taps of the fused kernel's regular rows: its LONG variant (FLL rows of 16 x 9 taps in 4-channel workgroups,
// 8 x 17 in 16-channel ones; 128 delay-line samples) -- or, on request (TETRA_FLAG_GENERIC_KERNEL), the generic kernel
inline bool needs_long(const Design& d) { return !needs_generic(d) && (d.ntaps > kF8Pad || d.ntaps_be > kF8Pad); }
inline long long bits_stride_for(const Design& d, long long n) {
    // K symbols are emitted only while (K - 1) min_step - 1 < n (the offsets of a call start at >= 0 and the fractional
    // parts of mu telescope to less than one sample):  K <= (n + 1) / min_step + 1; two symbols of margin for the float
    // rounding of the loop, rows a multiple of 16 bytes
    const long long k = (long long)std::floor((double)(n + 1) / (min_step(d) * (1.0 - 1e-6))) + 3;
    return (2 * k + 15) / 16 * 16;
}

// Full design (PI4DQPSK::init).  user_* may be null.  Returns false on unsupported parameters.
inline bool make_design(const DesignParams& p, const float* user_rrc, const float* user_be, const float* user_bank,
                        Design& d) {
    if (!params_ok(p)) return false;
    design_rrc(p, d);
    design_bandedge(p, d, p.rrc_tap_count);
    d.bank.assign(kInterpPhases * kInterpTaps, 0.f);
    if (user_be) {
        std::memcpy(d.be_re.data(), user_be, sizeof(float) * d.ntaps);
        std::memcpy(d.be_im.data(), user_be + d.ntaps, sizeof(float) * d.ntaps);
    }
    if (user_rrc) std::memcpy(d.rrc.data(), user_rrc, sizeof(float) * d.ntaps);
    if (user_bank) std::memcpy(d.bank.data(), user_bank, sizeof(float) * d.bank.size());
    else interp_bank(d.bank.data());
    design_loops(p, d);
    design_timing_limits(p, d);
    return true;
}

}  // namespace host
}  // namespace tdm
