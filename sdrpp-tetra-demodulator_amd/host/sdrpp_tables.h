// sdrpp_tables.h -- the demodulator's three tables from SDR++'s OWN generators (TETRA_WITH_SDRPP builds only).
//
// The reference designs its filters at run time with SDR++ core code that is neither vendored nor pinned by the reference
// repository:
//   RRC taps             dsp::taps::rootRaisedCosine<float>(count, beta, symbolrate, samplerate)      src/dsp/pi4dqpsk.cpp:18,38,50,63
//   FLL band-edge taps   FLL::createBandedgeFilters through dsp::math::sinc / dsp::math::phasor          src/dsp/fll.cpp:61-95
//   interpolator bank    dsp::taps::windowedSinc<float>(.., dsp::math::hzToRads(..), dsp::window::nuttall, ..)
//                        + dsp::multirate::buildPolyphaseBank<float>                                    src/dsp/complex_fd.cpp:153-158
// The HIP library carries a restatement of those formulas (csrc/design.hpp) for builds outside SDR++.  INSIDE an SDR++ build the
// real headers are on the include path, so the block calls THEM and hands the results to the kernels as caller tables
// (tetra_demod_config_t.rrc_taps / bandedge_taps / interp_bank, tetra_demod_set_tables): whatever the installed SDR++ computes --
// pi spelled as a float or a double, another window, a later fix -- is what the GPU runs, bit for bit, exactly like the
// reference built against the same tree.  Nothing here re-implements a core primitive; the band-edge design is the one piece of
// the reference's OWN arithmetic on this route (fll.cpp is the plugin's file, not SDR++'s) and follows it operation by operation
// on the included headers' math::sinc, math::phasor, complex_t::operator* and FL_M_PI.
#pragma once
#ifdef TETRA_WITH_SDRPP
// the include set of src/dsp/pi4dqpsk.h:2-19 (window::nuttall arrives through it, as it does for complex_fd.cpp:155)
#include <dsp/processor.h>
#include <dsp/loop/phase_control_loop.h>
#include <dsp/taps/windowed_sinc.h>
#include <dsp/multirate/polyphase_bank.h>
#include <dsp/math/step.h>
#include <dsp/loop/costas.h>
#include <dsp/clock_recovery/mm.h>
#include <dsp/taps/root_raised_cosine.h>
#include <dsp/filter/fir.h>
#include <dsp/loop/fast_agc.h>
#include <math.h>

#include <vector>

namespace dsp {
namespace demod {
namespace sdrpp_tables {

// pi4dqpsk.cpp:18 (and :38, :50, :63)
inline std::vector<float> rrc(int rrcTapCount, double rrcBeta, double symbolrate, double samplerate) {
    // (only members the reference itself touches are relied on: tap<T>::taps, fll.cpp:92; PolyphaseBank<T>::phases, complex_fd.cpp:102)
    dsp::tap<float> t = dsp::taps::rootRaisedCosine<float>(rrcTapCount, rrcBeta, symbolrate, samplerate);
    std::vector<float> out(t.taps, t.taps + rrcTapCount);
    dsp::taps::free(t);
    return out;
}

// fll.cpp:61-95 with the arguments of pi4dqpsk.cpp:17 -> FLL::init (fll.cpp:10-15: the rates pass through int parameters into
// double members, the roll-off through a float).  Returns [2][filt_size]: re, im of the LOWER band-edge filter; the upper one is
// built from the negated phasor argument (fll.cpp:90) = its conjugate.
inline std::vector<float> bandedge(int filt_size, float filt_a, int sym_rate, int samp_rate) {
    const double _symbolrate = sym_rate, _samplerate = samp_rate;
    const int _filt_size = filt_size;
    const float _filt_a = filt_a;
    float sps = _samplerate / _symbolrate;
    const int M = (_filt_size / sps);
    float power = 0;
    std::vector<float> bb_taps;
    for (int i = 0; i < _filt_size; i++) {
        float k = -M + i * 2.0f / sps;
        float tap = dsp::math::sinc(_filt_a * k - 0.5f) + dsp::math::sinc(_filt_a * k + 0.5f);
        power += tap;
        bb_taps.push_back(tap);
    }
    std::vector<float> out(2 * (size_t)_filt_size);
    int N = (bb_taps.size() - 1.0f) / 2.0f;
    for (int i = 0; i < _filt_size; i++) {
        float tap = bb_taps[i] / power;
        float k = (-N + (int)i) / (2.0f * sps);
        dsp::complex_t t1 = dsp::math::phasor(-2.0f * FL_M_PI * (1.0f + _filt_a) * k) * tap;
        out[(size_t)(_filt_size - i - 1)] = t1.re;
        out[(size_t)_filt_size + (size_t)(_filt_size - i - 1)] = t1.im;
    }
    return out;
}

// complex_fd.cpp:153-158 with COMPLEX_FD::init's defaults (complex_fd.h: interpPhaseCount 128, interpTapCount 8).
// Returns [phases][tapsPerPhase] row-major, the layout of the kernels' bank and of PolyphaseBank::phases[p][k].
inline std::vector<float> interpBank(int interpPhaseCount = 128, int interpTapCount = 8) {
    double bw = 0.5 / (double)interpPhaseCount;
    dsp::tap<float> lp = dsp::taps::windowedSinc<float>(interpPhaseCount * interpTapCount, dsp::math::hzToRads(bw, 1.0), dsp::window::nuttall,
                                                        interpPhaseCount);
    dsp::multirate::PolyphaseBank<float> bank = dsp::multirate::buildPolyphaseBank<float>(interpPhaseCount, lp);
    // read the way COMPLEX_FD::process does: interpBank.phases[phase] holds _interpTapCount taps (complex_fd.cpp:102)
    std::vector<float> out((size_t)interpPhaseCount * (size_t)interpTapCount);
    for (int p = 0; p < interpPhaseCount; p++)
        for (int k = 0; k < interpTapCount; k++) out[(size_t)p * (size_t)interpTapCount + (size_t)k] = bank.phases[p][k];
    dsp::multirate::freePolyphaseBank(bank);
    dsp::taps::free(lp);
    return out;
}

}  // namespace sdrpp_tables
}  // namespace demod
}  // namespace dsp
#endif  // TETRA_WITH_SDRPP
