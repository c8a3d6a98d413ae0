// Stand-in for SDR++ core's dsp/loop/phase_control_loop.h -- see ../../README.md.
#pragma once
#include <dsp/processor.h>

namespace dsp {
    namespace loop {
        template <class T, bool CLAMP_PHASE = true> class PhaseControlLoop {
        public:
            PhaseControlLoop() {}
            void init(T alpha, T beta, T phase_, T minPhase, T maxPhase, T freq_, T minFreq, T maxFreq) {
                _alpha = alpha; _beta = beta;
                phase = phase_; freq = freq_;
                _minPhase = minPhase; _maxPhase = maxPhase; _phaseDelta = maxPhase - minPhase;
                _minFreq = minFreq; _maxFreq = maxFreq;
            }
            static void criticallyDamped(T bandwidth, T& alpha, T& beta) {
                const T damping = sqrt(2.0) / 2.0;
                const T denom = (1.0 + 2.0 * damping * bandwidth + bandwidth * bandwidth);
                alpha = (4 * damping * bandwidth) / denom;
                beta = (4 * bandwidth * bandwidth) / denom;
            }
            void setCoefficients(T alpha, T beta) { _alpha = alpha; _beta = beta; }
            void setPhaseLimits(T minPhase, T maxPhase) { _minPhase = minPhase; _maxPhase = maxPhase; _phaseDelta = maxPhase - minPhase; }
            void setFreqLimits(T minFreq, T maxFreq) { _minFreq = minFreq; _maxFreq = maxFreq; }
            inline void advance(T error) {
                freq += _beta * error;
                if (freq > _maxFreq) { freq = _maxFreq; }
                else if (freq < _minFreq) { freq = _minFreq; }
                phase += freq + (_alpha * error);
                if (CLAMP_PHASE) {
                    while (phase > _maxPhase) { phase -= _phaseDelta; }
                    while (phase < _minPhase) { phase += _phaseDelta; }
                }
            }
            T freq = 0;
            T phase = 0;
        protected:
            T _alpha = 0, _beta = 0;
            T _minPhase = 0, _maxPhase = 0, _phaseDelta = 0;
            T _minFreq = 0, _maxFreq = 0;
        };
    }
}
