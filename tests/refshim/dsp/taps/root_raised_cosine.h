// Stand-in for SDR++ core's dsp/taps/root_raised_cosine.h.
#pragma once
#include <dsp/processor.h>

namespace dsp {
    namespace taps {
        template <class T> inline tap<T> rootRaisedCosine(int count, double beta, double Ts) {
            tap<T> taps = taps::alloc<T>(count);
            const double pi = REFSHIM_PI;
            const double half = (double)count / 2.0;
            const double limit = Ts / (4.0 * beta);
            for (int i = 0; i < count; i++) {
                const double t = (double)i - half + 0.5;
                double v;
                if (t == 0.0) { v = (1.0 + beta * (4.0 / pi - 1.0)) / Ts; }
                else if (t == limit || t == -limit) {
                    v = ((1.0 + 2.0 / pi) * sin(pi / (4.0 * beta)) + (1.0 - 2.0 / pi) * cos(pi / (4.0 * beta))) * beta / (Ts * sqrt(2.0));
                }
                else {
                    const double u = 4.0 * beta * t / Ts;
                    v = ((sin((1.0 - beta) * pi * t / Ts) + cos((1.0 + beta) * pi * t / Ts) * u) / ((1.0 - u * u) * pi * t / Ts)) / Ts;
                }
                taps.taps[i] = (T)v;
            }
            return taps;
        }
        template <class T> inline tap<T> rootRaisedCosine(int count, double beta, double symbolrate, double samplerate) {
            return rootRaisedCosine<T>(count, beta, samplerate / symbolrate);
        }
    }
}
