// Stand-in for SDR++ core's dsp/taps/windowed_sinc.h + window/nuttall.h.
#pragma once
#include <dsp/processor.h>

namespace dsp {
    namespace window {
        inline double nuttall(double n, double N) {
            static const double c[4] = { 0.355768, 0.487396, 0.144232, 0.012604 };
            double win = 0.0, sign = 1.0;
            for (int i = 0; i < 4; i++) {
                win += sign * c[i] * cos(2.0 * REFSHIM_PI * (double)i * n / N);
                sign = -sign;
            }
            return win;
        }
    }
    namespace taps {
        template <class T, class Func> inline tap<T> windowedSinc(int count, double omega, Func window, double norm = 1.0) {
            tap<T> taps = taps::alloc<T>(count);
            const double half = (double)count / 2.0;
            const double corr = norm * omega / REFSHIM_PI;
            for (int i = 0; i < count; i++) {
                const double t = (double)i - half + 0.5;
                taps.taps[i] = (T)(math::sinc(t * omega) * window(t - half, count) * corr);
            }
            return taps;
        }
    }
}
