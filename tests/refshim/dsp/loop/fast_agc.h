// Stand-in for SDR++ core's dsp/loop/fast_agc.h.
#pragma once
#include <dsp/processor.h>

namespace dsp {
    namespace loop {
        template <class T> class FastAGC : public Processor<T, T> {
            using base_type = Processor<T, T>;
        public:
            FastAGC() {}
            void init(stream<T>* in, double setPoint, double maxGain, double rate, double initGain = 1.0) {
                _setPoint = setPoint; _maxGain = maxGain; _rate = rate; _initGain = initGain; _gain = initGain;
                base_type::init(in);
            }
            void setRate(double rate) { _rate = rate; }
            void reset() { _gain = _initGain; }
            inline int process(int count, T* in, T* out) {
                for (int i = 0; i < count; i++) {
                    out[i] = in[i] * _gain;
                    const float amp = out[i].amplitude();
                    _gain += (_setPoint - amp) * _rate;
                    if (_gain > _maxGain) { _gain = _maxGain; }
                }
                return count;
            }
            int run() { return -1; }
        protected:
            float _gain = 1, _setPoint = 1, _rate = 0, _maxGain = 0, _initGain = 1;
        };
    }
}
