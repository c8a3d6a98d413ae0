// Stand-in for SDR++ core's dsp/filter/fir.h: history of size-1 zeros, taps applied un-reversed.
#pragma once
#include <type_traits>
#include <dsp/processor.h>

namespace dsp {
    namespace filter {
        template <class D, class T> class FIR : public Processor<D, D> {
            using base_type = Processor<D, D>;
        public:
            FIR() {}
            ~FIR() { if (buffer) { buffer::free(buffer); } }
            void init(stream<D>* in, tap<T>& taps) {
                _taps = taps;
                buffer = buffer::alloc<D>(STREAM_BUFFER_SIZE + 64000);
                bufStart = &buffer[_taps.size - 1];
                base_type::init(in);
            }
            void setTaps(tap<T>& taps) {
                const int old = _taps.size;
                _taps = taps;
                bufStart = &buffer[_taps.size - 1];
                // keep the newest history next to bufStart, zero what a longer filter newly looks back at
                if (_taps.size < old) { memmove(buffer, &buffer[old - _taps.size], (size_t)(_taps.size - 1) * sizeof(D)); }
                else if (_taps.size > old) {
                    memmove(&buffer[_taps.size - old], buffer, (size_t)(old - 1) * sizeof(D));
                    buffer::clear<D>(buffer, _taps.size - old);
                }
            }
            void reset() { buffer::clear<D>(buffer, _taps.size - 1); }
            inline int process(int count, const D* in, D* out) {
                memcpy(bufStart, in, (size_t)count * sizeof(D));
                for (int i = 0; i < count; i++) {
                    if constexpr (std::is_same_v<T, float>) { volk_32fc_32f_dot_prod_32fc((lv_32fc_t*)&out[i], (lv_32fc_t*)&buffer[i], _taps.taps, _taps.size); }
                    else { volk_32fc_x2_dot_prod_32fc((lv_32fc_t*)&out[i], (lv_32fc_t*)&buffer[i], (lv_32fc_t*)_taps.taps, _taps.size); }
                }
                memmove(buffer, &buffer[count], (size_t)(_taps.size - 1) * sizeof(D));
                return count;
            }
            int run() { return -1; }
        protected:
            tap<T> _taps;
            D* buffer = nullptr;
            D* bufStart = nullptr;
        };
    }
}
