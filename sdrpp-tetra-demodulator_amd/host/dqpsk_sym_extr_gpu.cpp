// dqpsk_sym_extr_gpu.cpp -- see dqpsk_sym_extr_gpu.h.
#include "dqpsk_sym_extr_gpu.h"

namespace dsp {
int DQPSKSymbolExtractor::process(int count, const complex_t* in, uint8_t* out) {
    (void)in;          // the kernels sliced these symbols when they produced them
    int got = 0;
    if (tap_) got = tap_->pop(count, out, nullptr, &standarderr, &sync);
    status_ = got == count ? TETRA_OK : TETRA_ERR_ARG;
    for (int i = got; i < count; i++) out[i] = 0;
    return count;
}
}  // namespace dsp
