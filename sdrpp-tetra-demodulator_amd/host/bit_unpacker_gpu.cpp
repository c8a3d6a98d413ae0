// bit_unpacker_gpu.cpp -- see bit_unpacker_gpu.h.
#include "bit_unpacker_gpu.h"

namespace dsp {
int BitUnpacker::process(int count, const uint8_t* in, uint8_t* out) {
    (void)in;          // the kernels unpacked these dibits when they decided them
    int got = 0;
    if (tap_) got = tap_->pop(count, nullptr, out, nullptr, nullptr);
    status_ = got == count ? TETRA_OK : TETRA_ERR_ARG;
    for (int i = 2 * got; i < 2 * count; i++) out[i] = 0;
    return count * 2;
}
}  // namespace dsp
