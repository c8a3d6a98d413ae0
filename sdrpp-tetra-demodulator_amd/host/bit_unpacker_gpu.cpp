// bit_unpacker_gpu.cpp -- see bit_unpacker_gpu.h.
#include "bit_unpacker_gpu.h"

namespace dsp {
int BitUnpacker::process(int count, const uint8_t* in, uint8_t* out) {
    if (count <= 0) return 0;
    // the handed dibits are what the queue must hold for them (byte equality; no arithmetic)
    int got = -1;
    if (tap_) got = tap_->popAligned(count, in, true, nullptr, out, nullptr, nullptr);
    if (got == count) status_ = TETRA_OK;
    else {
        // not in the queue: bit_unpacker.cpp:6-7 on the handed bytes themselves; nothing was consumed
        for (int i = 0; i < count; i++) { out[2 * i] = (uint8_t)((in[i] >> 1) & 1); out[2 * i + 1] = (uint8_t)(in[i] & 1); }
        status_ = TETRA_ERR_ARG;
        fallbacks_++;
    }
    return count * 2;
}
}  // namespace dsp
