// bit_unpacker_gpu.h -- GPU-backed mirror of the reference's BitUnpacker block (src/dsp/bit_unpacker.h:16-34,
// src/dsp/bit_unpacker.cpp:4-10): same class name, run() body and process() signature, so src/main.cpp:91
// (bitsUnpacker.init(&symbolExtractor.out)) builds unchanged -- plus attach(&mainDemodulator).  The kernels already write one
// bit per byte, MSB of each dibit first (kernel_fused.hpp, Costas wave: bit_unpacker.cpp:6-7 for every symbol): process() hands
// those bytes on, in stream order, from the demodulator's DecisionTap -- checked against the dibits it is handed (byte equality)
// and realigned / bypassed like DQPSKSymbolExtractor's (dqpsk_sym_extr_gpu.h) when a buffer was lost or repeated upstream.
#pragma once
#include "pi4dqpsk_gpu.h"

namespace dsp {
class BitUnpacker : public Processor<uint8_t, uint8_t> {
    using base_type = Processor<uint8_t, uint8_t>;

public:
    void attach(demod::PI4DQPSK* source) { tap_ = source->openTap(); }

    // src/dsp/bit_unpacker.h:19-30
    int run() override {
        int count = base_type::_in->read();
        if (count < 0) { return -1; }
        int outCount = process(count, base_type::_in->readBuf, base_type::out.writeBuf);
        base_type::_in->flush();
        if (outCount) {
            if (!base_type::out.swap(outCount)) { return -1; }
        }
        return outCount;
    }

    // src/dsp/bit_unpacker.h:32: count dibits in -> 2 count bits out, returns 2 count
    int process(int count, const uint8_t* in, uint8_t* out);

    int lastStatus() const { return status_; }      // TETRA_OK: from the queue; TETRA_ERR_ARG: unpacked from the handed bytes themselves
    long long resyncs() const { return tap_ ? tap_->resyncs() : 0; }
    long long fallbacks() const { return fallbacks_; }
    void attachTap(std::shared_ptr<demod::DecisionTap> tap) { tap_ = std::move(tap); }      // (tests: a tap fed by hand)

private:
    std::shared_ptr<demod::DecisionTap> tap_;
    long long fallbacks_ = 0;
    int status_ = TETRA_OK;
};
}  // namespace dsp
