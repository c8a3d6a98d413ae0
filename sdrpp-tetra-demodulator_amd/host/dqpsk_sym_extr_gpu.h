// dqpsk_sym_extr_gpu.h -- GPU-backed mirror of the reference's DQPSKSymbolExtractor block (src/dsp/dqpsk_sym_extr.h:19-46,
// src/dsp/dqpsk_sym_extr.cpp:4-55): same class name, run() body, process() signature and public `sync` / `standarderr`
// members (read by the GUI at src/main.cpp:211,215), so the plugin's wiring (src/main.cpp:90: symbolExtractor.init(&demodStream))
// builds unchanged -- plus ONE added call, attach(&mainDemodulator), naming the GPU demodulator whose symbols this block is
// handed.  No DSP arithmetic happens on the host: the slicer / differential decoder ran in the kernels for these very symbols
// (kernel_fused.hpp, Costas wave) and the statistic is kept by k_quality (tetra_demod_get_quality); process() takes both from
// the demodulator's DecisionTap in stream order.  The symbol VALUES in `in` are not looked at.
#pragma once
#include "pi4dqpsk_gpu.h"

namespace dsp {
class DQPSKSymbolExtractor : public Processor<complex_t, uint8_t> {
    using base_type = Processor<complex_t, uint8_t>;

public:
    // The demodulator this block sits behind (directly or through SDR++'s splitter, src/main.cpp:85-90: every symbol of
    // mainDemodulator.out reaches this block once, in order).  Call before the blocks are started.
    void attach(demod::PI4DQPSK* source) { tap_ = source->openTap(); }

    // src/dsp/dqpsk_sym_extr.h:22-33
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

    // src/dsp/dqpsk_sym_extr.h:35: count symbols in -> count dibits out (bit 1 = first bit of the TETRA symbol).  Like the
    // reference's it cannot fail; a stream that does not come from the attached demodulator (fewer decisions queued than
    // symbols handed in) yields zero dibits for the surplus and lastStatus() == TETRA_ERR_ARG.
    int process(int count, const complex_t* in, uint8_t* out);

    bool sync = false;          // src/dsp/dqpsk_sym_extr.h:36-37
    float standarderr = 0;

    int lastStatus() const { return status_; }

private:
    std::shared_ptr<demod::DecisionTap> tap_;
    int status_ = TETRA_OK;
};
}  // namespace dsp
