// dqpsk_sym_extr_gpu.h -- GPU-backed mirror of the reference's DQPSKSymbolExtractor block (src/dsp/dqpsk_sym_extr.h:19-46,
// src/dsp/dqpsk_sym_extr.cpp:4-55): same class name, run() body, process() signature and public `sync` / `standarderr`
// members (read by the GUI at src/main.cpp:211,215), so the plugin's wiring (src/main.cpp:90: symbolExtractor.init(&demodStream))
// builds unchanged -- plus ONE added call, attach(&mainDemodulator), naming the GPU demodulator whose symbols this block is
// handed.  No DSP arithmetic happens on the host: the slicer / differential decoder ran in the kernels for these very symbols
// (kernel_fused.hpp, Costas wave) and the statistic is kept by k_quality (tetra_demod_get_quality); process() takes both from
// the demodulator's DecisionTap in stream order -- CHECKED against the symbols it is handed: two sign tests per symbol give the
// dibit those symbols imply, the queue head must agree, and when it does not (a symbol buffer lost, repeated or held back
// between mainDemodulator.out and this block: src/main.cpp:85-90,130-167) the block finds the offset at which the queue lines up
// with its input again, drops what lies before it and counts the event (resyncs()); symbols that are nowhere in the queue are
// sliced from their own signs (fallbacks(), lastStatus() == TETRA_ERR_ARG), so the dibits that leave this block are the
// reference's for the symbols it was handed, whatever happened upstream.
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
    // reference's it cannot fail.  lastStatus() == TETRA_OK: the kernels' decisions (and statistic) for exactly these symbols;
    // TETRA_ERR_ARG: the symbols were not in the attached demodulator's queue and were sliced from their own signs.
    int process(int count, const complex_t* in, uint8_t* out);

    bool sync = false;          // src/dsp/dqpsk_sym_extr.h:36-37
    float standarderr = 0;

    int lastStatus() const { return status_; }
    // realignments of the side channel since attach (a gap upstream each), decisions skipped doing so, buffers sliced locally
    long long resyncs() const { return tap_ ? tap_->resyncs() : 0; }
    long long skippedSymbols() const { return tap_ ? tap_->skippedSymbols() : 0; }
    long long fallbacks() const { return fallbacks_; }
    void attachTap(std::shared_ptr<demod::DecisionTap> tap) { tap_ = std::move(tap); }      // (tests: a tap fed by hand)

private:
    std::shared_ptr<demod::DecisionTap> tap_;
    std::vector<uint8_t> expect_;
    std::vector<uint8_t> pending_;      // dibits sliced locally since the queue was lost (newest kPendingMax): context for realigning on short buffers
    static constexpr size_t kPendingMax = 64;
    uint8_t prev_ = 0;            // quadrant of the last symbol handed in (src/dsp/dqpsk_sym_extr.h:42)
    bool contiguous_ = true;      // the last buffer came out of the queue: this buffer's first dibit is a difference within the stream
    long long fallbacks_ = 0;
    int status_ = TETRA_OK;
};
}  // namespace dsp
