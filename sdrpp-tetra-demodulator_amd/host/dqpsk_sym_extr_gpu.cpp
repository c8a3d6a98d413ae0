// dqpsk_sym_extr_gpu.cpp -- see dqpsk_sym_extr_gpu.h.
#include "dqpsk_sym_extr_gpu.h"

namespace dsp {
int DQPSKSymbolExtractor::process(int count, const complex_t* in, uint8_t* out) {
    if (count <= 0) return count < 0 ? 0 : count;
    // What the handed symbols themselves say, from their signs alone (dqpsk_sym_extr.cpp:6-7,32-52: quadrant index, difference
    // to the previous one, remap) -- the cross-check of the side channel, and the fallback when it has nothing for them.
    expect_.resize((size_t)count);
    uint8_t prev = prev_;
    for (int i = 0; i < count; i++) {
        const bool a = in[i].im < 0, b = in[i].re < 0;
        const uint8_t sym = (uint8_t)((a << 1) | (a != b));
        static const uint8_t remap[4] = { 0, 1, 3, 2 };
        expect_[(size_t)i] = remap[(sym - prev + 4) & 3];
        prev = sym;
    }
    prev_ = prev;
    int got = -1;
    if (tap_) got = tap_->popAligned(count, expect_.data(), contiguous_, out, nullptr, &standarderr, &sync);
    if (got != count && tap_ && !pending_.empty() && pending_.size() + (size_t)count >= (size_t)demod::DecisionTap::kMinMatch) {
        // buffers too short to realign on by themselves (< DecisionTap::kMinMatch symbols): together with the dibits this block
        // sliced locally since it lost the queue they may be long enough -- look for the whole run, keep this buffer's part
        const size_t np = pending_.size();
        std::vector<uint8_t> run(pending_);
        run.insert(run.end(), expect_.begin(), expect_.end());
        std::vector<uint8_t> dec(run.size());
        if (tap_->popAligned((int)run.size(), run.data(), false, dec.data(), nullptr, &standarderr, &sync) == (int)run.size()) {
            for (int i = 0; i < count; i++) out[i] = dec[np + (size_t)i];
            got = count;
        }
    }
    if (got == count) {
        status_ = TETRA_OK;        // the kernels' decisions for exactly these symbols (and their statistic marks)
        contiguous_ = true;
        pending_.clear();
    } else {
        // not in the queue (a buffer handed twice, a stream from elsewhere, no demodulator attached): these symbols' own signs
        // decide, like the reference's block; nothing was consumed, standarderr / sync keep their last values
        for (int i = 0; i < count; i++) out[i] = expect_[(size_t)i];
        status_ = TETRA_ERR_ARG;
        fallbacks_++;
        contiguous_ = false;       // where the stream continues from here is unknown: the next first dibit proves nothing
        pending_.insert(pending_.end(), expect_.begin(), expect_.end());
        if (pending_.size() > kPendingMax) pending_.erase(pending_.begin(), pending_.end() - (std::ptrdiff_t)kPendingMax);
    }
    return count;
}
}  // namespace dsp
