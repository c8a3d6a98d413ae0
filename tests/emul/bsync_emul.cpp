// Host build of the burst synchroniser's logic (sdrpp-tetra-demodulator_amd/csrc/bsync_core.hpp): same event-driven
// state machine, bitmaps and literal fallback as the kernel, with the kernel's parallel phases done sequentially.
// Test infrastructure: checked against the literal restatement fed one bit per call (oracle/burst_sync_oracle.c).
#define TETRA_HOST_EMUL 1
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../sdrpp-tetra-demodulator_amd/csrc/bsync_core.hpp"

using namespace bsync_core;

extern "C" int bsync_emul_process(State* st, uint8_t* carry, const uint8_t* bits, int n_new, uint8_t* frames, int32_t* types,
                                  uint32_t* bitnums, int max_frames, int use_batch) {
    const int words = stream_words(n_new);
    std::vector<uint32_t> s(words, 0), ms(words, 0), m1(words, 0), m2(words, 0), ma(words, 0);
    const int x0 = kOff - (int)st->bits_in_buf, xe = kOff + n_new;
    for (int x = x0; x < kOff; ++x) s[x >> 5] |= (uint32_t)(carry[x - x0] & 1u) << (31 - (x & 31));
    for (int j = 0; j < n_new; ++j) s[(kOff + j) >> 5] |= (uint32_t)(bits[j] & 1u) << (31 - ((kOff + j) & 31));
    for (int w = x0 >> 5; w <= (xe - 1) >> 5 && w + 2 < words; ++w) { match_word(s.data(), w, x0, xe, ms[w], m1[w], m2[w]); ma[w] = ms[w] | m1[w] | m2[w]; }
    int carry_x = 0, overflow = 0;
    auto emit = [&](int f, int bx, int type, uint32_t bitnum) {
        if (f >= max_frames) { overflow = 1; return; }
        for (int i = 0; i < kTs; ++i) frames[(size_t)f * 512 + i] = (uint8_t)get_bit(s.data(), bx + i);
        types[f] = type;
        bitnums[f] = bitnum;
    };
    // the kernel's batch of LOCKED frames (one per lane, 64 per round, stopping behind the first frame that unlocks), frame by frame
    auto batch = [&](int bx, int K, int f0, uint32_t abs_bx, bool& unlocked) {
        if (!use_batch) return 0;
        int done = 0;
        while (done < K) {
            const FrameEval e = locked_frame_eval(s.data(), ms.data(), m1.data(), m2.data(), ma.data(), bx + kTs * done);
            emit(f0 + done, bx + kTs * done, e.reported, abs_bx + (uint32_t)(kTs * done));
            done++;
            if (e.unlocks) { unlocked = true; break; }
        }
        return done;
    };
    const int n = run(*st, s.data(), ms.data(), m1.data(), m2.data(), ma.data(), n_new, carry_x,
                      [](const uint32_t* m, int a, int b) { return first_set(m, a, b); }, emit, batch);
    for (int x = carry_x; x < xe; ++x) carry[x - carry_x] = (uint8_t)get_bit(s.data(), x);
    return overflow ? -1 : n;
}
