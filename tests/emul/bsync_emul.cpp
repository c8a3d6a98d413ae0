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

// ---- the demultiplexer kernels (csrc/demux_core.hpp): every (workgroup, thread) of the launch tetra_burst_sync.hip's launchers would
// make, in turn (threads descending inside a workgroup, workgroups ascending: no thread may depend on another's writes) -------------
#include "../../sdrpp-tetra-demodulator_amd/csrc/demux_core.hpp"

namespace {
template <class F> void for_grid(long long blocks, F f) {
    for (long long b = 0; b < blocks; ++b)
        for (int t = demux_core::kThreads - 1; t >= 0; --t) f((uint32_t)b, (uint32_t)t);
}
}  // namespace

// frames: [n][512] bytes or, packed != 0, [n][16] words.  Returns 0, or -1 for arguments the launchers refuse.
extern "C" int bsync_emul_demux(const uint8_t* frames, int packed, const int32_t* frame_type, int n, int tpsap, int blk_num, uint8_t* rows,
                                int row_stride, int32_t* valid) {
    using namespace demux_core;
    const PiecesLut lut = lut_for(tpsap, blk_num);
    int longest = 0;
    for (const Pieces* p : { &lut.sync, &lut.norm1, &lut.norm2 }) longest = p->len0 + p->len1 > longest ? p->len0 + p->len1 : longest;
    if (longest == 0 || row_stride < longest || (row_stride & 3)) return -1;
    const bool wide = !(row_stride & 7) && !((uintptr_t)rows & 7);
    if (use_rows_kernel(packed != 0, wide, row_stride)) {
        const int row_u = row_stride >> 3, rpw = 64 / row_u;
        for_grid(rows_grid(n, row_stride), [&](uint32_t b, uint32_t t) {
            rows_thread<false>(b, t, frames, frame_type, nullptr, n, lut, row_u, rpw, (uint32_t)((65536 + row_u - 1) / row_u), rows, valid);
        });
        return 0;
    }
    for_grid(units_grid(n, row_stride, wide), [&](uint32_t b, uint32_t t) {
        if (packed && wide) demux_thread<true, true>(b, t, frames, frame_type, n, tpsap, blk_num, rows, row_stride, valid);
        else if (packed) demux_thread<true, false>(b, t, frames, frame_type, n, tpsap, blk_num, rows, row_stride, valid);
        else if (wide) demux_thread<false, true>(b, t, frames, frame_type, n, tpsap, blk_num, rows, row_stride, valid);
        else demux_thread<false, false>(b, t, frames, frame_type, n, tpsap, blk_num, rows, row_stride, valid);
    });
    return 0;
}

// the compacting form: row_frame / n_rows as k_demux_count / _scan / _index leave them (frame order), then the gather
extern "C" int bsync_emul_demux_compact(const uint8_t* frames, int packed, const int32_t* frame_type, int n, int tpsap, int blk_num,
                                        uint8_t* rows, int row_stride, int32_t* row_frame, int32_t* n_rows) {
    using namespace demux_core;
    const PiecesLut lut = lut_for(tpsap, blk_num);
    int longest = 0;
    for (const Pieces* p : { &lut.sync, &lut.norm1, &lut.norm2 }) longest = p->len0 + p->len1 > longest ? p->len0 + p->len1 : longest;
    if (longest == 0 || row_stride < longest || (row_stride & 3)) return -1;
    int cnt = 0;
    for (int r = 0; r < n; ++r)
        if (pieces_for(frame_type[r], tpsap, blk_num).len0 > 0) row_frame[cnt++] = r;
    *n_rows = cnt;
    const bool wide = !(row_stride & 7) && !((uintptr_t)rows & 7);
    if (use_rows_kernel(packed != 0, wide, row_stride)) {
        const int row_u = row_stride >> 3, rpw = 64 / row_u;
        for_grid(rows_grid(n, row_stride), [&](uint32_t b, uint32_t t) {
            rows_thread<true>(b, t, frames, frame_type, row_frame, cnt, lut, row_u, rpw, (uint32_t)((65536 + row_u - 1) / row_u), rows, nullptr);
        });
        return 0;
    }
    for_grid(units_grid(n, row_stride, wide), [&](uint32_t b, uint32_t t) {
        if (packed && wide) gather_thread<true, true>(b, t, frames, frame_type, row_frame, cnt, n, tpsap, blk_num, rows, row_stride);
        else if (packed) gather_thread<true, false>(b, t, frames, frame_type, row_frame, cnt, n, tpsap, blk_num, rows, row_stride);
        else if (wide) gather_thread<false, true>(b, t, frames, frame_type, row_frame, cnt, n, tpsap, blk_num, rows, row_stride);
        else gather_thread<false, false>(b, t, frames, frame_type, row_frame, cnt, n, tpsap, blk_num, rows, row_stride);
    });
    return 0;
}
