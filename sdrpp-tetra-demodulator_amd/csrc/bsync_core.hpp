// bsync_core.hpp -- the burst synchroniser's logic (include/tetra_burst_sync.h), shared by the gfx950 kernel
// (tetra_burst_sync.hip) and its host build (tests/emul/bsync_emul.cpp, -DTETRA_HOST_EMUL).
//
// Contract: for one channel, one call over n_new bits leaves exactly the state, and reports exactly the frames, that the
// reference's tetra_burst_sync_in() (src/decoder/src/phy/tetra_burst_sync.c:54-155) produces when it is handed the same
// bits ONE BIT PER CALL.  (The reference consumes at most one frame per call and its 4096-byte buffer drops the oldest
// bits when a call overfills it, :38-51, so its result depends on how the caller chunks the stream; one bit per call is
// the chunking-independent limit of the small stream buffers the plugin feeds it, src/dsp/osmotetra_dec.h:182-184.)
//
// The per-bit calls are not replayed one by one.  The stream of a call is laid out once in "coordinates": the carried
// buffer occupies [kOff - bits_in_buf, kOff), the new bits [kOff, kOff + n_new); bits are packed 32 per word, MSB first.
// Three match bitmaps (sync, normal 1, normal 2 training sequence truly present at x and inside the stream) are built
// in parallel.  The state machine then jumps from event to event:
//   UNLOCKED     the first call that can search is A1 = max(A + 1, Bx + 1020) (buffer [Bx, A), :66-71).  Every later call
//                re-scans the whole buffer, but the only position that was not already rejected by an earlier call is the
//                newest one that fits, so the hit is the first true sync match p >= Bx + 21 and it is found by the call
//                A = max(A1, p + 38).  The first 21 buffer positions are special: the reference's pre-filter is
//                misaligned there (tetra_burst.c:289-296), so a true match only counts if that filter fires too; if the
//                bitmap shows a true match in that zone the call is evaluated literally (literal_find) -- rare.
//   KNOW_FSTART  nothing happens until the call A = max(A + 1, next_frame_start) (:90-92); that call moves the buffer start
//                to the frame start and falls through into LOCKED (:93-104).
//   LOCKED       a frame is consumed by every call that finds >= 510 bits buffered (:105-150): search [Bx, A) for the first
//                of {sync, normal 1, normal 2}; sync at 214 / normal at 244 -> burst reported with its type; sync
//                elsewhere or nothing found -> UNLOCKED; normal elsewhere -> frame dropped, still LOCKED.
// All bit numbers are uint32 and wrap like the reference's unsigned ints.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__) && !defined(TETRA_HOST_EMUL)
#define BS_FN __device__ __forceinline__
#else
#define BS_FN static inline
#endif

namespace bsync_core {

constexpr int kOff = 4096;          // coordinate of the first new bit = sizeof(trs->bitbuf), tetra_burst_sync.h:15
constexpr int kBuf = 4096;
constexpr int kTs = 510;            // TETRA_BITS_PER_TS
constexpr int kLook = 64;           // zero bits kept past the end of the stream for window reads
enum { kUnlocked = 0, kKnowFstart = 1, kLocked = 2 };                   // enum rx_state, tetra_burst_sync.h:6-10
enum { kNorm1 = 0, kNorm2 = 1, kNorm3 = 2, kSync = 3, kExt = 4 };       // enum tetra_train_seq, tetra_burst.h:26-32

struct State {                      // struct tetra_rx_state without its bitbuf (kept separately, one byte per bit)
    int32_t state;
    uint32_t bits_in_buf;
    uint32_t bitbuf_start_bitnum;
    uint32_t next_frame_start_bitnum;
};

// first 22 bits of y (sync), n, p, q (normal 1-3), x (extended) -- EN 300 392-2 9.4.4.3.2-4 -- MSB first, and the
// remaining 16 bits of y
constexpr uint32_t kHeadY = 0x30673a, kHeadN = 0x343a74, kHeadP = 0x1e90de, kHeadQ = 0x2dc1ad, kHeadX = 0x2743a7;
constexpr uint32_t kTailY = 0x7067;

constexpr int stream_words(int max_new) { return (kOff + max_new + kLook + 31) / 32 + 2; }

// ---- packed stream access -------------------------------------------------------------------------------------------
BS_FN uint32_t get_bit(const uint32_t* s, int x) { return (s[x >> 5] >> (31 - (x & 31))) & 1u; }
// bits x .. x+len-1 (len <= 32) as a number, first bit most significant
BS_FN uint32_t window(const uint32_t* s, int x, int len) {
    const uint64_t two = ((uint64_t)s[x >> 5] << 32) | s[(x >> 5) + 1];
    return (uint32_t)(two >> (64 - len - (x & 31))) & (uint32_t)((1ull << len) - 1);
}

// 32 stream bits starting k bits after the start of word w (k = 0..63), first bit most significant
BS_FN uint32_t shifted_word(uint32_t w0, uint32_t w1, uint32_t w2, int k) {
    const uint32_t hi = k < 32 ? w0 : w1, lo = k < 32 ? w1 : w2;
    const int r = k & 31;
    return r == 0 ? hi : (uint32_t)((((uint64_t)hi << 32) | lo) >> (32 - r));
}

BS_FN uint32_t bit_reverse(uint32_t v) {
#if defined(__HIPCC__) && !defined(TETRA_HOST_EMUL)
    return __builtin_bitreverse32(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
#endif
}

// bit b (LSB = position 32w) set where positions lo <= 32w + b < hi
BS_FN uint32_t range_mask(int w, int lo, int hi) {
    int a = lo - 32 * w, b = hi - 32 * w;
    a = a < 0 ? 0 : a;
    b = b > 32 ? 32 : b;
    if (a >= b) return 0u;
    const uint32_t upto_b = b == 32 ? 0xffffffffu : ((1u << b) - 1u);
    return upto_b & (0xffffffffu << a);
}

// match bits of the 32 positions 32w .. 32w+31 (LSB = position 32w) for the three sequences; a position counts only if
// the whole sequence lies inside [x0, xe).  Bit-sliced: for every offset k into the sequence, the 32 candidate windows'
// k-th bits are one funnel-shifted word, AND-ed (or AND-NOT-ed) into the three running match words.
BS_FN void match_word(const uint32_t* s, int w, int x0, int xe, uint32_t& m_sync, uint32_t& m_n1, uint32_t& m_n2) {
    const uint32_t w0 = s[w], w1 = s[w + 1], w2 = s[w + 2];
    uint32_t ay = 0xffffffffu, an = 0xffffffffu, ap = 0xffffffffu;
    constexpr uint64_t seq_y = ((uint64_t)kHeadY << 16) | kTailY;          // 38 bits, first bit = bit 37
#pragma unroll
    for (int k = 0; k < 38; ++k) {
        const uint32_t v = shifted_word(w0, w1, w2, k);
        ay &= ((seq_y >> (37 - k)) & 1u) ? v : ~v;
        if (k < 22) {
            an &= ((kHeadN >> (21 - k)) & 1u) ? v : ~v;
            ap &= ((kHeadP >> (21 - k)) & 1u) ? v : ~v;
        }
    }
    // ay/an/ap: bit 31 - b = position 32w + b  ->  bit b
    m_sync = bit_reverse(ay) & range_mask(w, x0, xe - 38 + 1);
    m_n1 = bit_reverse(an) & range_mask(w, x0, xe - 22 + 1);
    m_n2 = bit_reverse(ap) & range_mask(w, x0, xe - 22 + 1);
}

// first set bit of bitmap m (bit x at word x>>5, bit x&31) in [a, b), or -1
BS_FN int first_set(const uint32_t* m, int a, int b) {
    if (a >= b) return -1;
    for (int w = a >> 5; w <= (b - 1) >> 5; ++w) {
        uint32_t v = m[w];
        if (w == (a >> 5)) v &= 0xffffffffu << (a & 31);
        if (w == ((b - 1) >> 5) && ((b & 31) != 0)) v &= (1u << (b & 31)) - 1u;
        if (v) return 32 * w + __builtin_ctz(v);
    }
    return -1;
}

// tetra_find_train_seq() (tetra_burst.c:271-341) evaluated literally on the packed stream, buffer = [bx, bx + n)
BS_FN int literal_find(const uint32_t* s, int bx, int n, uint32_t mask, int& offs) {
    uint32_t filter = 0;
    for (int i = 0; i < 20; ++i) filter = (filter << 1) | get_bit(s, bx + i);
    for (int cur = 0; cur < n; ++cur) {
        filter = ((filter << 1) | get_bit(s, bx + cur + 21)) & 0x3fffffu;
        if (filter != kHeadY && filter != kHeadN && filter != kHeadP && filter != kHeadQ && filter != kHeadX) continue;
        const int remain = n - cur, x = bx + cur;
        if ((mask & (1u << kSync)) && remain >= 38 && window(s, x, 22) == kHeadY && window(s, x + 22, 16) == kTailY) { offs = cur; return kSync; }
        if ((mask & (1u << kNorm1)) && remain >= 22 && window(s, x, 22) == kHeadN) { offs = cur; return kNorm1; }
        if ((mask & (1u << kNorm2)) && remain >= 22 && window(s, x, 22) == kHeadP) { offs = cur; return kNorm2; }
        // normal 3 / extended are never enabled by the synchroniser
    }
    return -1;
}

// `first(m, a, b)` below is the "first set bit of bitmap m in [a, b)" primitive: first_set on the host, a wave-cooperative
// version (one word per lane + ballot) in the kernel, where all 64 lanes run the state machine in lock step.

// the LOCKED state's search over the buffer [bx, bx + n), n >= 510: first of {sync, normal 1, normal 2}.
// m_any = m_sync | m_n1 | m_n2 answers the common case with one search.
template <class First>
BS_FN int locked_find(const uint32_t* s, const uint32_t* m_sync, const uint32_t* m_n1, const uint32_t* m_n2, const uint32_t* m_any,
                      int bx, int n, int& offs, First first) {
    const uint32_t mask = (1u << kNorm1) | (1u << kNorm2) | (1u << kSync);
    const int p = first(m_any, bx, bx + n - 22 + 1);
    if (p < 0) return -1;
    if (p < bx + 21) return literal_find(s, bx, n, mask, offs);            // misaligned-filter zone: evaluate literally
    const bool is_sync = (m_sync[p >> 5] >> (p & 31)) & 1u;
    if (!is_sync || p + 38 <= bx + n) {
        offs = p - bx;
        return is_sync ? kSync : (((m_n1[p >> 5] >> (p & 31)) & 1u) ? kNorm1 : kNorm2);
    }
    // a sync sequence that does not fit the buffer any more: the three sequences separately
    const int ps = first(m_sync, bx + 21, bx + n - 38 + 1);
    const int p1 = first(m_n1, bx + 21, bx + n - 22 + 1);
    const int p2 = first(m_n2, bx + 21, bx + n - 22 + 1);
    int best = -1, type = -1;
    if (ps >= 0) { best = ps; type = kSync; }
    if (p1 >= 0 && (best < 0 || p1 < best)) { best = p1; type = kNorm1; }
    if (p2 >= 0 && (best < 0 || p2 < best)) { best = p2; type = kNorm2; }
    if (best >= 0) offs = best - bx;
    return type;
}

// One LOCKED call on a buffer of exactly one frame, [bx, bx + 510): what tetra_burst_sync_in reports for it (the rx_cb type, or -1)
// and whether the receiver falls back to UNLOCKED behind it (tetra_burst_sync.c:105-150).  A pure function of the bitmaps: frames in
// LOCKED steady state can be evaluated side by side, one per lane (run()'s `batch`).
struct FrameEval { int reported; bool unlocks; };
BS_FN FrameEval locked_frame_eval(const uint32_t* s, const uint32_t* m_sync, const uint32_t* m_n1, const uint32_t* m_n2, const uint32_t* m_any,
                                  int bx) {
    int offs = 0;
    const int rc = locked_find(s, m_sync, m_n1, m_n2, m_any, bx, kTs, offs, [](const uint32_t* m, int a, int b) { return first_set(m, a, b); });
    FrameEval e = { -1, false };
    if (rc == kSync) {
        if (offs == 214) e.reported = rc;
        else e.unlocks = true;
    } else if (rc == kNorm1 || rc == kNorm2) {
        if (offs == 244) e.reported = rc;
    } else {
        e.unlocks = true;
    }
    return e;
}

// Runs the state machine over the new bits.  emit(f, bx, type, bitnum) is called for the f-th consumed frame (buffer
// coordinate of its first bit, the reference's rx_cb type or -1, absolute bit number of its first bit).  On return
// st holds the new state and carry_x the coordinate of the first bit that stays buffered ([carry_x, xe) = the new bitbuf).
// batch(bx, K, f0, abs_bx, unlocked): LOCKED steady state (the fed position a equals the buffer start bx, so every call consumes
// exactly the next 510 bits): evaluate up to K whole frames [bx + 510 k, bx + 510 (k + 1)) and emit them as frames f0, f0 + 1, ... up to
// and including the first one that unlocks the receiver; returns how many it consumed (0 = not taken: the serial path runs).  The
// kernel does this one frame per lane -- ~70 dependent event steps per second of signal become two -- the host emulation in a loop.
template <class First, class Emit, class Batch>
BS_FN int run(State& st, const uint32_t* s, const uint32_t* m_sync, const uint32_t* m_n1, const uint32_t* m_n2, const uint32_t* m_any,
              int n_new, int& carry_x, First first, Emit emit, Batch batch) {
    const int x0 = kOff - (int)st.bits_in_buf, xe = kOff + n_new;
    const uint32_t abs0 = st.bitbuf_start_bitnum;               // absolute bit number of coordinate x0
    auto abs_of = [&](int x) { return abs0 + (uint32_t)(x - x0); };
    int state = st.state, bx = x0, a = kOff, frames = 0;
    uint32_t nfs = st.next_frame_start_bitnum;

    auto locked_call = [&]() {                                  // tetra_burst_sync.c:105-150, buffer [bx, a)
        const int n = a - bx;
        if (n < kTs) return;
        int offs = 0, reported = -1;
        const int rc = locked_find(s, m_sync, m_n1, m_n2, m_any, bx, n, offs, first);
        if (rc == kSync) {
            if (offs == 214) reported = rc;
            else state = kUnlocked;
        } else if (rc == kNorm1 || rc == kNorm2) {
            if (offs == 244) reported = rc;
        } else {
            state = kUnlocked;
        }
        emit(frames++, bx, reported, abs_of(bx));
        bx += kTs;
        nfs += kTs;
    };

    while (a < xe) {
        if (state == kUnlocked) {
            const int a1 = (a + 1 > bx + 2 * kTs) ? a + 1 : bx + 2 * kTs;
            if (a1 > xe) { a = xe; break; }
            const int bx1 = (a1 - kBuf > bx) ? a1 - kBuf : bx;
            if (first(m_sync, bx1, bx1 + 21) >= 0) {        // a true match in the misaligned-filter zone: literal call
                int offs = 0;
                const int rc = literal_find(s, bx1, a1 - bx1, 1u << kSync, offs);
                a = a1;
                bx = bx1;
                if (rc >= 0) { nfs = abs_of(bx) + (uint32_t)offs + 296u; state = kKnowFstart; }
                continue;
            }
            const int p = first(m_sync, bx1 + 21, xe - 38 + 1);
            if (p < 0) { a = xe; break; }
            a = (p + 38 > a1) ? p + 38 : a1;
            bx = (a - kBuf > bx) ? a - kBuf : bx;
            nfs = abs_of(p) + 296u;
            state = kKnowFstart;
        } else if (state == kKnowFstart) {
            const int nfs_x = x0 + (int)(int32_t)(nfs - abs0);
            const int at = (a + 1 > nfs_x) ? a + 1 : nfs_x;
            if (at > xe) { a = xe; break; }
            a = at;
            bx = nfs_x;
            nfs += kTs;
            state = kLocked;
            locked_call();
        } else {
            if (a == bx && xe - bx >= 2 * kTs) {
                bool unlocked = false;
                const int done = batch(bx, (xe - bx) / kTs, frames, abs_of(bx), unlocked);
                if (done > 0) {                                 // `done` LOCKED calls of exactly one frame each, a = bx + 510 after every one
                    frames += done;
                    bx += kTs * done;
                    nfs += (uint32_t)(kTs * done);
                    a = bx;
                    if (unlocked) state = kUnlocked;
                    continue;
                }
            }
            const int an = (a + 1 > bx + kTs) ? a + 1 : bx + kTs;
            if (an > xe) { a = xe; break; }
            a = an;
            bx = (a - kBuf > bx) ? a - kBuf : bx;
            locked_call();
        }
    }
    bx = (a - kBuf > bx) ? a - kBuf : bx;                       // make_bitbuf_space of the calls skipped at the end
    st.state = state;
    st.bits_in_buf = (uint32_t)(xe - bx);
    st.bitbuf_start_bitnum = abs_of(bx);
    st.next_frame_start_bitnum = nfs;
    carry_x = bx;
    return frames;
}

}  // namespace bsync_core
