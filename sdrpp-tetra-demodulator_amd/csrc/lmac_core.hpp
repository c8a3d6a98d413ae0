// lmac_core.hpp -- lane-level code of the batched lower-MAC channel decoder (include/tetra_lmac.h).
//
// One lane decodes one block.  The same source is compiled twice: by hipcc into the gfx950 kernel (tetra_lmac.hip, the
// accessors hit LDS) and by g++ (-DTETRA_HOST_EMUL) into tests/emul, where it is checked against the reference-built
// primitives without a GPU.  Everything is integer work, so "same source" means "same results".
//
// Restated from the reference (src/decoder/src/lower_mac/), not copied:
//   * scrambler        tetra_scramb.c:34-51 (Fibonacci LFSR, taps 32 26 23 22 16 12 11 10 8 7 5 4 2 1, shifts right, new
//                      bit enters at bit 31) -- here the tap XOR is one AND + parity;
//   * soft mapping     viterbi.c:12-23 (0 -> +127, 0xff -> 0, else -> -127) -- here in units of 127 as a 2-bit signed class;
//   * deinterleaver    tetra_interleave.c:36-39, :51-59: type3[i-1] = type4[(a*i) % K], i = 1..K -- here a running index;
//   * depuncturer      tetra_conv_enc.c:229-251 with punct_2_3 (:131-137, P = {0,1,2,5}, t = 3, period 8): type-3 bit j lands
//                      on mother-code bit 8*((j-1)/3) + P[1 + (j-1)%3] - 1, i.e. bits 0,1 (g1,g2) of every even trellis
//                      step and bit 0 (g1) of every odd one; all others are erasures (metric contribution 0);
//   * decoder          osmo_conv.c: K = 5, N = 4, 16 states, reg bit 3 = newest input bit (:370-392); predecessors of
//                      states i and i+8 are 2i and 2i+1 (:63-95); branch metric m_i = sum_q soft[q] * (1 - 2*g_q) for the
//                      transition 2i --0--> i (:121-133), the other three transitions of the butterfly are -m, -m, +m
//                      because every generator has the D^0 and D^4 terms; ties pick the even predecessor; path metric
//                      of state 0 starts 127*4*5 ahead (:528); CONV_TERM_FLUSH runs K-1 = 4 extra steps (:678-679)
//                      on zero soft bits (viterbi.c:8) and tracebacks from state 0 (:567-612).
//     The reference keeps int16 path metrics and subtracts the minimum every 59 steps (:137-153, :642); with the
//     rate-2/3 depunctured input at most two soft values per step are non-zero, so its sums stay far inside int16 and
//     renormalisation never changes a comparison.  Here the metrics are kept in units of 127 (all soft values are
//     0 or +-127), two per register in packed int16 lanes, without renormalisation: same decisions.
//   * generators       EN 300 392-2 8.2.3.1.1 (tetra_conv_enc.c:45-60 conv_enc_in_bit): g1 = 1+D+D^4, g2 = 1+D^2+D^3+D^4,
//                      g3 = 1+D+D^2+D^4, g4 = 1+D+D^3+D^4.  With i = (d0 d1 d2) the three newer delay bits of the even
//                      predecessor and input 0: g1 = d0, g2 = d1^d2 (g3, g4 only ever meet erasures here).
//   * CRC              crc_simple.c:59-77, :103-106: CRC16-CCITT (0x1021) over the bits, start 0xffff, good = 0x1d0f.
#pragma once

#include <stdint.h>

#include "demux_core.hpp"

#if defined(__HIPCC__) && !defined(TETRA_HOST_EMUL)
#define LM_FN __device__ __forceinline__
#else
#define LM_FN static inline
#endif

namespace tetra_lmac {

// bit (32 - y) for every tap y of the reference's ST(x, y) = x >> (32 - y)
constexpr uint32_t tap_bit(int y) { return 1u << (32 - y); }
constexpr uint32_t kScrambTaps = tap_bit(32) | tap_bit(26) | tap_bit(23) | tap_bit(22) | tap_bit(16) | tap_bit(12) | tap_bit(11) |
                                 tap_bit(10) | tap_bit(8) | tap_bit(7) | tap_bit(5) | tap_bit(4) | tap_bit(2) | tap_bit(1);
constexpr uint32_t kScrambInitSb1 = 3;   // SCRAMB_INIT, tetra_scramb.h:14
constexpr uint32_t kCrcOk = 0x1d0f;      // TETRA_CRC_OK, tetra_common.h:330
constexpr int kFlush = 4;                // K - 1
constexpr int kMaxType345 = 432;
constexpr int kMaxType2 = 288;

LM_FN uint32_t lfsr_next(uint32_t& lfsr) {
    const uint32_t bit = (uint32_t)__builtin_popcount(lfsr & kScrambTaps) & 1u;
    lfsr = (lfsr >> 1) | (bit << 31);
    return bit;
}

// bit-field extracts (one instruction each on the device)
#if defined(__HIPCC__) && !defined(TETRA_HOST_EMUL)
LM_FN uint32_t bfe_u(uint32_t x, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(x, off, width); }
LM_FN uint32_t bfe_mask(uint32_t x, uint32_t off) { return (uint32_t)__builtin_amdgcn_sbfe((int)x, off, 1u); }     // bit (off & 31) -> 0 / 0xffffffff
#else
LM_FN uint32_t bfe_u(uint32_t x, uint32_t off, uint32_t width) { return (x >> off) & ((1u << width) - 1u); }
LM_FN uint32_t bfe_mask(uint32_t x, uint32_t off) { return 0u - ((x >> (off & 31u)) & 1u); }
#endif

// 2-bit signed class of a descrambled byte: +1 (strong 0), 0 (erasure), -1 = 0b11 (strong 1)
LM_FN uint32_t soft_class(uint32_t v) { return v == 0u ? 1u : (v == 0xffu ? 0u : 3u); }

// Descrambles up to 64 bits of one row (a staging chunk) and packs the soft classes 16 per word.  `remaining` = type-5
// bits left in the row from the start of this chunk (a multiple of 4 for every coded block kind); ld4(d) returns bytes
// 4d..4d+3 of the chunk (little endian); st(w, word) receives chunk word w = classes of chunk bits 16w..16w+15 (2 bits
// each, bit 16w+u at bits 2u..2u+1).  Returns the LFSR state for the next chunk.
template <class Ld4, class St>
LM_FN uint32_t descramble_chunk(int remaining, uint32_t lfsr, Ld4 ld4, St st) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (16 * w < remaining) {
            uint32_t word = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bool live = (16 * w + 4 * d) < remaining;
                const uint32_t four = live ? ld4(4 * w + d) : 0u;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t bit = lfsr_next(lfsr);
                    const uint32_t v = ((four >> (8 * b)) & 0xffu) ^ bit;
                    const uint32_t c = live ? soft_class(v) : 0u;
                    word |= c << (2 * (4 * d + b));
                }
            }
            st(w, word);
        }
    }
    return lfsr;
}

// ---- clean rows (every byte 0 or 1: what this library's own demultiplexer writes), round 6 -----------------------------------
// The byte-serial route above costs ~12 vector instructions per type-5 bit (an LFSR step + a three-way classification per byte)
// and is what ANY input needs.  A row of plain bits stays bits: pack the bytes 32 to a word, XOR whole words of the scrambling
// sequence, and let the forward recursion pick its three bits per step pair straight from those words (no classes, no erasures).
// The sequence of a 32-bit code is linear in the code (the LFSR has no constant term), so it is the XOR of four table rows indexed
// by the code's bytes: seq_tab[t][byte][w] = word w of the sequence the code (byte << 8 t) generates (kSeqWords words = 448 bits,
// padded to kSeqStride); the host fills the table once per device with scramb_sequence_table below.
// Bit order: type-5 bit i sits at bit 31 - (i & 31) of word i >> 5 -- the order of the burst synchroniser's packed frames, so that
// a block cut out of a packed frame (decode straight from frames) needs no reversal.
constexpr int kSeqWords = (kMaxType345 + 31) / 32;      // 14
constexpr int kSeqStride = 16;
inline void scramb_sequence_words(uint32_t code, uint32_t* words) {
    uint32_t lfsr = code;
    for (int w = 0; w < kSeqWords; ++w) {
        uint32_t v = 0;
        for (int b = 0; b < 32; ++b) {
            const uint32_t bit = (uint32_t)__builtin_popcount(lfsr & kScrambTaps) & 1u;
            lfsr = (lfsr >> 1) | (bit << 31);
            v |= bit << (31 - b);
        }
        words[w] = v;
    }
}
inline void scramb_sequence_table(uint32_t* tab) {       // [4][256][kSeqStride]
    for (int t = 0; t < 4; ++t)
        for (int b = 0; b < 256; ++b) {
            uint32_t* row = tab + ((size_t)t * 256 + b) * kSeqStride;
            scramb_sequence_words((uint32_t)b << (8 * t), row);
            for (int w = kSeqWords; w < kSeqStride; ++w) row[w] = 0;
        }
}
// four bytes 0 / 1 -> a nibble, byte 0 (the first bit) at bit 3
LM_FN uint32_t pack4(uint32_t v) { return (v * 0x08040201u) >> 24; }

struct U2 { uint32_t x, y; };
// Packs the lane's own row, bytes -> bits; ld8(i) returns bytes 8i..8i+7 of the row.  Returns non-zero if any byte of the row is
// not 0 / 1 (then the byte route has to decode the row).  type345 is a multiple of 8 for every coded block kind.
template <class Ld8>
LM_FN uint32_t pack_row_bits(int type345, Ld8 ld8, uint32_t xb[kSeqWords]) {
    uint32_t dirty = 0;
#pragma unroll
    for (int w = 0; w < kSeqWords; ++w) {
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (32 * w + 8 * q < type345) {
                const U2 d = ld8(4 * w + q);
                dirty |= (d.x | d.y) & 0xfefefefeu;
                v |= ((pack4(d.x) << 4) | pack4(d.y)) << (24 - 8 * q);
            }
        }
        xb[w] = v;
    }
    return dirty;
}
// Descrambles the packed row: seq(t, byte, w) = seq_tab[t][byte][w]; st(w, word) = type-4 bits 32w..32w+31.
template <class Seq, class St>
LM_FN void descramble_bits(int type345, uint32_t code, const uint32_t xb[kSeqWords], Seq seq, St st) {
#pragma unroll
    for (int w = 0; w < kSeqWords; ++w)
        if (32 * w < type345)
            st(w, xb[w] ^ seq(0, code & 0xffu, w) ^ seq(1, (code >> 8) & 0xffu, w) ^ seq(2, (code >> 16) & 0xffu, w) ^ seq(3, code >> 24, w));
}

// ---- blocks straight from the burst synchroniser's packed frames (16 words, first bit most significant) ---------------------------
constexpr int kFrameWords = 16;
// 32 bits of a frame from burst bit s on; bits past the 512th read as zero.  (s is a constant wherever this is called: the block
// kind of a job selects one of a handful of literal layouts, and the loops around it are unrolled.)
LM_FN uint32_t frame_window(const uint32_t fw[kFrameWords], int s) {
    const int w = s >> 5, sh = s & 31;
    const uint32_t hi = w < kFrameWords ? fw[w] : 0u, lo = w + 1 < kFrameWords ? fw[w + 1] : 0u;
    return sh ? (hi << sh) | (lo >> (32 - sh)) : hi;
}
// the type-5 bits of a block = burst bits [off0, off0 + len0) then [off1, off1 + len1) (tetra_burst.c:343-393), as packed words
// (bit i at bit 31 - (i & 31) of word i >> 5), zero behind the block
LM_FN void extract_block(const uint32_t fw[kFrameWords], int off0, int len0, int off1, int len1, uint32_t xb[kSeqWords]) {
    const int total = len0 + len1;
#pragma unroll
    for (int v = 0; v < kSeqWords; ++v) {
        const int start = 32 * v;
        uint32_t x = 0;
        if (start < total) {
            if (start + 32 <= len0) x = frame_window(fw, off0 + start);
            else if (start >= len0) x = frame_window(fw, off1 + start - len0);
            else x = (frame_window(fw, off0 + start) & ~(0xffffffffu >> (len0 - start))) | (frame_window(fw, off1) >> (len0 - start));
            if (total - start < 32) x &= ~(0xffffffffu >> (total - start));
        }
        xb[v] = x;
    }
}

// where a kind's bits sit in its burst (demux_core::pieces_for), as literal layouts so that every shift is a constant
enum { kLayoutSb1 = 0, kLayoutSb2, kLayoutNdb1, kLayoutNdb2, kLayoutSchF, kLayoutBbk, kLayoutNone };
template <int TRAIN, int TPSAP, int BLK>
LM_FN void cut(const uint32_t fw[kFrameWords], int frame_type, uint32_t xb[kSeqWords]) {
    const demux_core::Pieces p = demux_core::pieces_for(TRAIN, TPSAP, BLK);
    extract_block(fw, p.off0, p.len0, p.off1, p.len1, xb);
    if (frame_type != TRAIN) {          // a listed frame of another burst type: the demultiplexer's all-zero row
#pragma unroll
        for (int v = 0; v < kSeqWords; ++v) xb[v] = 0;
    }
}
LM_FN void frame_block(int layout, const uint32_t fw[kFrameWords], int frame_type, uint32_t xb[kSeqWords]) {
    switch (layout) {
        case kLayoutSb1: cut<TETRA_TRAIN_SYNC, TETRA_TPSAP_T_SB1, 1>(fw, frame_type, xb); break;
        case kLayoutSb2: cut<TETRA_TRAIN_SYNC, TETRA_TPSAP_T_SB2, 2>(fw, frame_type, xb); break;
        case kLayoutNdb1: cut<TETRA_TRAIN_NORM_2, TETRA_TPSAP_T_NDB, 1>(fw, frame_type, xb); break;
        case kLayoutNdb2: cut<TETRA_TRAIN_NORM_2, TETRA_TPSAP_T_NDB, 2>(fw, frame_type, xb); break;
        case kLayoutSchF: cut<TETRA_TRAIN_NORM_1, TETRA_TPSAP_T_SCH_F, 0>(fw, frame_type, xb); break;
        default:
#pragma unroll
            for (int v = 0; v < kSeqWords; ++v) xb[v] = 0;
    }
}
// the 30 AACH bits of a frame (SYNC: one piece, NORM_1 / NORM_2: two), first bit most significant; 0 for any other frame type
LM_FN uint32_t bbk_bits(const uint32_t fw[kFrameWords], int frame_type) {
    uint32_t xs[kSeqWords], xn[kSeqWords];
    const demux_core::Pieces ps = demux_core::pieces_for(TETRA_TRAIN_SYNC, TETRA_TPSAP_T_BBK, 0);
    const demux_core::Pieces pn = demux_core::pieces_for(TETRA_TRAIN_NORM_1, TETRA_TPSAP_T_BBK, 0);
    extract_block(fw, ps.off0, ps.len0, ps.off1, ps.len1, xs);
    extract_block(fw, pn.off0, pn.len0, pn.off1, pn.len1, xn);
    return frame_type == TETRA_TRAIN_SYNC ? xs[0] : ((frame_type == TETRA_TRAIN_NORM_1 || frame_type == TETRA_TRAIN_NORM_2) ? xn[0] : 0u);
}
LM_FN uint32_t rev32(uint32_t x) {
#if defined(__HIPCC__) && !defined(TETRA_HOST_EMUL)
    return __builtin_bitreverse32(x);
#else
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i);
    return r;
#endif
}
LM_FN uint32_t spread4(uint32_t nib);
// bits 4k..4k+3 of a first-bit-most-significant word as four bytes, first bit in the low byte
LM_FN uint32_t bbk_bytes(uint32_t y, int k) { return spread4(rev32(y << (4 * k)) & 0xfu); }

// ---- packed 16-bit lanes: two path metrics per 32-bit register (v_pk_add_i16 / v_pk_sub_i16 / v_pk_max_i16) --------
// In units of 127 a path metric never leaves [-2*292, 20 + 2*292], so int16 needs no renormalisation at all.
// pk_lo2 / pk_hi2 / pk_swap feed a packed operation and cost nothing on the device: the compiler folds them into the operation's
// op_sel / op_sel_hi operand modifiers.
#if defined(__HIPCC__) && !defined(TETRA_HOST_EMUL)
typedef short Pk __attribute__((ext_vector_type(2)));
LM_FN Pk pk_make(int lo, int hi) { Pk r; r.x = (short)lo; r.y = (short)hi; return r; }
LM_FN Pk pk_add(Pk a, Pk b) { return a + b; }
LM_FN Pk pk_sub(Pk a, Pk b) { return a - b; }
LM_FN Pk pk_max(Pk a, Pk b) { return __builtin_elementwise_max(a, b); }
LM_FN Pk pk_lo2(Pk x) { return __builtin_shufflevector(x, x, 0, 0); }
LM_FN Pk pk_hi2(Pk x) { return __builtin_shufflevector(x, x, 1, 1); }
LM_FN Pk pk_swap(Pk x) { return __builtin_shufflevector(x, x, 1, 0); }
LM_FN uint32_t pk_bits(Pk a) { return __builtin_bit_cast(uint32_t, a); }
LM_FN Pk pk_from_bits(uint32_t v) { return __builtin_bit_cast(Pk, v); }
// v_perm_b32: byte k of the result = byte sel[k] of the eight bytes { hi (7..4), lo (3..0) }
LM_FN uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// bits of a where keep is set, bits of b elsewhere -- as v_bitop3_b32 (truth table 0xe4), which issues at twice the rate of the
// v_bfi_b32 the compiler picks for the same expression (profiles/r06/r06_o_int_rate.jsonl)
LM_FN uint32_t select_bits(uint32_t a, uint32_t b, uint32_t keep) { return __builtin_amdgcn_bitop3_b32(a, b, keep, 0xe4); }
#else
struct Pk { int16_t x, y; };
LM_FN Pk pk_make(int lo, int hi) { return Pk{ (int16_t)lo, (int16_t)hi }; }
LM_FN Pk pk_add(Pk a, Pk b) { return Pk{ (int16_t)(a.x + b.x), (int16_t)(a.y + b.y) }; }
LM_FN Pk pk_sub(Pk a, Pk b) { return Pk{ (int16_t)(a.x - b.x), (int16_t)(a.y - b.y) }; }
LM_FN Pk pk_max(Pk a, Pk b) { return Pk{ a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y }; }
LM_FN Pk pk_lo2(Pk x) { return Pk{ x.x, x.x }; }
LM_FN Pk pk_hi2(Pk x) { return Pk{ x.y, x.y }; }
LM_FN Pk pk_swap(Pk x) { return Pk{ x.y, x.x }; }
LM_FN uint32_t pk_bits(Pk a) { return (uint32_t)(uint16_t)a.x | ((uint32_t)(uint16_t)a.y << 16); }
LM_FN Pk pk_from_bits(uint32_t v) { return Pk{ (int16_t)(v & 0xffffu), (int16_t)(v >> 16) }; }
LM_FN uint32_t select_bits(uint32_t a, uint32_t b, uint32_t keep) { return (a & keep) | (b & ~keep); }
LM_FN uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t all = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int k = 0; k < 4; ++k) r |= (uint32_t)((all >> (8 * ((sel >> (8 * k)) & 7u))) & 0xffu) << (8 * k);
    return r;
}
#endif

// Path metrics of the 16 states: R[i] = (S[i], S[i + 8]).  Butterfly i has predecessors 2i (even) and 2i + 1 (odd) and produces
// states i and i + 8 -- exactly one register -- from one half each of R[2i & 7] and R[(2i + 1) & 7]: with the halves broadcast by
// operand modifiers a step is 8 x (add, subtract, maximum, difference) and not one move (round 6; until then (S[4k], S[4k+2]) /
// (S[4k+1], S[4k+3]) pairs and eight permutes per step to re-pair the results).
struct PathMetrics { Pk R[8]; };

// One add-compare-select step.  Mm[i] = (m_i, -m_i), m_i = branch metric of the transition 2i --0--> i (the other three transitions
// of the butterfly are -m, -m, +m).  D[i] = (difference into state i, into state i + 8): negative <=> the odd predecessor is
// strictly better (ties keep the even one).
LM_FN void acs_step(PathMetrics& pm, const Pk Mm[8], Pk D[8]) {
    Pk N[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const Pk a = pm.R[(2 * i) & 7], b = pm.R[(2 * i + 1) & 7];
        const Pk e = pk_add(i < 4 ? pk_lo2(a) : pk_hi2(a), Mm[i]);      // S[2i] + m, S[2i] - m
        const Pk o = pk_sub(i < 4 ? pk_lo2(b) : pk_hi2(b), Mm[i]);      // S[2i+1] - m, S[2i+1] + m
        N[i] = pk_max(e, o);
        D[i] = pk_sub(e, o);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) pm.R[i] = N[i];
}

// The 2 x 16 decisions of a step pair as one word: state s of the even step at bit 2 (15 - s) + 1, of the odd step at bit
// 2 (15 - s); 1 = the state took its odd predecessor.  A permute gathers the four sign bytes of D[k] and D[k + 4] (states k, k + 4,
// k + 8, k + 12 -> bytes 3, 2, 1, 0); the eight gathered words are then merged from bit 7 of every byte downwards, a shift and a
// bit-field insert each: 22 instructions per step pair (until round 6 a shift and a masked OR per difference register: 38).
LM_FN uint32_t decision_word(const Pk De[8], const Pk Do[8]) {
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t pe = perm_b32(pk_bits(De[k]), pk_bits(De[k + 4]), 0x05010703u);
        const uint32_t po = perm_b32(pk_bits(Do[k]), pk_bits(Do[k + 4]), 0x05010703u);
        const uint32_t keep_e = 0x01010101u * (0xffu & ~(0xffu >> (2 * k))), keep_o = 0x01010101u * (0xffu & ~(0xffu >> (2 * k + 1)));
        w = select_bits(w, pe >> (2 * k), keep_e);          // bits 7 - 2k and below of every byte from pe (below: overwritten next)
        w = select_bits(w, po >> (2 * k + 1), keep_o);
    }
    return w;
}

// Branch metrics of a step pair: P = (p, -p), Q = (q, -q) for the even step (g1 -> sa, g2 -> sb; butterfly i = (d0 d1 d2):
// sign(g1) = d0, sign(g2) = d1 ^ d2, so with p = sa + sb, q = sa - sb the metrics are m_0..7 = p, q, q, p, -q, -p, -p, -q),
// C = (sc, -sc) for the odd step (g1 -> sc only: m_0..3 = sc, m_4..7 = -sc).
struct Bm { Pk P, Q, C; };
LM_FN Bm bm_from_classes(int sa, int sb, int sc) {
    return Bm{ pk_make(sa + sb, -(sa + sb)), pk_make(sa - sb, sb - sa), pk_make(sc, -sc) };
}
// the same from three plain bits given as masks (0 -> class +1, all ones -> class -1): five + two instructions
LM_FN Bm bm_from_masks(uint32_t ma, uint32_t mb, uint32_t mc) {
    const uint32_t k = 0xfffe0002u;                                     // (2, -2)
    const Pk A = pk_from_bits(ma & k), B = pk_from_bits(mb & k);
    return Bm{ pk_sub(pk_sub(pk_from_bits(k), A), B), pk_sub(B, A), pk_from_bits(0xffff0001u ^ (mc & 0xfffefffeu)) };
}

// the interleaver's running index (a * i) % K, i = 1, 2, ...: returns the current position and advances (a < K for every kind)
LM_FN int interleave_next(int& pos, int a, int K) {
    const int p = pos;
    pos += a;
    pos = pos >= K ? pos - K : pos;
    return p;
}

// Forward recursion over n2 + 4 steps, two steps per round.  fetch() starts the loads of the next step pair's three soft values
// (type-4 bits at three consecutive interleaver positions) and returns what make(raw) needs to turn them into branch metrics;
// st(u, word) receives decision_word of steps 2u, 2u + 1.  The loads of pair u + 1 are issued BEFORE the add-compare-select of pair
// u and consumed after it (software pipelining by one round): a wavefront that has its SIMD to itself -- the last long blocks of a
// launch, every wave of a small one -- does not wait for LDS.  n2 is even for every block kind.
template <class Fetch, class Make, class St>
LM_FN void viterbi_forward(int n2, Fetch fetch, Make make, St st) {
    PathMetrics pm;
#pragma unroll
    for (int i = 0; i < 8; ++i) pm.R[i] = pk_make(0, 0);
    pm.R[0] = pk_make(4 * 5, 0);       // S[0] = 127 * N * K in units of 127
    auto raw = fetch();
    for (int u = 0; u < n2 / 2; ++u) {
        const Bm m = make(raw);
        if (u + 1 < n2 / 2) raw = fetch();
        Pk De[8], Do[8];
        const Pk Me[8] = { m.P, m.Q, m.Q, m.P, pk_swap(m.Q), pk_swap(m.P), pk_swap(m.P), pk_swap(m.Q) };
        acs_step(pm, Me, De);
        const Pk Mo[8] = { m.C, m.C, m.C, m.C, pk_swap(m.C), pk_swap(m.C), pk_swap(m.C), pk_swap(m.C) };
        acs_step(pm, Mo, Do);
        st(u, decision_word(De, Do));
    }
#pragma unroll
    for (int f = 0; f < kFlush / 2; ++f) {
        const Pk z = pk_make(0, 0);
        const Pk Mz[8] = { z, z, z, z, z, z, z, z };
        Pk De[8], Do[8];
        acs_step(pm, Mz, De);
        acs_step(pm, Mz, Do);
        st(n2 / 2 + f, decision_word(De, Do));
    }
}
// what fetch() hands to make(): the three LDS words that hold a step pair's soft values and where in them
struct Raw3 { uint32_t w[3]; uint32_t at[3]; };

// CRC16-CCITT (crc_simple.c:59-77, :103-106: x^16 + x^12 + x^5 + 1, start 0xffff, bits MSB first, good = 0x1d0f over type1 + 16
// bits).  The traceback meets the bits last to first, so it runs the register BACKWARDS from the good value and checks that it
// arrives at the start value: the shift register step is invertible (the polynomial's constant term is 1),
//     forward   fb = msb(r) ^ d;  r = (r << 1) ^ (fb ? 0x1021 : 0)            inverse   fb = r & 1;  r = ((r ^ fb * 0x1021) >> 1) | (fb ^ d) << 15
// and eight inverse steps are one table look-up: r = (r >> 8) ^ Tinv[r & 0xff] ^ (the eight bits, first one most significant, << 8)
// (the data bits enter at bit 15 and only shift right afterwards).  Until round 6 the CRC was folded in bit by bit (an AND + XOR with
// a position constant per bit, three instructions); this is five instructions and a look-up per eight bits.
// The register is carried shifted left by two (r4 = r << 2; whatever lands in its two low bits is never looked at) and the table
// holds Tinv << 2, so that r4 & 0x3fc is the byte offset of the table entry: one AND makes the LDS address.
struct CrcInvTable { uint32_t t[256]; };
constexpr uint32_t crc_inv_step(uint32_t r, uint32_t d) {
    const uint32_t fb = r & 1u;
    return ((r ^ (fb ? 0x1021u : 0u)) >> 1) | ((fb ^ d) << 15);
}
constexpr CrcInvTable make_crc_inv_table() {
    CrcInvTable c{};
    for (uint32_t b = 0; b < 256; ++b) {
        uint32_t r = b;
        for (int k = 0; k < 8; ++k) r = crc_inv_step(r, 0u);
        c.t[b] = r << 2;
    }
    return c;
}

// Traceback from state 0 after the flush steps with the CRC check run backwards alongside.  ld(u) returns the decision word of step
// pair u as the forward pass stored it; st(h, half) receives decoded bits 16h..16h+15 (bit 16h+b at bit b); tinv(off) = the entry of
// make_crc_inv_table at BYTE offset off (per-lane).  n2 = type1 + 16 CRC-covered bits + 4 tail bits, a multiple of 16.  Returns whether the CRC
// over the first n2 - 4 decoded bits is good.
// y = 15 - state runs in a shift register: the predecessor of state s under decision d is (2 s + d) & 15, so y' = 2 y + 1 - d, and
// the decision of state s sits at bit 2 y (odd step) or 2 y + 1 (even step) of the word: a shift, a one-bit signed extract (= -d)
// and an add per step.  The decoded bit of a step is the complement of bit 3 of y before the step, and the register only ever
// shifts, so after the 16 steps of group h bits 4..31 of ~y are the decoded bits 16h .. 16h+27: the CRC bytes, which end 4 bits
// before a group boundary, are cut from there -- bits 16h+12..16h+19 (not in the top group: tail bits) and 16h+4..16h+11; the four
// bits 0..3 left at the end take four single steps.
template <class Ld, class St, class Tinv>
LM_FN bool viterbi_traceback(int n2, Ld ld, St st, Tinv tinv) {
    uint32_t y = 15u;
#pragma unroll
    for (int f = kFlush / 2 - 1; f >= 0; --f) {
        const uint32_t w = ld(n2 / 2 + f);
        uint32_t t = y << 1;
        y = t + 1u + bfe_mask(w, t);
        t = (y << 1) | 1u;
        y = t + bfe_mask(w, t);
    }
    uint32_t r4 = kCrcOk << 2;
    const int top = n2 / 16 - 1;
    // the eight decision words of a group are requested one group ahead of their use (the scratch is in L2 / HBM: ~2000 clocks)
    uint32_t cur[8], nxt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int b2 = 0; b2 < 8; ++b2) cur[b2] = ld(8 * top + b2);
    for (int h = top; h >= 0; --h) {
        if (h > 0) {
#pragma unroll
            for (int b2 = 0; b2 < 8; ++b2) nxt[b2] = ld(8 * (h - 1) + b2);
        }
#pragma unroll
        for (int b2 = 7; b2 >= 0; --b2) {
            const uint32_t w = cur[b2];
            uint32_t t = y << 1;
            y = t + 1u + bfe_mask(w, t);
            t = (y << 1) | 1u;
            y = t + bfe_mask(w, t);
        }
        const uint32_t bits = ~y;
        st(h, (bits >> 4) & 0xffffu);
        const uint32_t rev = rev32(bits);                 // decoded bit 16h + k at bit 27 - k
        if (h != top) r4 = (r4 >> 8) ^ tinv(r4 & 0x3fcu) ^ ((rev << 2) & 0x3fc00u);          // bits 16h+12 .. 16h+19
        r4 = (r4 >> 8) ^ tinv(r4 & 0x3fcu) ^ ((rev >> 6) & 0x3fc00u);                        // bits 16h+4 .. 16h+11
#pragma unroll
        for (int b2 = 0; b2 < 8; ++b2) cur[b2] = nxt[b2];
    }
    uint32_t r = (r4 >> 2) & 0xffffu;
#pragma unroll
    for (int k = 3; k >= 0; --k) r = crc_inv_step(r, (~y >> (4 + k)) & 1u);                         // bits 3 .. 0
    return r == 0xffffu;
}

// 4 decoded bits -> 4 bytes (one bit per byte, little endian)
LM_FN uint32_t spread4(uint32_t nib) { return (nib * 0x00204081u) & 0x01010101u; }

}  // namespace tetra_lmac
