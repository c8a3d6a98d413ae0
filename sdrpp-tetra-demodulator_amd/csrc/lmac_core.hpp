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
LM_FN uint32_t bfe_mask(uint32_t x, uint32_t off) { return (uint32_t)__builtin_amdgcn_sbfe((int)x, off, 1u); }     // bit -> 0 / 0xffffffff
#else
LM_FN uint32_t bfe_u(uint32_t x, uint32_t off, uint32_t width) { return (x >> off) & ((1u << width) - 1u); }
LM_FN uint32_t bfe_mask(uint32_t x, uint32_t off) { return 0u - ((x >> off) & 1u); }
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
// and is what ANY input needs.  A row of plain bits takes another route: pack the bytes 32 to a word, XOR whole words of the
// scrambling sequence, and spread the bits to classes (+1 / -1, never an erasure) -- ~2 instructions per bit.  The sequence of a
// 32-bit code is linear in the code (the LFSR has no constant term), so it is the XOR of four table rows indexed by the code's
// bytes: seq_tab[t][byte][w] = word w of the sequence the code (byte << 8 t) generates (kSeqWords words = 448 bits, padded to
// kSeqStride); the host fills the table once per device with scramb_sequence_words below.
constexpr int kSeqWords = (kMaxType345 + 31) / 32;      // 14
constexpr int kSeqStride = 16;
// bit i of the sequence = bit (i & 31) of word i >> 5
inline void scramb_sequence_words(uint32_t code, uint32_t* words) {
    uint32_t lfsr = code;
    for (int w = 0; w < kSeqWords; ++w) {
        uint32_t v = 0;
        for (int b = 0; b < 32; ++b) {
            const uint32_t bit = (uint32_t)__builtin_popcount(lfsr & kScrambTaps) & 1u;
            lfsr = (lfsr >> 1) | (bit << 31);
            v |= bit << b;
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
// four bytes 0 / 1 -> a nibble, byte k at bit k
LM_FN uint32_t pack4(uint32_t v) { return (v * 0x01020408u) >> 24; }
// the low 16 bits of x, bit k moved to bit 2 k
LM_FN uint32_t spread16(uint32_t x) {
    x &= 0xffffu;
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
// 16 descrambled bits -> their class word (2 bits each, bit u at 2u..2u+1): 0 -> +1 = 0b01, 1 -> -1 = 0b11; positions at and beyond
// `live` (bits of the row left from this word's first) get class 0 like the byte route gives them
LM_FN uint32_t class_word(uint32_t bits16, int live) {
    const uint32_t w = (spread16(bits16) << 1) | 0x55555555u;
    return live >= 16 ? w : (live <= 0 ? 0u : (w & ((1u << (2 * live)) - 1u)));
}

struct U2 { uint32_t x, y; };
// Packs the lane's own row, bytes -> bits (type-5 bit i at bit i & 31 of xb[i >> 5]); ld8(i) returns bytes 8i..8i+7 of the row.
// Returns non-zero if any byte of the row is not 0 / 1 (then the byte route has to decode the row).  type345 is a multiple of 8
// for every coded block kind.
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
                v |= (pack4(d.x) | (pack4(d.y) << 4)) << (8 * q);
            }
        }
        xb[w] = v;
    }
    return dirty;
}
// Descrambles the packed row and stores its class words: seq(t, byte, w) = seq_tab[t][byte][w]; st(i, word) = class word i
// (classes of type-4 bits 16i..16i+15), the same words descramble_chunk produces for a row of plain bits.
template <class Seq, class St>
LM_FN void classes_from_bits(int type345, uint32_t code, const uint32_t xb[kSeqWords], Seq seq, St st) {
#pragma unroll
    for (int w = 0; w < kSeqWords; ++w) {
        if (32 * w < type345) {
            const uint32_t sw = seq(0, code & 0xffu, w) ^ seq(1, (code >> 8) & 0xffu, w) ^ seq(2, (code >> 16) & 0xffu, w) ^ seq(3, code >> 24, w);
            const uint32_t x = xb[w] ^ sw;
            const int live = type345 - 32 * w;
            st(2 * w, class_word(x, live));
            if (live > 16) st(2 * w + 1, class_word(x >> 16, live - 16));
        }
    }
}

// ---- packed 16-bit lanes: two path metrics per 32-bit register (v_pk_add_i16 / v_pk_sub_i16 / v_pk_max_i16) --------
// In units of 127 a path metric never leaves [-2*292, 20 + 2*292], so int16 needs no renormalisation at all.
#if defined(__HIPCC__) && !defined(TETRA_HOST_EMUL)
typedef short Pk __attribute__((ext_vector_type(2)));
LM_FN Pk pk_make(int lo, int hi) { Pk r; r.x = (short)lo; r.y = (short)hi; return r; }
LM_FN Pk pk_add(Pk a, Pk b) { return a + b; }
LM_FN Pk pk_sub(Pk a, Pk b) { return a - b; }
LM_FN Pk pk_max(Pk a, Pk b) { return __builtin_elementwise_max(a, b); }
LM_FN Pk pk_lolo(Pk x, Pk y) { return __builtin_shufflevector(x, y, 0, 2); }
LM_FN Pk pk_hihi(Pk x, Pk y) { return __builtin_shufflevector(x, y, 1, 3); }
LM_FN Pk pk_swap(Pk x) { return __builtin_shufflevector(x, x, 1, 0); }
LM_FN uint32_t pk_bits(Pk a) { return __builtin_bit_cast(uint32_t, a); }
#else
struct Pk { int16_t x, y; };
LM_FN Pk pk_make(int lo, int hi) { return Pk{ (int16_t)lo, (int16_t)hi }; }
LM_FN Pk pk_add(Pk a, Pk b) { return Pk{ (int16_t)(a.x + b.x), (int16_t)(a.y + b.y) }; }
LM_FN Pk pk_sub(Pk a, Pk b) { return Pk{ (int16_t)(a.x - b.x), (int16_t)(a.y - b.y) }; }
LM_FN Pk pk_max(Pk a, Pk b) { return Pk{ a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y }; }
LM_FN Pk pk_lolo(Pk x, Pk y) { return Pk{ x.x, y.x }; }
LM_FN Pk pk_hihi(Pk x, Pk y) { return Pk{ x.y, y.y }; }
LM_FN Pk pk_swap(Pk x) { return Pk{ x.y, x.x }; }
LM_FN uint32_t pk_bits(Pk a) { return (uint32_t)(uint16_t)a.x | ((uint32_t)(uint16_t)a.y << 16); }
#endif

// Path metrics of the 16 states, packed for the butterflies: butterfly j has predecessors 2j (even) and 2j+1 (odd) and
// produces states j and j+8.  E[k] = (S[4k], S[4k+2]) and O[k] = (S[4k+1], S[4k+3]) are the even / odd predecessors of
// butterflies 2k (low half) and 2k+1 (high half).
struct PathMetrics { Pk E[4], O[4]; };

// One add-compare-select step.  Mk = (m_2k, m_2k+1) are the branch metrics of butterflies 2k and 2k+1.  Returns the 16
// decision bits, state s at bit 15 - s (1 = state s took its odd predecessor).
LM_FN uint32_t acs_pk(PathMetrics& pm, Pk M0, Pk M1, Pk M2, Pk M3) {
    const Pk M[4] = { M0, M1, M2, M3 };
    Pk NN[8];      // NN[j] = (S'[2j], S'[2j+1])
    Pk D[8];       // sign bits = decisions of states (2r, 2r+1)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const Pk e0 = pk_add(pm.E[k], M[k]), o0 = pk_sub(pm.O[k], M[k]);   // into states 2k, 2k+1
        const Pk e1 = pk_sub(pm.E[k], M[k]), o1 = pk_add(pm.O[k], M[k]);   // into states 2k+8, 2k+9
        NN[k] = pk_max(e0, o0);
        NN[k + 4] = pk_max(e1, o1);
        D[k] = pk_sub(e0, o0);          // negative <=> odd predecessor strictly better (ties keep the even one)
        D[k + 4] = pk_sub(e1, o1);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int r = 7; r >= 0; --r) acc = (acc >> 2) | (pk_bits(D[r]) & 0x80008000u);
    // low-half sign of D[r] (state 2r) now sits at bit 15 - 2r, high-half sign (state 2r+1) at bit 31 - 2r
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        pm.E[k] = pk_lolo(NN[2 * k], NN[2 * k + 1]);
        pm.O[k] = pk_hihi(NN[2 * k], NN[2 * k + 1]);
    }
    return (acc & 0xaaaau) | ((acc >> 17) & 0x5555u);
}

LM_FN Pk pk_neg(Pk a) { return pk_sub(pk_make(0, 0), a); }

// Forward recursion over n2 + 4 steps, two steps per call-back (round 6: one 32-bit store per step pair instead of two 16-bit ones).
// cls(idx) returns the soft class (-1, 0, +1) of type-4 bit idx (0-based); st(u, word) receives the decisions of steps 2u (low
// half) and 2u + 1 (high half), INVERTED: state s at bit 15 - s of its half, 1 = the state kept its EVEN predecessor -- the form
// the traceback consumes without a complement.  n2 is even for every block kind.
template <class Cls, class St>
LM_FN void viterbi_forward(int n2, int K, int a, Cls cls, St st) {
    PathMetrics pm;
#pragma unroll
    for (int k = 0; k < 4; ++k) { pm.E[k] = pk_make(0, 0); pm.O[k] = pk_make(0, 0); }
    pm.E[0] = pk_make(4 * 5, 0);       // S[0] = 127 * N * K in units of 127
    int pos = a;                       // (a * i) % K for i = 1 (a < K for every block kind)
    for (int u = 0; u < n2 / 2; ++u) {
        const int sa = cls(pos);
        pos += a; pos = pos >= K ? pos - K : pos;
        const int sb = cls(pos);
        pos += a; pos = pos >= K ? pos - K : pos;
        const int sc = cls(pos);
        pos += a; pos = pos >= K ? pos - K : pos;
        // even step: g1 -> sa, g2 -> sb.  butterfly i = (d0 d1 d2): sign(g1) = d0, sign(g2) = d1 ^ d2, so with
        // p = sa + sb, q = sa - sb the metrics are m_0..7 = p, q, q, p, -q, -p, -p, -q
        const Pk pq = pk_make(sa + sb, sa - sb), qp = pk_swap(pq);
        const uint32_t even = acs_pk(pm, pq, qp, pk_neg(qp), pk_neg(pq));
        // odd step: g1 -> sc only: m_0..3 = sc, m_4..7 = -sc
        const Pk cc = pk_make(sc, sc), nc = pk_neg(cc);
        const uint32_t odd = acs_pk(pm, cc, cc, nc, nc);
        st(u, (even | (odd << 16)) ^ 0xffffffffu);
    }
#pragma unroll
    for (int f = 0; f < kFlush / 2; ++f) {
        const Pk z = pk_make(0, 0);
        const uint32_t even = acs_pk(pm, z, z, z, z);
        const uint32_t odd = acs_pk(pm, z, z, z, z);
        st(n2 / 2 + f, (even | (odd << 16)) ^ 0xffffffffu);
    }
}

// CRC16-CCITT (crc_simple.c:59-77, :103-106: x^16 + x^12 + x^5 + 1, start 0xffff, bits MSB first) is affine in the message: the
// register after n bits = Z(n) ^ XOR over the 1 bits of T[distance of the bit from the end], T[j] = x^(16 + j) mod the polynomial,
// Z(n) = the register after n zero bits.  The traceback meets the bits last to first, so it folds the CRC in: one AND + XOR per bit
// with a wave-uniform constant instead of a five-instruction shift register step.
constexpr int kCrcPad = 4;            // leading zero entries: the 4 tail bits behind the CRC field meet a zero constant, no branch
struct CrcTable {
    uint32_t t[kCrcPad + kMaxType2];   // t[kCrcPad + j] = T[j] (dwords: a wave-uniform index then loads through the scalar cache)
};
constexpr CrcTable make_crc_table() {
    CrcTable c{};
    uint32_t v = 0x1021u;                                  // one 1 bit into a zero register
    for (int j = 0; j < kMaxType2; ++j) {
        c.t[kCrcPad + j] = v;
        v = (v & 0x8000u) ? (((v << 1) ^ 0x1021u) & 0xffffu) : ((v << 1) & 0xffffu);
    }
    return c;
}
// Z(n) ^ XOR of T[0 .. n - 1]: what the traceback's accumulator (XOR of T[j] over the ZERO bits) has to be XORed with
constexpr uint32_t crc_fold_constant(const CrcTable& c, int n) {
    uint32_t z = 0xffffu;
    for (int i = 0; i < n; ++i) z = (z & 0x8000u) ? (((z << 1) ^ 0x1021u) & 0xffffu) : ((z << 1) & 0xffffu);
    for (int j = 0; j < n; ++j) z ^= c.t[kCrcPad + j];
    return z;
}

// Traceback from state 0 after the flush steps, CRC folded in.  ld(u) returns the decision word of step pair u as the forward pass
// stored it; st(h, half) receives decoded bits 16h..16h+15 (bit 16h+b at bit b); tbl(k) = CrcTable::t[k] (wave-uniform index);
// n_crc = type1 + 16 bits are covered by the CRC (n2 = n_crc + 4 tail bits), fold = crc_fold_constant(n_crc).  Returns the CRC
// register.  n2 is a multiple of 16.
// y = 15 - state runs in a shift register: y' = (y << 1 | inverted decision) & 15, and the decoded bit of a step is the complement
// of bit 3 of y before the step, so after 16 steps the 16 decoded bits sit, complemented, at bits 4..19.
template <class Ld, class St, class Tbl>
LM_FN uint32_t viterbi_traceback(int n2, int n_crc, uint32_t fold, Ld ld, St st, Tbl tbl) {
    uint32_t y = 15u;
#pragma unroll
    for (int f = kFlush / 2 - 1; f >= 0; --f) {
        const uint32_t w = ld(n2 / 2 + f);
        y = (y << 1) | bfe_u(w, (y & 15u) | 16u, 1);
        y = (y << 1) | bfe_u(w, y & 15u, 1);
    }
    uint32_t acc = 0;
    for (int h = n2 / 16 - 1; h >= 0; --h) {
        const int k0 = n_crc - 1 + kCrcPad - 16 * h;          // table index of bit 16 h (>= 0 for every bit of the row)
#pragma unroll
        for (int b2 = 7; b2 >= 0; --b2) {
            const uint32_t w = ld(8 * h + b2);
            // step 16h + 2 b2 + 1, then step 16h + 2 b2: y bit 3 set <=> the decoded bit is 0
            acc ^= tbl(k0 - 2 * b2 - 1) & bfe_mask(y, 3);
            y = (y << 1) | bfe_u(w, (y & 15u) | 16u, 1);
            acc ^= tbl(k0 - 2 * b2) & bfe_mask(y, 3);
            y = (y << 1) | bfe_u(w, y & 15u, 1);
        }
        st(h, (~y >> 4) & 0xffffu);
    }
    return (acc ^ fold) & 0xffffu;
}

// 4 decoded bits -> 4 bytes (one bit per byte, little endian)
LM_FN uint32_t spread4(uint32_t nib) { return (nib * 0x00204081u) & 0x01010101u; }

}  // namespace tetra_lmac
