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

// Forward recursion over n2 + 4 steps.  cls(idx) returns the soft class (-1, 0, +1) of type-4 bit idx (0-based);
// st(t, mask) receives the decision mask of step t (state s at bit 15 - s).
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
        {   // even step: g1 -> sa, g2 -> sb.  butterfly i = (d0 d1 d2): sign(g1) = d0, sign(g2) = d1 ^ d2, so with
            // p = sa + sb, q = sa - sb the metrics are m_0..7 = p, q, q, p, -q, -p, -p, -q
            const Pk pq = pk_make(sa + sb, sa - sb), qp = pk_swap(pq);
            st(2 * u, acs_pk(pm, pq, qp, pk_neg(qp), pk_neg(pq)));
        }
        {   // odd step: g1 -> sc only: m_0..3 = sc, m_4..7 = -sc
            const Pk cc = pk_make(sc, sc), nc = pk_neg(cc);
            st(2 * u + 1, acs_pk(pm, cc, cc, nc, nc));
        }
    }
#pragma unroll
    for (int f = 0; f < kFlush; ++f) {
        const Pk z = pk_make(0, 0);
        st(n2 + f, acs_pk(pm, z, z, z, z));
    }
}

// Traceback from state 0 after the flush steps.  ld(t) returns the decision mask of step t (state s at bit 15 - s);
// st(h, half) receives decoded bits 16h..16h+15 (bit 16h+b at bit b).  n2 is a multiple of 16 for every coded block kind.
template <class Ld, class St>
LM_FN void viterbi_traceback(int n2, Ld ld, St st) {
    uint32_t state = 0;
#pragma unroll
    for (int f = kFlush - 1; f >= 0; --f) state = ((state << 1) & 0xfu) | ((ld(n2 + f) >> (state ^ 15u)) & 1u);
    for (int h = n2 / 16 - 1; h >= 0; --h) {
        uint32_t half = 0;
#pragma unroll
        for (int b = 15; b >= 0; --b) {
            half |= (state >> 3) << b;                                        // vals[state]: the newest input bit
            state = ((state << 1) & 0xfu) | ((ld(16 * h + b) >> (state ^ 15u)) & 1u);
        }
        st(h, half);
    }
}

// CRC16-CCITT over decoded bits 0..nbits-1; ld(h) returns bits 16h..16h+15.
template <class Ld>
LM_FN uint32_t crc16_bits(int nbits, Ld ld) {
    uint32_t crc = 0xffffu;
    for (int h = 0; h * 16 < nbits; ++h) {
        const uint32_t half = ld(h);
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            if (16 * h + b < nbits) {
                crc ^= ((half >> b) & 1u) << 15;
                crc = (crc & 0x8000u) ? (((crc << 1) ^ 0x1021u) & 0xffffu) : ((crc << 1) & 0xffffu);
            }
        }
    }
    return crc;
}

// 4 decoded bits -> 4 bytes (one bit per byte, little endian)
LM_FN uint32_t spread4(uint32_t nib) { return (nib * 0x00204081u) & 0x01010101u; }

}  // namespace tetra_lmac
