// demod_core.hpp -- per-lane arithmetic of the batched TETRA pi/4-DQPSK demodulator kernels.
//
// One source, two targets:
//   * device (hipcc, gfx950): V = float, one value per lane, cross-lane moves are DPP
//     row shifts inside a 16-lane row;
//   * host emulation (-DTETRA_HOST_EMUL, g++): V = Row16, a 16-lane row stepped in lockstep, with
//     the DPP moves emulated.  tests/ use it to check the systolic schedule against the CPU
//     oracle without a GPU.  It is NOT a fallback: the library has no CPU path.
//
// Arithmetic contract (must stay bit-identical to oracle/tetra_oracle.c, which restates the
// reference): binary32, no contraction (-ffp-contract=off), explicit fma only inside dot-product
// chains and inside sincos, loop arithmetic as separate mul/add, correctly rounded sqrt.
//
// Reference citations (cropinghigh/sdrpp-tetra-demodulator):
//   AGC      SDR++ core loop::FastAGC::process, called at src/dsp/pi4dqpsk.cpp:134
//   FLL      src/dsp/fll.cpp:135-149
//   RRC      SDR++ core filter::FIR<complex_t,float>::process, called at src/dsp/pi4dqpsk.cpp:136
//   timing   src/dsp/complex_fd.cpp:89-151
//   Costas   src/dsp/pi4dqpsk_costas.cpp:5-28
//   slicer   src/dsp/dqpsk_sym_extr.cpp:6-7,32-52 ; unpacker src/dsp/bit_unpacker.cpp:4-10
#pragma once

#include <stdint.h>
#include <type_traits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TD_FN __device__ __forceinline__
#define TD_MFN __device__ __forceinline__
#define TD_DEVICE 1
#else
#include <math.h>
// host builds: plain inline (forcing inlining of the unrolled row program makes g++ take minutes)
#define TD_FN static inline
#define TD_MFN inline
#define TD_DEVICE 0
#endif

namespace tdm {

constexpr float kFlPi = 3.1415926535f;       // SDR++ core FL_M_PI
constexpr int kPadTaps = 80;                  // table capacity of the C ABI (get_tables, channel_state.hist); filters are <= 72 taps
constexpr int kHist = kPadTaps;               // stored delay-line samples per channel
constexpr int kInterpPhases = 128;
constexpr int kInterpTaps = 8;

// ---------------------------------------------------------------------------------------------
// float backend (device lanes, and host scalar code)
// ---------------------------------------------------------------------------------------------
TD_FN float v_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
TD_FN float v_sqrt(float a) { return __builtin_sqrtf(a); }
TD_FN float v_rint(float a) { return __builtin_rintf(a); }
TD_FN float v_floor(float a) { return __builtin_floorf(a); }
TD_FN float v_abs(float a) { return __builtin_fabsf(a); }
// clamp to [lo, hi] (lo <= hi): `x > hi ? hi : (x < lo ? lo : x)`; one v_med3_f32 on the device
#if TD_DEVICE
TD_FN float v_clamp(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
#else
TD_FN float v_clamp(float x, float lo, float hi) { return x > hi ? hi : (x < lo ? lo : x); }
#endif
TD_FN float v_max(float a, float b) { return __builtin_fmaxf(a, b); }
TD_FN float v_min(float a, float b) { return __builtin_fminf(a, b); }
TD_FN float v_sel(bool m, float a, float b) { return m ? a : b; }
TD_FN int v_sel(bool m, int a, int b) { return m ? a : b; }
TD_FN int v_ftoi(float a) { return (int)a; }
TD_FN bool v_ieq(int a, int b) { return a == b; }
TD_FN int v_iand(int a, int b) { return a & b; }
// x with its sign flipped when k is odd
TD_FN float v_flip_if_odd(float x, int k) {
    return __builtin_bit_cast(float, __builtin_bit_cast(int, x) ^ (int)((unsigned)k << 31));
}
// x with its sign flipped when the float k (an integer value with |k| <= 2) is odd: the exponent field of +-1.0f has its
// lowest bit set and that of 0.0f / +-2.0f has not, so `bits(k) << 8` is exactly the sign mask -- one shift instead of
// a convert and a shift.  Identical to v_flip_if_odd(x, (int)k) for |k| <= 2.
TD_FN float v_flip_by_k8(float x, float k) {
    return __builtin_bit_cast(float, __builtin_bit_cast(int, x) ^ (int)(__builtin_bit_cast(unsigned, k) << 8));
}
// Keep a loop-invariant value in a vector register (device): the constant-bus limit lets a VOP3 instruction read only one
// scalar register, and without this the compiler re-materialises the second scalar operand of v_med3 with a v_mov per use.
#if TD_DEVICE
TD_FN float v_pin(float x) { asm volatile("" : "+v"(x)); return x; }
#else
TD_FN float v_pin(float x) { return x; }
#endif
// |x| > lim ? x - copysign(delta, x) : x   -- the phase wrap of PhaseControlLoop for symmetric limits
TD_FN float v_wrap_sym(float x, float lim, float delta) {
    const float t = x - __builtin_copysignf(delta, x);
    return __builtin_fabsf(x) > lim ? t : x;
}

template <class V> struct vtraits;
template <> struct vtraits<float> { using M = bool; using I = int; };

#if TD_DEVICE
typedef float pk2 __attribute__((ext_vector_type(2)));
// Pair<float> on the device is a 64-bit register pair so that pk_fma lowers to v_pk_fma_f32.
template <class V> struct Pair;
template <> struct Pair<float> {
    pk2 v;
    TD_MFN Pair() {}
    TD_MFN Pair(float x, float y) { v.x = x; v.y = y; }
    TD_MFN float x() const { return v.x; }
    TD_MFN float y() const { return v.y; }
};
TD_FN Pair<float> pk_fma(Pair<float> a, Pair<float> b, Pair<float> c) {
    Pair<float> r;
    r.v = __builtin_elementwise_fma(a.v, b.v, c.v);
    return r;
}
TD_FN Pair<float> pk_mul(Pair<float> a, Pair<float> b) { Pair<float> r; r.v = a.v * b.v; return r; }
TD_FN Pair<float> pk_add(Pair<float> a, Pair<float> b) { Pair<float> r; r.v = a.v + b.v; return r; }
TD_FN Pair<float> pk_sub(Pair<float> a, Pair<float> b) { Pair<float> r; r.v = a.v - b.v; return r; }
TD_FN Pair<float> pk_swap(Pair<float> a) { Pair<float> r; r.v = __builtin_shufflevector(a.v, a.v, 1, 0); return r; }
// DPP row moves (gfx9 DPP controls): row_shr:1 = 0x111 (lane l <- lane l-1), row_shl:1 = 0x101
// (lane l <- lane l+1).  With bound_ctrl off, lanes whose source is outside the 16-lane row keep `old`.
TD_FN float row_shr1(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, src), 0x111, 0xf, 0xf, false));
}
TD_FN float row_shl1(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, src), 0x101, 0xf, 0xf, false));
}
// Two-lane variants for rows that interleave two channels on even/odd lanes (8 lanes per channel).
TD_FN float row_shr2(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, src), 0x112, 0xf, 0xf, false));
}
TD_FN float row_shl2(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, src), 0x102, 0xf, 0xf, false));
}
TD_FN float row_shl2_z(float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), 0x102, 0xf, 0xf, true));
}
// H-lane variants (H = 2: two channels interleaved on the lanes of a row, 8 lanes per channel; H = 4: four channels, 4 lanes each)
template <int H> TD_FN float row_shr_h(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, src), 0x110 + H, 0xf, 0xf, false));
}
template <int H> TD_FN float row_shl_h(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, src), 0x100 + H, 0xf, 0xf, false));
}
template <int H> TD_FN float row_shl_h_z(float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), 0x100 + H, 0xf, 0xf, true));
}
// row_shl:1 with zero fill (bound_ctrl): lane 15 of each row receives +0.
TD_FN float row_shl1_z(float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), 0x101, 0xf, 0xf, true));
}
// AGC amplitude square root: hardware v_sqrt_f32 (<= 1 ulp) + the standard two-fma correction = correctly
// rounded for every normal input and for 0.  Inputs below 2^-96 (|out| < 3.5e-15, where hipcc's generic
// expansion rescales) may come out inexact, which cannot change the AGC: any amp < 2^-25 gives
// (setPoint - amp) == 1.0f exactly.  Saves 7 VALU ops per sample over __builtin_sqrtf.
TD_FN float v_sqrt_agc(float x) {
    float y = __builtin_amdgcn_sqrtf(x);
    float yd = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1);
    float yu = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    float rd = __builtin_fmaf(-yd, y, x);
    float ru = __builtin_fmaf(-yu, y, x);
    const bool down = rd <= 0.0f, up = ru > 0.0f;     // never both: both compares first, then both selects (hazard slots overlap)
    y = down ? yd : y;
    y = up ? yu : y;
    return y;
}
#else
TD_FN float v_sqrt_agc(float x) { return __builtin_sqrtf(x); }
template <class V> struct Pair {
    V vx, vy;
    Pair() {}
    Pair(V x, V y) : vx(x), vy(y) {}
    V x() const { return vx; }
    V y() const { return vy; }
};
template <class V> TD_FN Pair<V> pk_fma(Pair<V> a, Pair<V> b, Pair<V> c) {
    return Pair<V>(v_fma(a.x(), b.x(), c.x()), v_fma(a.y(), b.y(), c.y()));
}
template <class V> TD_FN Pair<V> pk_mul(Pair<V> a, Pair<V> b) { return Pair<V>(a.x() * b.x(), a.y() * b.y()); }
template <class V> TD_FN Pair<V> pk_add(Pair<V> a, Pair<V> b) { return Pair<V>(a.x() + b.x(), a.y() + b.y()); }
template <class V> TD_FN Pair<V> pk_sub(Pair<V> a, Pair<V> b) { return Pair<V>(a.x() - b.x(), a.y() - b.y()); }
template <class V> TD_FN Pair<V> pk_swap(Pair<V> a) { return Pair<V>(a.y(), a.x()); }
#endif

#if defined(TETRA_HOST_EMUL)
// ---------------------------------------------------------------------------------------------
// Row16 backend: a 16-lane row in lockstep (host emulation of one DPP row).
// ---------------------------------------------------------------------------------------------
struct Row16m { bool l[16]; };
struct Row16i { int l[16]; };
struct Row16 {
    float l[16];
    Row16() {}
    Row16(float s) { for (int i = 0; i < 16; i++) l[i] = s; }
};
#define TD_R16_BIN(op) \
    TD_FN Row16 operator op(Row16 a, Row16 b) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = a.l[i] op b.l[i]; return r; } \
    TD_FN Row16 operator op(Row16 a, float b) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = a.l[i] op b; return r; } \
    TD_FN Row16 operator op(float a, Row16 b) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = a op b.l[i]; return r; }
TD_R16_BIN(+) TD_R16_BIN(-) TD_R16_BIN(*)
#undef TD_R16_BIN
#define TD_R16_CMP(op) \
    TD_FN Row16m operator op(Row16 a, Row16 b) { Row16m r; for (int i = 0; i < 16; i++) r.l[i] = a.l[i] op b.l[i]; return r; } \
    TD_FN Row16m operator op(Row16 a, float b) { Row16m r; for (int i = 0; i < 16; i++) r.l[i] = a.l[i] op b; return r; }
TD_R16_CMP(>) TD_R16_CMP(<) TD_R16_CMP(>=) TD_R16_CMP(<=)
#undef TD_R16_CMP
TD_FN Row16 operator-(Row16 a) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = -a.l[i]; return r; }
TD_FN Row16 v_fma(Row16 a, Row16 b, Row16 c) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = fmaf(a.l[i], b.l[i], c.l[i]); return r; }
TD_FN Row16 v_sqrt(Row16 a) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = sqrtf(a.l[i]); return r; }
TD_FN Row16 v_rint(Row16 a) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = rintf(a.l[i]); return r; }
TD_FN Row16 v_abs(Row16 a) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = fabsf(a.l[i]); return r; }
TD_FN Row16 v_max(Row16 a, Row16 b) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = fmaxf(a.l[i], b.l[i]); return r; }
TD_FN Row16 v_min(Row16 a, Row16 b) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = fminf(a.l[i], b.l[i]); return r; }
TD_FN Row16 v_clamp(Row16 x, float lo, float hi) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = x.l[i] > hi ? hi : (x.l[i] < lo ? lo : x.l[i]); return r; }
TD_FN Row16 v_sel(Row16m m, Row16 a, Row16 b) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = m.l[i] ? a.l[i] : b.l[i]; return r; }
TD_FN Row16i v_ftoi(Row16 a) { Row16i r; for (int i = 0; i < 16; i++) r.l[i] = (int)a.l[i]; return r; }
TD_FN Row16m v_ieq(Row16i a, int b) { Row16m r; for (int i = 0; i < 16; i++) r.l[i] = a.l[i] == b; return r; }
TD_FN Row16 v_flip_if_odd(Row16 x, Row16i k) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = v_flip_if_odd(x.l[i], k.l[i]); return r; }
TD_FN Row16 v_flip_by_k8(Row16 x, Row16 k) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = v_flip_by_k8(x.l[i], k.l[i]); return r; }
TD_FN Row16 v_wrap_sym(Row16 x, float lim, float delta) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = v_wrap_sym(x.l[i], lim, delta); return r; }
TD_FN Row16i v_iand(Row16i a, int b) { Row16i r; for (int i = 0; i < 16; i++) r.l[i] = a.l[i] & b; return r; }
template <> struct vtraits<Row16> { using M = Row16m; using I = Row16i; };
TD_FN Row16 row_shr1(Row16 old, Row16 src) { Row16 r; r.l[0] = old.l[0]; for (int i = 1; i < 16; i++) r.l[i] = src.l[i - 1]; return r; }
TD_FN Row16 row_shl1(Row16 old, Row16 src) { Row16 r; r.l[15] = old.l[15]; for (int i = 0; i < 15; i++) r.l[i] = src.l[i + 1]; return r; }
TD_FN Row16 row_shl1_z(Row16 src) { Row16 r; r.l[15] = 0.0f; for (int i = 0; i < 15; i++) r.l[i] = src.l[i + 1]; return r; }
TD_FN Row16 v_sqrt_agc(Row16 a) { return v_sqrt(a); }
TD_FN Row16 row_shr2(Row16 old, Row16 src) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = i < 2 ? old.l[i] : src.l[i - 2]; return r; }
TD_FN Row16 row_shl2(Row16 old, Row16 src) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = i < 14 ? src.l[i + 2] : old.l[i]; return r; }
TD_FN Row16 row_shl2_z(Row16 src) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = i < 14 ? src.l[i + 2] : 0.0f; return r; }
template <int H> TD_FN Row16 row_shr_h(Row16 old, Row16 src) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = i < H ? old.l[i] : src.l[i - H]; return r; }
template <int H> TD_FN Row16 row_shl_h(Row16 old, Row16 src) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = i < 16 - H ? src.l[i + H] : old.l[i]; return r; }
template <int H> TD_FN Row16 row_shl_h_z(Row16 src) { Row16 r; for (int i = 0; i < 16; i++) r.l[i] = i < 16 - H ? src.l[i + H] : 0.0f; return r; }
#endif  // TETRA_HOST_EMUL

template <int H, class V> TD_FN Pair<V> row_shr_h(Pair<V> old, Pair<V> src) {
    return Pair<V>(row_shr_h<H>(old.x(), src.x()), row_shr_h<H>(old.y(), src.y()));
}
template <int H, class V> TD_FN Pair<V> row_shl_h(Pair<V> old, Pair<V> src) {
    return Pair<V>(row_shl_h<H>(old.x(), src.x()), row_shl_h<H>(old.y(), src.y()));
}
template <int H, class V> TD_FN Pair<V> row_shl_h_z(Pair<V> src) { return Pair<V>(row_shl_h_z<H>(src.x()), row_shl_h_z<H>(src.y())); }
template <class V> TD_FN Pair<V> row_shr1(Pair<V> old, Pair<V> src) {
    return Pair<V>(row_shr1(old.x(), src.x()), row_shr1(old.y(), src.y()));
}
template <class V> TD_FN Pair<V> row_shl1(Pair<V> old, Pair<V> src) {
    return Pair<V>(row_shl1(old.x(), src.x()), row_shl1(old.y(), src.y()));
}
template <class V> TD_FN Pair<V> row_shl1_z(Pair<V> src) { return Pair<V>(row_shl1_z(src.x()), row_shl1_z(src.y())); }
template <class V> TD_FN Pair<V> row_shr2(Pair<V> old, Pair<V> src) {
    return Pair<V>(row_shr2(old.x(), src.x()), row_shr2(old.y(), src.y()));
}
template <class V> TD_FN Pair<V> row_shl2(Pair<V> old, Pair<V> src) {
    return Pair<V>(row_shl2(old.x(), src.x()), row_shl2(old.y(), src.y()));
}
template <class V> TD_FN Pair<V> row_shl2_z(Pair<V> src) { return Pair<V>(row_shl2_z(src.x()), row_shl2_z(src.y())); }

// ---------------------------------------------------------------------------------------------
// Run-time phasor (replaces libm cosf/sinf of SDR++ core math::phasor; same function as
// tetra_oracle_sincosf): k = rint(x/pi), three-term Cody-Waite reduction to r in [-pi/2, pi/2], minimax
// polynomials in r^2, sign (-1)^k applied to both results with one shift and two xors.  |error| <= 1.6e-7.
// SMALL: the caller guarantees |x| <= 2.5 pi (every phase the loops hand over is wrapped to [-pi, pi] or (-2 pi, 2 pi)),
// so k is in {-2..2} and the sign is applied with v_flip_by_k8 -- same bits, one instruction less.
template <class V, bool SMALL = false> TD_FN void sincos_t(V x, V& s, V& c) {
    V k = v_rint(x * 0.318309886183790672f);
    V nk = -k;
    V r = v_fma(nk, V(3.140625f), x);
    r = v_fma(nk, V(9.67502593994140625e-4f), r);
    r = v_fma(nk, V(1.509957990978376432e-7f), r);
    V z = r * r;
    V ps = v_fma(V(2.597026877992903e-06f), z, V(-0.0001980524102691561f));
    ps = v_fma(ps, z, V(0.008332998491823673f));
    ps = v_fma(ps, z, V(-0.16666656732559204f));
    ps = ps * z;
    V sr = v_fma(ps, r, r);
    V pc = v_fma(V(-2.604826931928983e-07f), z, V(2.476031113474164e-05f));
    pc = v_fma(pc, z, V(-0.0013888374669477344f));
    pc = v_fma(pc, z, V(0.04166663810610771f));
    pc = v_fma(pc, z, V(-0.5f));
    V cr = v_fma(pc, z, V(1.0f));
    if (SMALL) {
        s = v_flip_by_k8(sr, k);
        c = v_flip_by_k8(cr, k);
    } else {
        typename vtraits<V>::I ki = v_ftoi(k);
        s = v_flip_if_odd(sr, ki);
        c = v_flip_if_odd(cr, ki);
    }
}

// Two independent angles at once, (x.x, x.y) -> s = (sin x.x, sin x.y), c = (cos x.x, cos x.y): the same operations as two
// sincos_t<float, true> calls, issued as packed instructions (the Costas wave evaluates its loop phasor and the pi/4
// rotation phasor of pi4dqpsk_costas.cpp:7,16 together).  |x| <= 2.5 pi.
TD_FN void sincos_pair(Pair<float> x, Pair<float>& s, Pair<float>& c) {
    typedef Pair<float> P;
    const P t = pk_mul(x, P(0.318309886183790672f, 0.318309886183790672f));
    const P k(v_rint(t.x()), v_rint(t.y()));
    const P nk(-k.x(), -k.y());
    P r = pk_fma(nk, P(3.140625f, 3.140625f), x);
    r = pk_fma(nk, P(9.67502593994140625e-4f, 9.67502593994140625e-4f), r);
    r = pk_fma(nk, P(1.509957990978376432e-7f, 1.509957990978376432e-7f), r);
    const P z = pk_mul(r, r);
    P ps = pk_fma(P(2.597026877992903e-06f, 2.597026877992903e-06f), z, P(-0.0001980524102691561f, -0.0001980524102691561f));
    ps = pk_fma(ps, z, P(0.008332998491823673f, 0.008332998491823673f));
    ps = pk_fma(ps, z, P(-0.16666656732559204f, -0.16666656732559204f));
    ps = pk_mul(ps, z);
    const P sr = pk_fma(ps, r, r);
    P pc = pk_fma(P(-2.604826931928983e-07f, -2.604826931928983e-07f), z, P(2.476031113474164e-05f, 2.476031113474164e-05f));
    pc = pk_fma(pc, z, P(-0.0013888374669477344f, -0.0013888374669477344f));
    pc = pk_fma(pc, z, P(0.04166663810610771f, 0.04166663810610771f));
    pc = pk_fma(pc, z, P(-0.5f, -0.5f));
    const P cr = pk_fma(pc, z, P(1.0f, 1.0f));
    s = P(v_flip_by_k8(sr.x(), k.x()), v_flip_by_k8(sr.y(), k.y()));
    c = P(v_flip_by_k8(cr.x(), k.x()), v_flip_by_k8(cr.y(), k.y()));
}

// SDR++ core complex_t::operator*: a * (c + j s) = (a.re*c - a.im*s, a.im*c + a.re*s), every product and the
// sum/difference rounded separately.  Written with packed ops: (ar*c, ai*c) + (ai*(-s), ar*s).
template <class V> TD_FN Pair<V> cmul_phasor(Pair<V> a, V c, V s) {
    Pair<V> t1 = pk_mul(a, Pair<V>(c, c));
    Pair<V> t2 = pk_mul(pk_swap(a), Pair<V>(-s, s));      // (-(ai*s), ar*s): negating a factor is exact
    return pk_add(t1, t2);
}

// SDR++ core complex_t::fastAmplitude: `r > i ? r + 0.4f*i : i + 0.4f*r` with r = |re|, i = |im|, written as
// max + 0.4f*min -- the same two roundings on the same operands, so bit-identical for every non-NaN input.
template <class V> TD_FN V fast_amp(V re, V im) {
    V r = v_abs(re), i = v_abs(im);
    return v_max(r, i) + 0.4f * v_min(r, i);
}

// fll.cpp:141-145 from the four real band-edge sums c14 = (S1, S4), c32 = (S3, S2):
//   lbe = (S1 - S2, S4 + S3), hbe = (S1 + S2, S4 - S3), err = fastAmplitude(hbe) - fastAmplitude(lbe).
// Written on pairs so that the device issues two packed adds, one packed multiply and one packed add:
// d = (S1 - S2, S4 - S3) = (lbe.re, hbe.im), u = (S1 + S2, S4 + S3) = (hbe.re, lbe.im); every scalar operation and its
// rounding is the one fast_amp() performs.
template <class V> TD_FN V fll_error(Pair<V> c14, Pair<V> c32) {
    const Pair<V> sw = pk_swap(c32);                 // (S2, S3)
    const Pair<V> d = pk_sub(c14, sw), u = pk_add(c14, sw);
    const V hr = v_abs(u.x()), hi = v_abs(d.y()), lr = v_abs(d.x()), li = v_abs(u.y());
    const Pair<V> mx(v_max(hr, hi), v_max(lr, li)), mn(v_min(hr, hi), v_min(lr, li));
    const Pair<V> fa = pk_add(mx, pk_mul(Pair<V>(V(0.4f), V(0.4f)), mn));     // max + 0.4f*min, see fast_amp
    return fa.x() - fa.y();
}

// SDR++ core PhaseControlLoop<float, CLAMP>::advance.  The reference wraps with while loops; one
// conditional step each way is identical as long as |freq + alpha*err| < 2*pi, which the frequency
// limits of pi4dqpsk.cpp:17,21 guarantee (|freq| <= pi/2, |alpha*err| < 1).
// ALPHA0 (alpha known to be exactly 0): `freq + 0*err` equals `freq` for every finite err (it can differ
// only in the sign of a zero, which needs freq == -0, never produced by these loops), so the two ops are skipped.
template <class V, bool CLAMP, bool ALPHA0 = false> TD_FN void pcl_advance(V err, V& phase, V& freq, float alpha, float beta,
                                                      float minf, float maxf) {
    freq = v_clamp(freq + beta * err, minf, maxf);
    if (ALPHA0) phase = phase + freq;
    else phase = phase + (freq + alpha * err);
    if (CLAMP) {
        // limits are +-FL_M_PI: `phase > pi -> phase - 2pi`, `phase < -pi -> phase + 2pi` as one select
        // (x - (-d) is x + d exactly)
        const float pmax = kFlPi, pdelta = pmax - (-kFlPi);
        phase = v_wrap_sym(phase, pmax, pdelta);
    }
}

// ---------------------------------------------------------------------------------------------
// AGC and FLL constants + the AGC step (sample rate side of the chain).
// ---------------------------------------------------------------------------------------------
struct K1Consts {
    float agc_set_point, agc_rate, agc_max_gain;
    float fll_alpha, fll_beta, fll_min_freq, fll_max_freq;
};

// SDR++ core loop::FastAGC<complex_t>::process, one sample (called at src/dsp/pi4dqpsk.cpp:134).
template <class V> TD_FN Pair<V> agc_step(const K1Consts& k, Pair<V> in, V& g) {
    V ar = in.x() * g, ai = in.y() * g;
    V amp = v_sqrt_agc(ar * ar + ai * ai);
    g = g + (k.agc_set_point - amp) * k.agc_rate;
    g = v_sel(g > k.agc_max_gain, V(k.agc_max_gain), g);
    return Pair<V>(ar, ai);
}

// ---------------------------------------------------------------------------------------------
// Fused kernel building blocks.
//
// FLL row with LANES lanes per channel: a 16-lane row carries 16/LANES channels interleaved on its lanes (lane = HOP * pos
// + channel-in-row, HOP = 16/LANES), so every cross-lane move is a HOP-lane DPP shift and all channels' heads (lanes
// 0..HOP-1) and tails (lanes 16-HOP..15) fall on the row boundary where DPP's keep-old / zero-fill do the right thing.
// Band-edge taps are zero-padded at the old end to LANES positions x TAPS; padded tap kp lives at position
// LANES - 1 - kp/TAPS, slot kp % TAPS.  The FIRs run as ONE systolic array along the row: the derotated sample x_i is
// produced in the head lane and travels outward one position per step, partial sums are created at the tail and travel
// inward one position per TAPS - 1 steps, receiving their taps in ascending tap order -- bit-identical to a direct-form
// `for k: acc = fmaf(hist[k], tap[k], acc)` -- and complete in the head lane in the very step that produces x_i, where
// the FLL error needs them.  Three geometries are used: 8 x 9 (two channels per row; the 16-channel workgroup, fll_asm.inc),
// 4 x 17 (four channels per row: half the loop code per channel; the 32-channel workgroup, fll4_asm.inc) and 16 x 5 (one
// channel per row: the fewest tap FMAs per step, the shortest step; the 4-channel workgroup, fll16_asm.inc).
// ---------------------------------------------------------------------------------------------
constexpr int kF8Lanes = 8;
constexpr int kF8Taps = 9;
constexpr int kF8Pad = kF8Lanes * kF8Taps;   // 72: the longest filters the kernels take (the RRC window walk ends there too)
constexpr int kF4Lanes = 4;
constexpr int kF4Taps = 17;
constexpr int kF4Pad = kF4Lanes * kF4Taps;   // 68
constexpr int kF16Lanes = 16;                // a whole DPP row per channel: the 4-channel workgroup (at most 4 channels per CU)
constexpr int kF16Taps = 5;
constexpr int kF16Pad = kF16Lanes * kF16Taps;   // 80
constexpr int kF16LTaps = 9;                 // the LONG 4-channel workgroup: 16 positions x 9 taps = 144 padded taps, for filters of 73 .. 129
constexpr int kF16LPad = kF16Lanes * kF16LTaps; // taps (PI4DQPSK::setRRCTapCount takes any count, pi4dqpsk.cpp:56-70; fll16l_asm.inc)
constexpr int kF8LTaps = 17;                 // ... and the LONG 16-channel workgroup: 8 positions x 17 taps = 136 padded taps (fll8l_asm.inc)
constexpr int kF8LPad = kF8Lanes * kF8LTaps;
constexpr int kHistLong = 128;               // delay-line samples that variant keeps: the newest 80 (hist) + the 48 before them (hist_far)
constexpr int kBePadLong = kF16LPad;         // its band-edge tap tables: zero-padded (old end) to 144 entries
constexpr int kBePad = kPadTaps;             // band-edge tap tables are handed to the kernel zero-padded (old end) to 80 entries
constexpr int ct_gcd(int a, int b) { return b == 0 ? a : ct_gcd(b, a % b); }

template <int I, int N, class F> TD_FN void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

template <class V, int LANES, int TAPS> struct FllRowT {
    typedef Pair<V> P;
    static constexpr int kLanes = LANES, kTaps = TAPS, kRes = TAPS - 1, kHop = 16 / LANES;
    // the drivers below walk groups of kGroup = lcm(kRes, LANES) steps: whole schedule periods (the step's phase is a
    // compile-time index) and whole lane groups (delay-line samples are fetched, and x is stored, LANES at a time)
    static constexpr int kGroup = kRes / ct_gcd(kRes, LANES) * LANES;
    // the replay walks whole groups: the newest kReplay >= LANES * TAPS stored samples (older ones only reach
    // sums that complete, unused, before the first real step)
    static constexpr int kReplay = ((LANES * TAPS + kGroup - 1) / kGroup) * kGroup;
    // (the long row replays 144 ring slots: its 128 stored samples and, under zero taps, 16 zeros in front of them)
    static_assert(16 % LANES == 0 && (kReplay <= kHist || LANES * TAPS == kF16LPad || LANES * TAPS == kF8LPad), "row geometry");
    V ta[TAPS], tb[TAPS];
    P r14[kRes], r32[kRes];
    P xs;        // lane (pos, channel-in-row) holds x_{i-pos} of its channel
    V ph, fr;    // FLL phase / freq (meaningful in the head lanes)

    TD_MFN void clear_pipeline() {
        for (int q = 0; q < kRes; q++) { r14[q] = P(V(0.0f), V(0.0f)); r32[q] = P(V(0.0f), V(0.0f)); }
        xs = P(V(0.0f), V(0.0f));
    }

    // One sample step; PH = step index mod (TAPS - 1).  `a` = AGC output (or a stored x when REPLAY), valid in
    // the head lanes.  After the step xs holds the new x pipeline.
    template <int PH, bool REPLAY, bool ALPHA0> TD_MFN void step(const K1Consts& k, P a) {
        P x;
        if (REPLAY) {
            x = a;
        } else {
            V s, c;
            sincos_t<V, true>(-ph, s, c);                             // fll.cpp:137-138
            x = cmul_phasor<V>(a, c, s);
        }
        xs = row_shr_h<kHop>(x, xs);
        P c14 = pk_fma(xs, P(ta[TAPS - 1], ta[TAPS - 1]), r14[PH]);
        P c32 = pk_fma(xs, P(tb[TAPS - 1], tb[TAPS - 1]), r32[PH]);
        r14[PH] = pk_fma(xs, P(ta[0], ta[0]), row_shl_h_z<kHop>(c14));
        r32[PH] = pk_fma(xs, P(tb[0], tb[0]), row_shl_h_z<kHop>(c32));
        static_for<1, TAPS - 1>([&](auto Q) {
            constexpr int q = decltype(Q)::value, i = (PH + q) % kRes;
            r14[i] = pk_fma(xs, P(ta[TAPS - 1 - q], ta[TAPS - 1 - q]), r14[i]);
            r32[i] = pk_fma(xs, P(tb[TAPS - 1 - q], tb[TAPS - 1 - q]), r32[i]);
        });
        if (!REPLAY) {
            V err = fll_error<V>(c14, c32);                              // fll.cpp:141-145
            pcl_advance<V, true, ALPHA0>(err, ph, fr, k.fll_alpha, k.fll_beta, k.fll_min_freq, k.fll_max_freq);
        }
    }
};
template <class V> using FllRow8 = FllRowT<V, kF8Lanes, kF8Taps>;
template <class V> using FllRow4 = FllRowT<V, kF4Lanes, kF4Taps>;
template <class V> using FllRow16 = FllRowT<V, kF16Lanes, kF16Taps>;
template <class V> using FllRow16L = FllRowT<V, kF16Lanes, kF16LTaps>;
template <class V> using FllRow8L = FllRowT<V, kF8Lanes, kF8LTaps>;

// Drivers of an FLL row.  IO (device: LDS accesses of one lane; host emulation: arrays):
//   P    load_hist(int g)               lane (pos, ch) <- stored delay-line sample g*LANES + pos of the last Row::kReplay
//   P    sample(int s)                  AGC output s of the current tile, broadcast to each channel's lanes
//   void xs_store(int iend, int cnt, P xs)   lane with pos < cnt holds x_{iend-1-pos} (tile-relative iend)
template <class Row, class IO> TD_FN void fll_replay(Row& R, const K1Consts& k, IO& io) {
    typedef typename Row::P P;
    R.clear_pipeline();
    for (int grp = 0; grp < Row::kReplay / Row::kGroup; grp++) {
        P cur(0.0f, 0.0f);
        static_for<0, Row::kGroup>([&](auto S) {
            constexpr int s = decltype(S)::value;
            if (s % Row::kLanes == 0) cur = io.load_hist(grp * (Row::kGroup / Row::kLanes) + s / Row::kLanes);
            R.template step<s % Row::kRes, true, true>(k, cur);
            cur = row_shl_h<Row::kHop>(cur, cur);
        });
    }
}
// One tile of cnt <= tile_len samples (tile_len a multiple of the group).
template <class Row, class IO, bool ALPHA0> TD_FN void fll_tile(Row& R, const K1Consts& k, IO& io, int cnt) {
    for (int s0 = 0; s0 < cnt; s0 += Row::kGroup) {
        const int cg = (cnt - s0 < Row::kGroup) ? (cnt - s0) : Row::kGroup;
        if (cg == Row::kGroup) {
            static_for<0, Row::kGroup>([&](auto S) {
                constexpr int s = decltype(S)::value;
                R.template step<s % Row::kRes, false, ALPHA0>(k, io.sample(s0 + s));
                if (s % Row::kLanes == Row::kLanes - 1) io.xs_store(s0 + s + 1, Row::kLanes, R.xs);
            });
        } else {
            static_for<0, Row::kGroup - 1>([&](auto S) {
                constexpr int s = decltype(S)::value;
                if (s < cg) {
                    R.template step<s % Row::kRes, false, ALPHA0>(k, io.sample(s0 + s));
                    if (s % Row::kLanes == Row::kLanes - 1 || s == cg - 1) io.xs_store(s0 + s + 1, s % Row::kLanes + 1, R.xs);
                }
            });
        }
    }
}

// RRC matched filter, direct form, eight consecutive outputs per lane (SDR++ core FIR<complex_t,float>,
// called at src/dsp/pi4dqpsk.cpp:136).  With nt taps, output i0+m needs x_{i0+m-(nt-1)} .. x_{i0+m}; the
// eight outputs share the window x_{i0-(nt-1)} .. x_{i0+7} = nt+7 samples, walked in nchunks = ceil((nt+7)/8)
// chunks of 8.  ld(p) = x_{i0-(nt-1)+p}; tap4(q) = taps [4q, 4q+4) of the EXTENDED tap array
// ext[7 + k] = h[k] (k < nt), zero elsewhere (7 zeros in front, >= 16 behind).  Every output is one fmaf
// chain per component in ascending tap order; the zero taps outside a chain leave its accumulator untouched
// bit for bit (x finite: x*0 = +-0, and acc + +-0 == acc because a chain started at +0 is never -0).
// A runtime loop over chunks keeps only one chunk live in registers.
constexpr int kRrcMaxTaps = 80;
constexpr int kRrcOut = 8;
constexpr int kRrcExt = 104;   // >= 7 + 7 (alignment pad) + kRrcMaxTaps - 1 + 16, a multiple of 4
constexpr int kRrcMaxTapsLong = 129;      // the long 4-channel workgroup (filters of 73 .. 129 taps)
constexpr int kRrcExtLong = 160;          // >= 7 + 7 + 128 + 16
struct Tap4 { float v[4]; };
// One chunk of the walk.  ENDS: 0 = every product; 1 = the FIRST chunk of a window whose taps start exactly at ext[7]
// (no alignment pad): sample j meets tap kk = j - m of output m, which exists only for j >= m; 2 = the LAST chunk of
// a window whose newest tap sits at its first sample (nt - 1 a multiple of 8): kk = nt - 1 + j - m exists only for j <= m.
// The skipped products are the ones against zero taps, which leave the chain untouched bit for bit (above).
template <int ENDS, class LD, class LT> TD_FN void rrc_chunk8(int ck, LD& ld, LT& tap4, Pair<float>* acc) {
    const int p0 = ck * 8;
    Pair<float> x[8];
    float h[16];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = ld(p0 + j);
#pragma unroll
    for (int q = 0; q < 4; q++) {            // ext[p0 .. p0+15]: tap index kk = p0 - 7 + q'
        if ((ENDS == 1 && q == 0) || (ENDS == 2 && q >= 2)) continue;      // those taps meet no kept product
        const Tap4 t = tap4(2 * ck + q);
        h[4 * q] = t.v[0]; h[4 * q + 1] = t.v[1]; h[4 * q + 2] = t.v[2]; h[4 * q + 3] = t.v[3];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int m = 0; m < kRrcOut; m++) {
            if ((ENDS == 1 && j < m) || (ENDS == 2 && j > m)) continue;
            // sample p0+j meets tap kk = p0 + j - m of output m  ->  ext index kk + 7 = p0 + (j - m + 7)
            acc[m] = pk_fma(x[j], Pair<float>(h[j - m + 7], h[j - m + 7]), acc[m]);
        }
    }
}
// TRI: the caller guarantees an unpadded window with nt = 8 (nchunks - 1) + 1 taps (65 taps: 9 chunks): the two end
// chunks are triangular, 36 products instead of 64 each.
template <bool TRI, class LD, class LT> TD_FN void rrc_direct8(int nchunks, LD ld, LT tap4, Pair<float>* out) {
    Pair<float> acc[kRrcOut];
#pragma unroll
    for (int m = 0; m < kRrcOut; m++) acc[m] = Pair<float>(0.0f, 0.0f);
    if (TRI) {
        rrc_chunk8<1>(0, ld, tap4, acc);
#pragma unroll 1
        for (int ck = 1; ck < nchunks - 1; ck++) rrc_chunk8<0>(ck, ld, tap4, acc);
        rrc_chunk8<2>(nchunks - 1, ld, tap4, acc);
    } else {
#pragma unroll 1
        for (int ck = 0; ck < nchunks; ck++) rrc_chunk8<0>(ck, ld, tap4, acc);
    }
#pragma unroll
    for (int m = 0; m < kRrcOut; m++) out[m] = acc[m];
}

// ---------------------------------------------------------------------------------------------
// Kernel 2 per-channel symbol step: timing recovery -> Costas -> slicer/differential decoder.
// One lane per channel; plain float code (host builds use it for unit tests only).
// ---------------------------------------------------------------------------------------------
struct K2Consts {
    float tr_alpha, tr_beta, tr_min_freq, tr_max_freq;
    float costas_alpha, costas_beta, costas_min_freq, costas_max_freq;
};

struct K2State {
    float mu, omega;
    int offset;
    float cph, cfr, ph2;
    int prev;
};

// Timing recovery step (complex_fd.cpp:101-143), second half: from the three interpolated values v = f(T) (bank row `phase`),
// a = f(T+1) (row min(phase+1,127)), b = f(T-1) (row max(phase-1,0)) to the loop update.  Advances mu / omega / offset.
// MINADV = the smallest offset advance a symbol may make: 1 for parameter sets whose every symbol moves at least one sample
// (omega_min - |alpha| >= 1: the clamp is then neutral and guarantees loop termination when NaN/Inf has poisoned mu), 0 for the
// rest of the reference's domain, where floor(mu) = 0 makes COMPLEX_FD emit several symbols from ONE offset
// (complex_fd.cpp:141-143: `offset += delta` with delta == 0) -- there a poisoned loop stops advancing and is cut off at its
// row's capacity by the caller's per-symbol check.
template <int MINADV = 1>
TD_FN void k2_timing_tail(const K2Consts& k, K2State& st, int phase, float vr, float vi, float ar, float ai, float br, float bi) {
    // complex_fd.cpp:107-123, branch-free: one-sided differences at the bank edges, central difference inside
    // At the low edge the "row below" IS row `phase` (the caller clamps the neighbour rows), so b equals v bit for bit and a - b
    // is the reference's one-sided f(T+1) - f(T); likewise a == v at the high edge.
    const bool edge = phase == 0 || phase == kInterpPhases - 1;
    const float sc = edge ? 1.0f : 0.5f;                  // x*1.0f is exact, so the edge cases stay `a - b`
    const float dr = (ar - br) * sc, di = (ai - bi) * sc;
    // complex_fd.cpp:126,136-137
    float terr = ((vr > 0 ? 1.0f : -1.0f) * dr) + ((vi > 0 ? 1.0f : -1.0f) * di);
    terr = v_clamp(terr, -1.0f, 1.0f);
    // complex_fd.cpp:140-143
    pcl_advance<float, false>(terr, st.mu, st.omega, k.tr_alpha, k.tr_beta, k.tr_min_freq, k.tr_max_freq);
    float delta = v_floor(st.mu);
    // MINADV 1: a finite stream always advances by >= 1 sample there, so the max() is neutral; it guarantees forward progress
    // (loop termination) when NaN/Inf has poisoned mu.  MINADV 0: mu >= 0 for every finite stream (omega_min - |alpha| > 0), so
    // this max() is neutral too and only keeps a poisoned offset from running backwards.
    const int adv = (int)delta;
    st.offset += adv > MINADV ? adv : MINADV;
    st.mu = st.mu - delta;
}

// Timing recovery step (complex_fd.cpp:101-143).  w[0..7]: the 8 complex samples buffer[offset..offset+7];
// rows tm1/t0/tp1: interpolator bank rows max(phase-1,0), phase, min(phase+1,127).  Returns the interpolated
// symbol (vr, vi) and advances mu / omega / offset.  The three 8-tap dots run as packed (re,im) fmaf chains.
template <int MINADV = 1>
TD_FN void k2_timing(const K2Consts& k, K2State& st, int phase, const Pair<float>* w,
                     const float* tm1, const float* t0, const float* tp1, float* out_re, float* out_im) {
    Pair<float> v(0.0f, 0.0f), a(0.0f, 0.0f), b(0.0f, 0.0f);
#pragma unroll
    for (int j = 0; j < kInterpTaps; j++) {
        v = pk_fma(w[j], Pair<float>(t0[j], t0[j]), v);
        a = pk_fma(w[j], Pair<float>(tp1[j], tp1[j]), a);
        b = pk_fma(w[j], Pair<float>(tm1[j], tm1[j]), b);
    }
    k2_timing_tail<MINADV>(k, st, phase, v.x(), v.y(), a.x(), a.y(), b.x(), b.y());
    *out_re = v.x();
    *out_im = v.y();
}

#if TD_DEVICE
// The same step with the three dots on three lanes of a quad (the timing wave of the 16- and 4-channel workgroups spends four
// lanes on a channel): lane kq of the quad holds ONE bank row -- kq 0: row `phase`, 1: min(phase+1,127), 2: max(phase-1,0)
// (3: any) -- and runs one 8-tap chain; the three results are then broadcast over the quad and every lane makes the same loop
// update, so the four lanes carry identical state.  8 packed FMAs and 2 row loads per symbol instead of 24 and 6, at the price
// of six quad_perm moves; every value is the one k2_timing computes (each dot is the same fmaf chain).
template <int Q> TD_FN float quad_bcast(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), Q * 0x55, 0xf, 0xf, true));
}
template <int MINADV = 1>
TD_FN void k2_timing_quad(const K2Consts& k, K2State& st, int phase, const Pair<float>* w, const float* trow,
                          float* out_re, float* out_im) {
    Pair<float> d(0.0f, 0.0f);
#pragma unroll
    for (int j = 0; j < kInterpTaps; j++) d = pk_fma(w[j], Pair<float>(trow[j], trow[j]), d);
    const float vr = quad_bcast<0>(d.x()), vi = quad_bcast<0>(d.y());
    const float ar = quad_bcast<1>(d.x()), ai = quad_bcast<1>(d.y());
    const float br = quad_bcast<2>(d.x()), bi = quad_bcast<2>(d.y());
    k2_timing_tail<MINADV>(k, st, phase, vr, vi, ar, ai, br, bi);
    *out_re = vr;
    *out_im = vi;
}
// The same with TWO lanes per channel (the timing wave of the 32-channel workgroup: 32 channels on 64 lanes): both lanes run
// the chain of row `phase` (v), lane 0 also that of row min(phase+1,127) (a), lane 1 that of row max(phase-1,0) (b); a and b
// are then swapped between the neighbours.  16 packed FMAs and 4 row loads per symbol instead of 24 and 6.
TD_FN float pair_swap(float x) {      // quad_perm:[1,0,3,2]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xb1, 0xf, 0xf, true));
}
template <int MINADV = 1>
TD_FN void k2_timing_pair(const K2Consts& k, K2State& st, int phase, const Pair<float>* w, const float* t0, const float* t2,
                          bool second, float* out_re, float* out_im) {
    Pair<float> v(0.0f, 0.0f), d(0.0f, 0.0f);
#pragma unroll
    for (int j = 0; j < kInterpTaps; j++) {
        v = pk_fma(w[j], Pair<float>(t0[j], t0[j]), v);
        d = pk_fma(w[j], Pair<float>(t2[j], t2[j]), d);
    }
    const float ox = pair_swap(d.x()), oy = pair_swap(d.y());      // the neighbour's second dot
    const float ar = second ? ox : d.x(), ai = second ? oy : d.y();
    const float br = second ? d.x() : ox, bi = second ? d.y() : oy;
    k2_timing_tail<MINADV>(k, st, phase, v.x(), v.y(), ar, ai, br, bi);
    *out_re = v.x();
    *out_im = v.y();
}
#endif

// complex_fd.cpp:101: interpolator phase of the next symbol.
TD_FN int k2_phase(float mu) {
    int phase = (int)v_floor(mu * (float)kInterpPhases);
    phase = phase < 0 ? 0 : phase;
    return phase > kInterpPhases - 1 ? kInterpPhases - 1 : phase;
}

// Costas loop for one symbol (pi4dqpsk_costas.cpp:7-28): (*zr, *zi) = PI4DQPSK::process output; advances the loop state.
TD_FN void k2_costas_rot(const K2Consts& k, K2State& st, float vr, float vi, float* zr_out, float* zi_out) {
    // pi4dqpsk_costas.cpp:10-15.  ph2 only ever decreases (by pi/4 per symbol) and starts inside (-2 pi, 2 pi) -- a fresh
    // chain at 0, tetra_demod_set_state refuses anything else -- so `ph2 >= 2 pi -> ph2 - 2 pi` can never fire and only the
    // lower wrap is evaluated: the same values as the reference's if / else-if for every reachable state.
    const float t2 = st.ph2 + (-kFlPi / 4.0f);
    const float up2 = t2 + 2 * kFlPi;
    const float ph2 = v_sel(t2 <= -2 * kFlPi, up2, t2);
    st.ph2 = ph2;
    Pair<float> s2, c2;                                   // (loop phasor, pi/4-rotation phasor) in one packed evaluation
    sincos_pair(Pair<float>(-st.cph, ph2), s2, c2);
    const Pair<float> xx = cmul_phasor<float>(Pair<float>(vr, vi), c2.x(), s2.x());
    const Pair<float> zz = cmul_phasor<float>(xx, c2.y(), s2.y());
    const float zr = zz.x(), zi = zz.y();
    float cerr = ((zr > 0 ? 1.0f : -1.0f) * zi) - ((zi > 0 ? 1.0f : -1.0f) * zr);
    cerr = v_clamp(cerr, -1.0f, 1.0f);
    // pcl.advance (PhaseControlLoop<float, true>): the frequency and the sum as pcl_advance writes them; the wrap to [-pi, pi]
    // as a rounding, w = rint(x / 2 pi) in {-1, 0, 1} and x - w 2 pi in one fma -- three instructions and no compare / select
    // pair on VCC instead of copysign, subtract, compare, select.  Bit for bit the reference's `x > pi -> x - 2 pi, x < -pi ->
    // x + 2 pi` for every binary32 x of [-2 pi, 2 pi] (this sum stays inside 1.2 pi) except x = -0 -> +0, which a phase that
    // starts at +0 never reaches and tetra_demod_set_state stores as +0: the FLL blocks' wrap (gen_fll_asm.py),
    // tests/test_oracle.py::test_rint_phase_wrap_is_exact_for_every_phase.
    st.cfr = v_clamp(st.cfr + k.costas_beta * cerr, k.costas_min_freq, k.costas_max_freq);
    const float x = st.cph + (st.cfr + k.costas_alpha * cerr);
    const float w = v_rint(x * __builtin_bit_cast(float, 0x3e22f983u));      // the binary32 nearest 1 / (2 pi)
    st.cph = v_fma(-w, kFlPi - (-kFlPi), x);
    *zr_out = zr;
    *zi_out = zi;
}
// dqpsk_sym_extr.cpp:6-7,32: quadrant index of a symbol, counter-clockwise
TD_FN int k2_quadrant(float zr, float zi) {
    const int a = zi < 0, b = zr < 0;
    return (a << 1) | (a != b);
}
// dqpsk_sym_extr.cpp:33-51: phase step between two quadrants -> dibit, {0,1,2,3} -> {0,1,3,2}
TD_FN int k2_dibit(int symq, int prevq) {
    const int pd = (symq - prevq + 4) & 3;
    return pd ^ (pd >> 1);
}
// Costas loop + slicer + differential decoder for one symbol (pi4dqpsk_costas.cpp:7-28,
// dqpsk_sym_extr.cpp:6-7,32-52).  Returns the dibit; (*zr, *zi) = PI4DQPSK::process output.
TD_FN int k2_costas(const K2Consts& k, K2State& st, float vr, float vi, float* zr_out, float* zi_out) {
    k2_costas_rot(k, st, vr, vi, zr_out, zi_out);
    const int symq = k2_quadrant(*zr_out, *zi_out);
    const int d = k2_dibit(symq, st.prev);
    st.prev = symq;
    return d;
}

// Sync/quality statistic of DQPSKSymbolExtractor::process (dqpsk_sym_extr.cpp:8-31): per symbol the angular distance
// between the symbol and its quadrant's ideal point goes into a 4096-entry ring; every 256 symbols the ring's mean is
// published as `standarderr`, and `sync = standarderr < 0.35`.  This is the GUI's signal-quality meter, NOT on the bit
// path, so it is held to a tolerance instead of bit equality: the distance is pi/4 - atan(min/max) of the symbol's
// |re|,|im| (equal to |atan2(ideal) - atan2(sym)| up to float rounding; Abramowitz-Stegun 4.4.49 polynomial,
// |error| <= 2e-8, hardware reciprocal).  The ring and the mean are kept by a separate small kernel after the launch
// (k_quality in tetra_demod.hip), from the symbols the Costas wave wrote: only the value at the last 256-symbol boundary
// of a call is observable, so nothing of it has to ride on the chain's critical wave.
TD_FN float quality_distance(float zr, float zi) {
    const float ar = v_abs(zr), ai = v_abs(zi);
    const float hi = v_max(ar, ai), lo = v_min(ar, ai);
#if TD_DEVICE
    const float inv = __builtin_amdgcn_rcpf(hi);
#else
    const float inv = 1.0f / hi;
#endif
    const float r = hi > 0.0f ? lo * inv : 0.0f;     // atan2f(+0, +0) = 0 -> distance pi/4 (the other zeros: below)
    const float z = r * r;
    float p = v_fma(0.0028662257f, z, -0.0161657367f);
    p = v_fma(p, z, 0.0429096138f);
    p = v_fma(p, z, -0.0752896400f);
    p = v_fma(p, z, 0.1065626393f);
    p = v_fma(p, z, -0.1420889944f);
    p = v_fma(p, z, 0.1999355085f);
    p = v_fma(p, z, -0.3333314528f);
    p = v_fma(p, z, 1.0f);
    // atan2f's signed zeros (symbols with an exactly-zero component: digital silence).  The slicer files both zeros under
    // "positive" (x < 0 is false), atan2f does not: an imaginary part of -0 beside a negative (or -0) real part is the angle
    // -pi, +0 beside -0 is +pi -- |ideal - angle| = 3pi/4 + pi, pi/4 + pi, pi - pi/4 (the floats atan2f returns)
    const unsigned ui = __builtin_bit_cast(unsigned, zi), ur = __builtin_bit_cast(unsigned, zr);
    if (ui == 0x80000000u && zr < 0.0f) return 2.3561945f + 3.14159274f;
    if (ui == 0x80000000u && ur == 0x80000000u) return 0.785398185f + 3.14159274f;
    if (ui == 0u && ur == 0x80000000u) return 3.14159274f - 0.785398185f;
    return 0.785398163397448f - p * r;
}

}  // namespace tdm
