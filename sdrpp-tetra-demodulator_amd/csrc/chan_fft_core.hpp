// chan_fft_core.hpp -- the polyphase channeliser's M = 800 frame as a register / LDS mixed-radix FFT (32 x 5 x 5), lane level.
//
// One output frame of the analysis bank (include/tetra_chan.h) is an 800-point DFT of the folded, prototype-weighted samples:
//     X[k] = sum_r v[r] W^(k r),   W = exp(-j 2 pi / 800),   v[r] = sum_q h[l0(r) + 800 q] x[n - l0(r) - 800 q],  l0(r) = (n - r) mod 800.
// Round 4 evaluated it as a 25 x 32 matrix product (57 complex MACs per output: 365 kflop per frame, vector-pipe bound at 0.109 ms per
// 12500 frames); as a mixed-radix FFT it is ~39 kflop and the kernel is bound by its 120 MB of HBM traffic instead.
//
// Index map (decimation in time over r = n1 + 25 n2, decimation in frequency over k = k2 + 32 k1; n1, k1 < 25; n2, k2 < 32):
//     W^((n1 + 25 n2)(k2 + 32 k1)) = W32^(n2 k2) . W800^(n1 k2) . W25^(n1 k1)
//   stage 1   C[n1][k2] = sum_n2 v[n1 + 25 n2] W32^(n2 k2)                 25 FFTs of 32 points, one per LANE (n1), data in registers
//   twiddle   C'[n1][k2] = C[n1][k2] W800^(n1 k2)
//   stage 2   X[k2 + 32 k1] = sum_n1 C'[n1][k2] W25^(n1 k1)                32 DFTs of 25 points (5 x 5), one per LANE (k2)
// so the last stage leaves lane k2 holding X[k2 + 32 k1]: for every k1 the 32 lanes store 32 CONSECUTIVE bins = one 256-byte
// run of the frame's row out[m][0 .. 800) -- fully coalesced [frame][channel] stores -- and stage 1 reads v[] with consecutive lanes
// on consecutive addresses.  Between the stages the 25 x 32 block turns through LDS (row stride 33: conflict-free both ways).
//
// This header holds the two register-level transforms and the fold of one (slot, 8-frame block); chan_fft_kernel (tetra_chan.hip)
// arranges them over a workgroup.  It compiles for the host as well (tests/emul/chan_emul.cpp runs the kernel's phases thread by
// thread against the double-precision definition, oracle/chan_oracle.c), so everything here is plain C++ on float pairs.
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#define CHAN_HD __host__ __device__ __forceinline__
#else
#define CHAN_HD inline
#endif
// The library is compiled with -ffp-contract=off (the demodulator's arithmetic contract); the channeliser is held to a tolerance
// against the double-precision definition, so its transforms may fuse a * b + c: first statement of every function body below.
#if defined(__clang__)
#define CHAN_FP_FAST _Pragma("clang fp contract(fast)")
#else
#define CHAN_FP_FAST
#endif

namespace chanfft {

constexpr int kM = 800, kN1 = 25, kN2 = 32;
constexpr int kBlockFrames = 8;                 // frames per workgroup pass: 4 waves x 2 frames, = 4 M samples at D = M / 2
constexpr int kRowStride = 33;                  // complex elements per n1-row of the transposed block in LDS
constexpr int kFrameLds = kN1 * kRowStride;     // complex elements of LDS per frame (825 >= 800: the fold's linear v[r] fits too)
constexpr int kFoldThreads = 200;               // threads that fold: 4 slots each = 800 bins

struct alignas(8) c32 {          // one 8-byte load / store / LDS access per complex number
    float x, y;
};
CHAN_HD c32 mk(float x, float y) { c32 r; r.x = x; r.y = y; return r; }
CHAN_HD c32 cadd(c32 a, c32 b) { return mk(a.x + b.x, a.y + b.y); }
CHAN_HD c32 csub(c32 a, c32 b) { return mk(a.x - b.x, a.y - b.y); }
CHAN_HD c32 cmul(c32 a, c32 b) {
    CHAN_FP_FAST
    return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
CHAN_HD c32 cmul_mj(c32 a) { return mk(a.y, -a.x); }          // a . (-j)
CHAN_HD c32 cscale(c32 a, float s) { return mk(a.x * s, a.y * s); }

// cos / sin of -2 pi k / 32, k = 0 .. 15 (forward transform)
constexpr float kC32[16] = { 1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960218f,
                             0.38268343236508977f, 0.19509032201612825f, 0.0f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960218f,
                             -0.70710678118654752f, -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323043f };
constexpr float kS32[16] = { -0.0f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960218f, -0.70710678118654752f, -0.83146961230254524f,
                             -0.92387953251128674f, -0.98078528040323043f, -1.0f, -0.98078528040323043f, -0.92387953251128674f, -0.83146961230254524f,
                             -0.70710678118654752f, -0.55557023301960218f, -0.38268343236508977f, -0.19509032201612825f };

CHAN_HD constexpr int bitrev5(int i) { return ((i & 1) << 4) | ((i & 2) << 2) | (i & 4) | ((i & 8) >> 2) | ((i & 16) >> 4); }

// 32-point forward FFT, decimation in frequency, in place: on return x[p] = X[bitrev5(p)].  Fully unrolled on the device, so every
// twiddle is a literal and the trivial ones (1, -j) cost nothing.
CHAN_HD void fft32_dif(c32 x[32]) {
    CHAN_FP_FAST
#pragma unroll
    for (int span = 16; span >= 1; span >>= 1) {
        const int step = 16 / span;             // twiddle index stride: W_(2 span)^i = W32^(i step)
#pragma unroll
        for (int base = 0; base < 32; base += 2 * span) {
#pragma unroll
            for (int i = 0; i < span; i++) {
                const c32 a = x[base + i], b = x[base + i + span];
                x[base + i] = cadd(a, b);
                const c32 d = csub(a, b);
                const int t = i * step;
                if (t == 0) x[base + i + span] = d;
                else if (t == 8) x[base + i + span] = cmul_mj(d);
                else x[base + i + span] = cmul(d, mk(kC32[t], kS32[t]));
            }
        }
    }
}

// 5-point forward DFT in place (W5 = exp(-j 2 pi / 5)).
constexpr float kC5a = 0.30901699437494742f, kC5b = -0.80901699437494742f;   // cos 2pi/5, cos 4pi/5
constexpr float kS5a = 0.95105651629515357f, kS5b = 0.58778525229247313f;    // sin 2pi/5, sin 4pi/5
CHAN_HD void dft5(c32& x0, c32& x1, c32& x2, c32& x3, c32& x4) {
    CHAN_FP_FAST
    const c32 t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
    const c32 a1 = mk(x0.x + kC5a * t1.x + kC5b * t2.x, x0.y + kC5a * t1.y + kC5b * t2.y);
    const c32 a2 = mk(x0.x + kC5b * t1.x + kC5a * t2.x, x0.y + kC5b * t1.y + kC5a * t2.y);
    const c32 b1 = mk(kS5a * t3.x + kS5b * t4.x, kS5a * t3.y + kS5b * t4.y);
    const c32 b2 = mk(kS5b * t3.x - kS5a * t4.x, kS5b * t3.y - kS5a * t4.y);
    x0 = mk(x0.x + t1.x + t2.x, x0.y + t1.y + t2.y);
    // X1 = a1 - j b1, X4 = a1 + j b1, X2 = a2 - j b2, X3 = a2 + j b2
    x1 = mk(a1.x + b1.y, a1.y - b1.x);
    x4 = mk(a1.x - b1.y, a1.y + b1.x);
    x2 = mk(a2.x + b2.y, a2.y - b2.x);
    x3 = mk(a2.x - b2.y, a2.y + b2.x);
}

// cos / sin of -2 pi k / 25, k = 0 .. 16 (the 5 x 5 map needs W25^(b c), b, c <= 4)
constexpr float kC25[17] = { 1.0f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899666f, 0.30901699437494742f,
                             0.062790519529313374f, -0.18738131458572463f, -0.42577929156507272f, -0.63742398974868975f, -0.80901699437494742f,
                             -0.92977648588825146f, -0.99211470131447788f, -0.99211470131447788f, -0.92977648588825146f, -0.80901699437494742f,
                             -0.63742398974868975f };
constexpr float kS25[17] = { -0.0f, -0.24868988716485479f, -0.48175367410171532f, -0.68454710592868873f, -0.84432792550201508f, -0.95105651629515357f,
                             -0.99802672842827156f, -0.98228725072868872f, -0.90482705246601958f, -0.77051324277578925f, -0.58778525229247313f,
                             -0.36812455268467797f, -0.12533323356430426f, 0.12533323356430426f, 0.36812455268467797f, 0.58778525229247313f,
                             0.77051324277578925f };

// 25-point forward DFT, in place, natural order in and out: n = 5 a + b, k = c + 5 d,
//   Y[b][c] = sum_a x[5 a + b] W5^(a c);  Z = Y W25^(b c);  X[c + 5 d] = sum_b Z[b][c] W5^(b d).
CHAN_HD void dft25(c32 x[25]) {
    CHAN_FP_FAST
#pragma unroll
    for (int b = 0; b < 5; b++) dft5(x[b], x[5 + b], x[10 + b], x[15 + b], x[20 + b]);       // a -> c: x[5 c + b] = Y[b][c]
#pragma unroll
    for (int b = 1; b < 5; b++)
#pragma unroll
        for (int c = 1; c < 5; c++) x[5 * c + b] = cmul(x[5 * c + b], mk(kC25[b * c], kS25[b * c]));
#pragma unroll
    for (int c = 0; c < 5; c++) dft5(x[5 * c], x[5 * c + 1], x[5 * c + 2], x[5 * c + 3], x[5 * c + 4]);   // b -> d: x[5 c + d] = X[c + 5 d]
    // natural order: X[k] with k = c + 5 d sits at x[5 c + d]: transpose the 5 x 5 block (register renaming on the device)
#pragma unroll
    for (int c = 0; c < 5; c++)
#pragma unroll
        for (int d = c + 1; d < 5; d++) {
            const c32 t = x[5 * c + d];
            x[5 * c + d] = x[5 * d + c];
            x[5 * d + c] = t;
        }
}

// ---- the fold of one slot over one block of kBlockFrames frames (decimation D = M / 2) ----------------------------------
// The block's first frame has its newest sample at x0[0] (x0 points INTO the sample buffer) with absolute time = a (mod 800).  Slot
// u in [0, 800) owns bin r = (a - u) mod 800; frame t of the block (newest sample x0[400 t]) weights it with
//   t even: l0 = u               samples x0[400 t - u - 800 q]
//   t odd : l0 = (u + 400) % 800 samples x0[400 t - l0 - 800 q]
// which for u < 400 (class 0) are the SAME samples as frame t - 1 and for u >= 400 (class 1) the same as frame t + 1: with
// S[m] = x0[-u + 800 m] every frame is  v_t = sum_q coef[t & 1][q] S[((t + cls) >> 1) - q],  m in [-(P - 1), 4]: P + 4 loads serve 8
// frames (4 new samples per slot and block; the rest is re-read from L1 / L2 by the next block).
//   coef[0][q] = h[u + 800 q]                    (even frames)
//   coef[1][q] = h[(u + 400) % 800 + 800 q]      (odd frames)
template <int P> struct FoldCoef {
    float c[2][P];
};
// The prototype re-ordered for the fold (host side, once per handle): ht[u][parity][q], so that a slot's 2 P coefficients are 8 P
// contiguous bytes and the lanes of a wave (consecutive u) read one contiguous run with 16-byte loads.
inline void fold_transpose_prototype(const float* h, int P, float* ht) {
    for (int u = 0; u < kM; u++) {
        const int uo = u < kM / 2 ? u + kM / 2 : u - kM / 2;
        for (int q = 0; q < P; q++) {
            ht[(2 * u + 0) * P + q] = h[u + kM * q];
            ht[(2 * u + 1) * P + q] = h[uo + kM * q];
        }
    }
}
template <int P> CHAN_HD void fold_load_coef(const float* ht, int u, FoldCoef<P>& k) {
#pragma unroll
    for (int q = 0; q < P; q++) {
        k.c[0][q] = ht[(2 * u + 0) * P + q];
        k.c[1][q] = ht[(2 * u + 1) * P + q];
    }
}
// ---- one block of kBlockFrames frames on a workgroup of 256 threads: three phases, a barrier after each ------------------
// Sample formats of the wideband input (round 6): complex64, or what an SDR's DMA delivers -- interleaved int16 / int8 I, Q pairs,
// converted in the fold's load (value / 32768, value / 128: exact in binary32), so that an integer capture crosses HBM once at 4 or
// 2 bytes per sample instead of 8.
enum { kFmtC32 = 0, kFmtCs16 = 1, kFmtCs8 = 2 };
struct cs16 { int16_t x, y; };
struct cs8 { int8_t x, y; };
template <int FMT> CHAN_HD c32 load_sample(const void* x, long long s) {
    if (FMT == kFmtCs16) {
        const cs16 v = reinterpret_cast<const cs16*>(x)[s];
        return mk((float)v.x * (1.0f / 32768.0f), (float)v.y * (1.0f / 32768.0f));
    }
    if (FMT == kFmtCs8) {
        const cs8 v = reinterpret_cast<const cs8*>(x)[s];
        return mk((float)v.x * (1.0f / 128.0f), (float)v.y * (1.0f / 128.0f));
    }
    return reinterpret_cast<const c32*>(x)[s];
}

struct BlockCtx {
    const void* x;        // the call's n_in NEW samples in the format the kernel is compiled for, where the caller left them (read in place: no staging copy)
    const c32* hist;      // the L - 1 samples before them, oldest first (the handle's carried delay line)
    int n_in;
    c32* out;             // [frames][800]
    const float* h;       // prototype, re-ordered for the fold: [800][2][P] (fold_transpose_prototype)
    const c32* tw;        // [25][32]: W800^(n1 k2)
    int frames;           // frames of this call
    int ph0;              // samples already consumed towards the call's first frame
    long long abs0;       // absolute time of the first new sample
    int L;                // 800 P
};

// Sample s of the stream, s counted from the call's first new sample.  Blocks in the middle of a call read only new samples and
// index the caller's buffer directly (CAREFUL = false).  The first two blocks reach back into the delay line (s < 0), and the last
// block of a call whose frame count is not a multiple of 8 asks for samples past the call's end for the frames it does not store:
// those read the last sample instead (any value would do: it only reaches results that are never stored).
template <bool CAREFUL, int FMT> CHAN_HD c32 sample_at(const BlockCtx& c, long long s) {
    if (!CAREFUL) return load_sample<FMT>(c.x, s);
    if (s < 0) return c.hist[(c.L - 1) + s];
    return load_sample<FMT>(c.x, s < c.n_in ? s : c.n_in - 1);
}
// does a block with its first frame's newest sample at index newest0 need the careful accessor?  (it reads
// [newest0 - (L - 1), newest0 + 7 * 400])
CHAN_HD bool block_is_careful(const BlockCtx& c, long long newest0) {
    return newest0 < c.L - 1 || newest0 + (kBlockFrames - 1) * (kM / 2) > c.n_in - 1;
}

template <int P, int CLS, bool CAREFUL, int FMT> CHAN_HD void fold_slot(const BlockCtx& c, long long s0, const FoldCoef<P>& k, c32 out[kBlockFrames]) {
    CHAN_FP_FAST
    constexpr int kS = P + 3 + CLS;                   // class 0 frames reach sample m = 3, class 1 frames m = 4
    c32 S[kS];                                        // S[j] = sample m = j - (P - 1) = stream sample s0 + 800 m  (s0 = newest0 - u)
#pragma unroll
    for (int j = 0; j < kS; j++) S[j] = sample_at<CAREFUL, FMT>(c, s0 + kM * (j - (P - 1)));
#pragma unroll
    for (int t = 0; t < kBlockFrames; t++) {
        const int s = (t + CLS) >> 1;
        float ar = 0.f, ai = 0.f;
#pragma unroll
        for (int q = 0; q < P; q++) {
            const c32 xv = S[s - q + (P - 1)];
            const float hv = k.c[t & 1][q];
            ar += hv * xv.x;
            ai += hv * xv.y;
        }
        out[t] = mk(ar, ai);
    }
}

// phase 1 (threads < kFoldThreads): fold the block's 8 frames, v_t[r] -> lds[t][r].  Slot after slot: a slot's 2 P coefficients and
// P + 4 samples live only while its 8 frames are summed.
template <int P, bool CAREFUL, int FMT> CHAN_HD void phase_fold_t(const BlockCtx& c, int blk, int tid, c32* lds) {
    if (tid >= kFoldThreads) return;
    const long long newest0 = (long long)(kBlockFrames * blk + 1) * (kM / 2) - 1 - c.ph0;     // index of the block's first frame's newest sample among the new samples
    const int a = (int)((c.abs0 + newest0) % kM);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int u = kFoldThreads * i + tid;
        FoldCoef<P> k;
        fold_load_coef<P>(c.h, u, k);
        c32 v[kBlockFrames];
        if (i < 2) fold_slot<P, 0, CAREFUL, FMT>(c, newest0 - u, k, v);
        else fold_slot<P, 1, CAREFUL, FMT>(c, newest0 - u, k, v);
        int r = a - u;
        r += r < 0 ? kM : 0;
#pragma unroll
        for (int t = 0; t < kBlockFrames; t++) lds[t * kFrameLds + r] = v[t];
#if defined(__HIP_DEVICE_COMPILE__)
        // two slots' loads in flight at a time, not four: the scheduler would otherwise hoist all 4 x (P + 4 + 2 P) loads to the top
        // and spill (256 VGPRs); with the fence after every second slot the phase needs ~120
        if (i == 1) asm volatile("" ::: "memory");
#endif
    }
}
template <int P, int FMT = kFmtC32> CHAN_HD void phase_fold(const BlockCtx& c, int blk, int tid, c32* lds) {
    const long long newest0 = (long long)(kBlockFrames * blk + 1) * (kM / 2) - 1 - c.ph0;
    if (block_is_careful(c, newest0)) phase_fold_t<P, true, FMT>(c, blk, tid, lds);      // (uniform over the workgroup)
    else phase_fold_t<P, false, FMT>(c, blk, tid, lds);
}

// phase 2 (lanes n1 < 25 of each half-wave; wave w, half f -> frame 2 w + f): 32-point FFT over n2, result transposed IN PLACE:
// lane n1 reads v[n1 + 25 n2] and writes C[n1][k2] at [33 n1 + k2] of the same frame region.  That is safe on the GPU because the
// lanes of a frame are lanes of ONE wavefront: every lane's 32 reads have returned before any lane issues a write (each output
// depends on all 32 inputs, so no write can move above a read).  The host emulation, which runs threads one after the other, calls
// the two halves separately for the same reason.
CHAN_HD bool phase_fft32_compute(int tid, const c32* lds, c32 x[32]) {
    const int n1 = tid & 31, t = tid >> 5;              // t = 2 (tid >> 6) + ((tid >> 5) & 1)
    if (n1 >= kN1) return false;
    const c32* f = lds + t * kFrameLds;
#pragma unroll
    for (int n2 = 0; n2 < 32; n2++) x[n2] = f[n1 + kN1 * n2];
    fft32_dif(x);
    return true;
}
CHAN_HD void phase_fft32_store(int tid, c32* lds, const c32 x[32]) {
    const int n1 = tid & 31, t = tid >> 5;
    c32* f = lds + t * kFrameLds;
#pragma unroll
    for (int p = 0; p < 32; p++) f[kRowStride * n1 + bitrev5(p)] = x[p];
}

// phase 3 (every lane: k2 = tid & 31, frame t = tid >> 5): twiddle, 25-point DFT over n1, coalesced stores of X[k2 + 32 k1]
// The inter-stage twiddles W800^(n1 k2), n1 = 1 .. 24, of lane k2: 24 loads from a 6.4 KB table.  Requested BEFORE stage 1 (they do not
// depend on it) so that their latency passes behind the 32-point FFTs instead of in front of the 5 x 5 DFTs (measured: 4 us of 37).
CHAN_HD void load_twiddles(const BlockCtx& c, int tid, c32 tw[kN1 - 1]) {
    const int k2 = tid & 31;
#pragma unroll
    for (int n1 = 1; n1 < kN1; n1++) tw[n1 - 1] = c.tw[n1 * kN2 + k2];
}
template <int EXP = 0> CHAN_HD void phase_dft25_store(const BlockCtx& c, long long j0, int tid, const c32* lds, const c32 tw[kN1 - 1]) {
    CHAN_FP_FAST
    const int k2 = tid & 31, t = tid >> 5;
    const long long j = j0 + t;               // j0 = the first frame of the block (of the pair)
    const c32* f = lds + t * kFrameLds;
    c32 x[25];
#pragma unroll
    for (int n1 = 0; n1 < kN1; n1++) {
        const c32 v = f[kRowStride * n1 + k2];
        x[n1] = (n1 == 0 || (EXP & 2)) ? v : cmul(v, tw[n1 - 1]);
    }
    dft25(x);
    if (j >= c.frames) return;
    c32* dst = c.out + j * kM + k2;
    if (EXP & 1) {                            // (experiment: the whole transform, one store)
        c32 acc = x[0];
#pragma unroll
        for (int k1 = 1; k1 < kN1; k1++) acc = cadd(acc, x[k1]);
        dst[0] = acc;
        return;
    }
    // (16-byte stores through a lane-pair exchange were measured: no gain, 38.2 vs 37.7 us -- profiles/r05/README.md)
#pragma unroll
    for (int k1 = 0; k1 < kN1; k1++) dst[kN2 * k1] = x[k1];
}

}  // namespace chanfft
