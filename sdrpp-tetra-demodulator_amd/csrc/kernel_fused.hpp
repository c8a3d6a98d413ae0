// kernel_fused.hpp -- the whole demodulator chain of 16 channels in ONE workgroup of seven specialised
// wavefronts (included by tetra_demod.hip; device code only).
//
// Why this shape (measured on MI355X, profiles/r02/r02_a_issue_model.md): a gfx950 wavefront issues one instruction
// of ANY kind (VALU, SALU, s_nop, s_waitcnt, LDS) per ~4.7 clocks whether or not it depends on the previous one, so
// a role's time is its instruction count, and the chain's pace is the instruction count of its busiest wave.  With
// 4096 channels there are 16 channels per CU; the loop code (NCO sincos, error functions, loop filters) costs the
// same whether a wave carries 4 or 64 channels.  A CU's 16 channels are therefore split by STAGE, each stage at the
// widest lane occupancy its recurrence allows, and the one stage that is too long for one wave -- the FLL, 65-tap
// complex band-edge FIR pair inside a per-sample feedback loop -- is cut in two at the 16 newest taps:
//
//   wave  role                                              lanes/channel   SIMD
//   L0,L1 FLL loop: NCO, 16 newest band-edge taps, error,   8 (two channels 0, 1   <- the pace-setters
//         loop filter (FllNear8)                              per DPP row)
//   H     FLL helper: the 68 older (zero-padded) taps of     4 (four per     3
//         both FIRs as a systolic array (FllFar4)             row)
//   D     ML timing recovery                                 1               2
//   A     AGC                                                1               0 (with L0)
//   E     Costas + slicer + diff. decoder + bit unpacker     1               1 (with L1)
//   C     RRC matched filter (time-parallel)                 4 x 8 outputs   2 (with D)
//
// The issue arbiter serves the OLDEST wave of a SIMD first, so the recurrence-bound roles take the lower wave index of
// their SIMD.  Stages are connected by LDS rings (AGC out -> FLL out x -> RRC out y -> symbols) and run as a software
// pipeline over 32-sample tiles with one workgroup barrier per tile: in epoch e, A works on tile e, L and H on e-1, C
// on e-2, D consumes y of tiles <= e-3 and E the symbols D published before the epoch.  Inside an epoch L and H hand
// over through LDS without barriers: H follows L's x by reading a progress counter, and L injects H's partial sums
// F_n eight samples before output n completes (see FllNear8 / FllFar4 in demod_core.hpp), so H has eight sample
// periods for its round trip.  No intermediate touches HBM: the kernel reads 8 B and writes 1 B per input sample.
#pragma once

#include "fll_asm.inc"

#ifndef TETRA_ROLE_MAP
#define TETRA_ROLE_MAP 3
#endif

namespace {

constexpr int kFT = 32;                  // samples per tile (pipeline epoch)
constexpr int kFCh = 16;                 // channels per workgroup
constexpr int kFWaves = (TETRA_ROLE_MAP == 2 || TETRA_ROLE_MAP == 4) ? 8 : 7;
constexpr int kFThreads = 64 * kFWaves;
constexpr int kFX = 256;                 // x ring (FLL output) per channel
constexpr int kFXS = kFX + 1;            // row stride (odd: spreads channels over LDS banks)
constexpr int kFY = 128;                 // y ring (RRC output) per channel ...
constexpr int kFYM = 8;                  // ... plus a mirror of the first slots so the interpolator window never wraps
constexpr int kFYS = kFY + kFYM + 1;
constexpr int kFS = 64;                  // symbol ring per channel
constexpr int kFF = 32;                  // far-sum ring (F_n) per channel
constexpr int kFFS = kFF + 1;
constexpr int kTH = 17;                  // far taps per position of the helper wave: 4 x 17 + 16 = 84 padded taps
constexpr int kFarTaps = 4 * kTH;
constexpr int kPadBe = kFarTaps + kNearTaps;
static_assert(kPadBe >= kHist, "padded band-edge filter must cover the longest supported filter");
static_assert((kTH - 1) == 16 && kFT % (kTH - 1) == 0, "the helper wave's schedule period must divide the tile");

// wave index -> role; waves w and w+4 share a SIMD (TETRA_ROLE_MAP: placement experiments, profiles/r02)
#if TETRA_ROLE_MAP == 0
enum { kRoleL0 = 0, kRoleL1 = 1, kRoleD = 2, kRoleH = 3, kRoleA = 4, kRoleE = 5, kRoleC = 6, kRoleNone = 7 };      // {L0,A} {L1,E} {D,C} {H}
#elif TETRA_ROLE_MAP == 1
enum { kRoleL0 = 0, kRoleL1 = 1, kRoleD = 2, kRoleH = 3, kRoleA = 4, kRoleC = 5, kRoleE = 6, kRoleNone = 7 };      // {L0,A} {L1,C} {D,E} {H}
#elif TETRA_ROLE_MAP == 2
enum { kRoleL0 = 0, kRoleL1 = 1, kRoleD = 2, kRoleH = 3, kRoleC = 4, kRoleNone = 5, kRoleE = 6, kRoleA = 7 };      // {L0,C} {L1} {D,E} {H,A}
#elif TETRA_ROLE_MAP == 3
enum { kRoleL0 = 0, kRoleL1 = 1, kRoleE = 2, kRoleH = 3, kRoleA = 4, kRoleC = 5, kRoleD = 6, kRoleNone = 7 };      // {L0,A} {L1,C} {E,D} {H}
#elif TETRA_ROLE_MAP == 4
enum { kRoleL0 = 0, kRoleL1 = 1, kRoleD = 2, kRoleH = 3, kRoleNone = 4, kRoleC = 5, kRoleE = 6, kRoleA = 7 };      // {L0} {L1,C} {D,E} {H,A}
#endif

struct FusedParams {
    const float2* iq;
    long long in_ch_stride, in_t_stride;
    int n, n_channels;
    // state
    float *agc_g, *fll_ph, *fll_fr;
    float2* hist;        // [C][kHist]
    float *mu, *omega;
    int* offset;
    float *cph, *cfr, *ph2;
    int* prev;
    float2* ybuf;        // [C][7]
    // tables
    const float* be_re84;   // band-edge taps zero-padded (old end) to kPadBe
    const float* be_im84;
    const float* rrc_ext;   // [kRrcExt] RRC taps as rrc_direct8 wants them: ext[7 + rrc_pad + k] = h[k], zero elsewhere
    int ntaps;
    const float* bank;
    // outputs
    uint8_t* bits;
    long long bits_stride;
    int* n_bits;
    float2* sym;         // optional
    float2* y_dbg;       // optional: time-major scratch [(7+n)][C], row 7+i = y_i
    // optional sync/quality statistic (TETRA_FLAG_QUALITY): ring [C][4096] + per-channel state; null = off
    float* q_ring;
    double* q_sum;
    int *q_ptr, *q_disp, *q_sync;
    float* q_err;
    K1Consts k1;
    K2Consts k2;
    long long* prof;     // TETRA_DEMOD_DEBUG builds only: [workgroups][8] = busy clocks of waves 0..6 inside their epoch
                         // bodies (barrier waits excluded), [7] = clocks from kernel entry to exit of wave 0; null = off
};

struct FusedLds {
    float2 a_buf[2][kFCh][kFT];
    float2 x_ring[kFCh][kFXS];
    float2 y_ring[kFCh][kFYS];
    float2 s_ring[kFCh][kFS];
    __attribute__((aligned(16))) float4 f_ring[kFCh][kFFS];
    // where the lanes that hold nothing worth keeping send their share of a wave-wide ring store (the loop waves' x of
    // the non-head positions, the helper's partial sums of the non-head positions): lane j writes at 8*j resp. 16*j' plus
    // the store's immediate offset, so no two lanes of one store ever hit the same address
    __attribute__((aligned(16))) float4 dump[80];
    float be84[2][kPadBe];   // band-edge taps (re, im), zero-padded: the helper wave's assembly loads its 34 taps from here
    int s_avail[kFCh];
    // hand-over counters of the FLL pair: x_i is in x_ring for i < x_done[w] (loop wave w), F_m is in f_ring for m < f_done
    int x_done[2];
    int f_done;
    int stuck;           // set by a wave whose wait ran into the watchdog: every wait gives up (results invalid, no hang)
    // interpolator bank with row 0 repeated in front and row 127 behind: rows max(p-1,0), p, min(p+1,127) of
    // complex_fd.cpp:102-121 are then the 24 contiguous floats at bank[p * 8]
    __attribute__((aligned(16))) float bank[(kInterpPhases + 2) * kInterpTaps];
    __attribute__((aligned(16))) float rrc[kRrcExt];       // zero-extended taps, see rrc_direct8
};
static_assert(sizeof(FusedLds) <= 80 * 1024, "two workgroups must fit one CU's 160 KB of LDS");

// Typed LDS pointers built from a 32-bit LDS byte address.  Keeping the (loop-invariant) row base in one pinned vector
// register makes the compiler address a sliding window as `base register + immediate offsets` (ds_read2_b64 /
// ds_read_b128 with offset fields) instead of re-deriving every element's address from the LDS struct offset.
typedef float vfloat2 __attribute__((ext_vector_type(2)));
typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const vfloat2 lds_cfloat2;
typedef __attribute__((address_space(3))) const vfloat4 lds_cfloat4;
typedef __attribute__((address_space(3))) vfloat2 lds_float2;
typedef __attribute__((address_space(3))) vfloat4 lds_float4;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }
__device__ __forceinline__ unsigned pin_u32(unsigned x) { asm volatile("" : "+v"(x)); return x; }

__device__ __forceinline__ void y_ring_put(FusedLds& L, int c, int i, float2 v) {
    const int s = i & (kFY - 1);
    L.y_ring[c][s] = v;
    if (s < kFYM) L.y_ring[c][s + kFY] = v;
}

// Barrier-free hand-over between the FLL loop waves and the helper wave.  LDS executes one wave's instructions in issue
// order, so "data stores, then counter store" on the producer and "counter load, then data loads" on the consumer need no
// fence, only that the compiler keeps the order (volatile + memory clobbers).  A wait that spins longer than any healthy
// run could (the peer is at most a few hundred clocks behind) raises `stuck` and every wait returns: wrong output, no hang.
// The wait is four instruction slots when the counter is already there (ds_read_b32, s_waitcnt, v_cmp, s_cbranch): hipcc
// unrolls and if-converts a C++ spin loop into dozens of instructions, so it is written out.  Every lane passes the LDS byte
// address of the counter IT depends on (the helper wave's upper and lower half follow different loop waves) and the wait
// ends when no lane is behind.
// A wave-wide LDS store / atomic to ONE address is executed lane after lane, which would hold up the LDS unit of the whole
// CU for every publication; so only lane 0 of the publishing wave addresses the counter, the other lanes address their own
// word of the dump area (ho_pub_addr).
__device__ __forceinline__ void ho_publish(unsigned pub_addr, int v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(pub_addr), "v"(v) : "memory");
}
__device__ __forceinline__ void ho_wait(unsigned ctr_addr, int need, unsigned stuck_addr) {
    int got, spins;
    asm volatile(
        "s_mov_b32 %1, 0\n"
        "1:\n"
        "ds_read_b32 %0, %2\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_cmp_lt_i32 vcc, %0, %3\n"
        "s_cbranch_vccz 3f\n"
        "s_add_u32 %1, %1, 1\n"
        "s_cmp_lt_u32 %1, 0x40000\n"
        "s_cbranch_scc0 2f\n"
        "ds_read_b32 %0, %4\n"                 // somebody else gave up: give up too
        "s_waitcnt lgkmcnt(0)\n"
        "v_cmp_eq_u32 vcc, 0, %0\n"
        "s_cbranch_vccnz 1b\n"
        "2:\n"
        "v_mov_b32 %0, 1\n"
        "ds_write_b32 %4, %0\n"
        "3:\n"
        : "=&v"(got), "=&s"(spins)
        : "v"(ctr_addr), "v"(need), "v"(stuck_addr)
        : "vcc", "scc", "memory");
}

#ifdef TETRA_DEMOD_DEBUG
#define FUSED_PROF_T0 const long long tb_ = PROF ? __builtin_readcyclecounter() : 0;
#define FUSED_PROF_T1 if (PROF) busy_ += __builtin_readcyclecounter() - tb_;
#else
#define FUSED_PROF_T0
#define FUSED_PROF_T1
#endif

// Every role runs its own copy of the epoch loop (same trip count, one workgroup barrier per epoch): the
// branch on the wave index is wave-uniform, and keeping the roles in separate code paths keeps each role's
// registers out of the others' live ranges.
#define FUSED_EPOCHS(...)                                                        \
    for (int e = 0; e < ntiles + 4; e++) {                                       \
        FUSED_PROF_T0                                                            \
        __VA_ARGS__                                                              \
        FUSED_PROF_T1                                                            \
        __syncthreads();                                                         \
    }

template <bool ALPHA0, bool QUALITY, bool PROF = false> __global__ __launch_bounds__(kFThreads) void k_fused(FusedParams p) {
    __shared__ FusedLds L;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int ch0 = blockIdx.x * kFCh;
    const int n = p.n;
    const int ntiles = (n + kFT - 1) / kFT;
    long long busy_ = 0;
    (void)busy_;
    const long long t_entry_ = PROF ? __builtin_readcyclecounter() : 0;
    auto chan = [&](int c) { const int ch = ch0 + c; return ch < p.n_channels ? ch : p.n_channels - 1; };
    auto live = [&](int c) { return ch0 + c < p.n_channels; };

    // ---- prologue, phase 1: tables and delay lines into LDS ----------------------------------------
    for (int i = tid; i < (kInterpPhases + 2) * kInterpTaps; i += kFThreads) {
        int row = i / kInterpTaps - 1;
        row = row < 0 ? 0 : (row > kInterpPhases - 1 ? kInterpPhases - 1 : row);
        L.bank[i] = p.bank[row * kInterpTaps + i % kInterpTaps];
    }
    if (tid < kRrcExt) L.rrc[tid] = p.rrc_ext[tid];
    if (tid < kPadBe) { L.be84[0][tid] = p.be_re84[tid]; L.be84[1][tid] = p.be_im84[tid]; }
    // rings start at zero: FIR windows touch slots that were never written (weighted by zero taps, so they must be finite)
    for (int i = tid; i < kFCh * kFXS; i += kFThreads) (&L.x_ring[0][0])[i] = make_float2(0.f, 0.f);
    for (int i = tid; i < kFCh * kFYS; i += kFThreads) (&L.y_ring[0][0])[i] = make_float2(0.f, 0.f);
    if (tid < kFCh) L.s_avail[tid] = 0;
    // the helper's counter starts 96 samples back: its pipeline rebuild runs three tiles (samples -96 .. -1)
    if (tid == 0) { L.x_done[0] = 0; L.x_done[1] = 0; L.f_done = kNearTaps - 96; L.stuck = 0; }
    __syncthreads();
    for (int i = tid; i < kFCh * kHist; i += kFThreads) {
        const int c = i / kHist, m = i % kHist;
        L.x_ring[c][(m - kHist) & (kFX - 1)] = p.hist[(long long)chan(c) * kHist + m];
    }
    for (int i = tid; i < kFCh * (kInterpTaps - 1); i += kFThreads) {
        const int c = i / (kInterpTaps - 1), m = i % (kInterpTaps - 1);
        y_ring_put(L, c, m - (kInterpTaps - 1), p.ybuf[(long long)chan(c) * (kInterpTaps - 1) + m]);
    }
    __syncthreads();

    if (wave == kRoleA) {
        // ---- AGC: lane c < 16 owns channel c; tile e in epoch e --------------------------------------
        const bool on = lane < kFCh;
        const int c = on ? lane : 0;
        float g = p.agc_g[chan(c)];
        const float2* in = p.iq + (long long)chan(c) * p.in_ch_stride;
        Pair<float> buf[kFT];   // slot s: sample s of the tile about to be processed (prefetched a tile ahead)
#pragma unroll
        for (int s = 0; s < kFT; s++) {
            buf[s] = Pair<float>(0.f, 0.f);
            if (on && s < n) buf[s] = ld_pair(in + (long long)s * p.in_t_stride);
        }
        __syncthreads();
        FUSED_EPOCHS(
            if (e < ntiles && on) {
                const int base = e * kFT;
                float2* dst = &L.a_buf[e & 1][c][0];
                if (base + 2 * kFT <= n) {
                    // this tile and the next are complete: no per-sample bookkeeping
                    _Pragma("unroll")
                    for (int s = 0; s < kFT; s++) {
                        const Pair<float> x = buf[s];
                        buf[s] = ld_pair(in + (long long)(base + kFT + s) * p.in_t_stride);
                        const Pair<float> a = agc_step<float>(p.k1, x, g);
                        dst[s] = make_float2(a.x(), a.y());
                    }
                } else {
                    _Pragma("unroll")
                    for (int s = 0; s < kFT; s++) {
                        const Pair<float> x = buf[s];
                        const int inext = base + kFT + s;
                        if (inext < n) buf[s] = ld_pair(in + (long long)inext * p.in_t_stride);
                        if (base + s < n) {
                            const Pair<float> a = agc_step<float>(p.k1, x, g);
                            dst[s] = make_float2(a.x(), a.y());
                        }
                    }
                }
            }
        )
        if (on && live(c)) p.agc_g[ch0 + c] = g;
    } else if (wave == kRoleL0 || wave == kRoleL1) {
        // ---- FLL loop wave: lane -> (row r = lane>>4, pos = (lane&15)>>1, parity = lane&1); tile e-1 in epoch e ---
        const int fw = wave - kRoleL0;
        const int f_pos = (lane & 15) >> 1;
        const int f_c = fw * 8 + (lane >> 4) * 2 + (lane & 1);
        FllNear8<float> R;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int kp = kFarTaps + 2 * (7 - f_pos) + j;
            R.ta[j] = p.be_re84[kp];
            R.tb[j] = p.be_im84[kp];
        }
        R.ph = p.fll_ph[chan(f_c)];
        R.fr = p.fll_fr[chan(f_c)];
        K1Consts k1 = p.k1;
        k1.fll_max_freq = v_pin(k1.fll_max_freq);
        const unsigned xd = lane == 0 ? lds_addr(&L.x_done[fw]) : lds_addr(&L.dump[0]) + 4u * (unsigned)lane;   // publishing address
        const unsigned fd = lds_addr(&L.f_done);
        const unsigned stuck = lds_addr(&L.stuck);
        lds_cfloat4* const frow = (lds_cfloat4*)(size_t)lds_addr(&L.f_ring[f_c][0]);
        lds_float2* const xrow = (lds_float2*)(size_t)lds_addr(&L.x_ring[f_c][0]);
        __syncthreads();      // (the helper wave has rebuilt its pipeline and published F_0 .. F_15)
        // rebuild the 8 in-flight near sums by replaying the last 16 stored samples
        R.clear_pipeline();
        for (int s = -kNearTaps; s < 0; s++) {
            const vfloat2 xv = xrow[s & (kFX - 1)];
            vfloat4 f = { 0.f, 0.f, 0.f, 0.f };
            if (s + kFarLead >= 0) f = frow[(s + kFarLead) & (kFF - 1)];
            R.template step<true, true>(k1, Pair<float>(xv.x, xv.y), Pair<float>(f.x, f.y), Pair<float>(f.z, f.w));
        }
        __syncthreads();      // epoch 0: nothing to do yet
        // every complete tile of the call: the assembly block of fll_asm.inc (FllNear8<float>::step<false, true> x 32 per
        // tile, one barrier per tile = epochs 1 .. nfull)
#ifndef TETRA_FLL_CPP_LOOP
        const int nfull = n / kFT;
#else
        const int nfull = 0;      // verification build: every tile through the C++ form of the step
#endif
        if (nfull > 0) {
            const bool head = f_pos == 0;
            const unsigned x_base = head ? lds_addr(&L.x_ring[f_c][0]) : lds_addr(&L.dump[0]) + 8u * (unsigned)lane;
            const unsigned headmask = head ? 0xffffffffu : 0u;
            const unsigned a_addr = lds_addr(&L.a_buf[0][f_c][0]);
            const unsigned f_addr = lds_addr(&L.f_ring[f_c][0]);
            int base_ = 0, tiles_ = nfull, st_, spins_;
            const int need0 = kNearTaps;          // first check (end of step 2 of tile 0): F_12 .. F_15 <=> f_done >= 16
            vfloat2 xs_ = { R.xs.x(), R.xs.y() }, r14_ = { R.r14.x(), R.r14.y() }, r32_ = { R.r32.x(), R.r32.y() };
            const vfloat2 ta_ = { R.ta[0], R.ta[1] }, tb_ = { R.tb[0], R.tb[1] };
            float ph_ = R.ph, fr_ = R.fr;
            const unsigned long long p4 = (unsigned long long)__builtin_bit_cast(unsigned, 0.4f);
            asm volatile(FLL_LOOP_ASM
                         : [ph] "+v"(ph_), [fr] "+v"(fr_), [r14] "+v"(r14_), [r32] "+v"(r32_), [xs] "+v"(xs_),
                           [base] "+s"(base_), [tiles] "+s"(tiles_), [st] "=&s"(st_), [spins] "=&s"(spins_)
                         : [ta] "v"(ta_), [tb] "v"(tb_), [need] "v"(need0), [a_addr] "v"(a_addr), [f_addr] "v"(f_addr),
                           [x_base] "v"(x_base), [headmask] "v"(headmask), [xd_addr] "v"(xd), [fd_addr] "v"(fd),
                           [stuck_addr] "v"(stuck), [maxf] "v"(k1.fll_max_freq), [two] "v"(2),
                           [negc1] "s"(-3.140625f), [beta] "s"(k1.fll_beta), [minf] "s"(k1.fll_min_freq),
                           [absmask] "s"(0x7fffffff), [pi] "s"(kFlPi), [p4] "s"(p4)
                         : "vcc", "scc", "memory", FLL_LOOP_CLOBBERS);
            R.xs = Pair<float>(xs_.x, xs_.y); R.r14 = Pair<float>(r14_.x, r14_.y); R.r32 = Pair<float>(r32_.x, r32_.y);
            R.ph = ph_; R.fr = fr_;
        }
        // the partial tile at the end of the call (same step, C++ form) and the trailing epochs
        for (int e = nfull + 1; e < ntiles + 4; e++) {
            const int t = e - 1;
            if (t < ntiles) {
                const int base = t * kFT;
                const int cnt = (n - base < kFT) ? (n - base) : kFT;
                lds_cfloat2* a_tile = (lds_cfloat2*)(size_t)lds_addr(&L.a_buf[t & 1][f_c][0]);
                for (int s = 0; s < cnt; s++) {
                    if ((s & 3) == 0) ho_wait(fd, base + s + kFarLead + 4, stuck);      // F_{s+8} .. F_{s+11}
                    const vfloat2 av = a_tile[s];
                    const vfloat4 f = frow[(base + s + kFarLead) & (kFF - 1)];
                    R.template step<false, ALPHA0>(k1, Pair<float>(av.x, av.y), Pair<float>(f.x, f.y), Pair<float>(f.z, f.w));
                    // lane (pos) holds x_{base+s-pos}: every position rewrites its sample (the older ones unchanged)
                    xrow[(base + s - f_pos) & (kFX - 1)] = vfloat2{ R.xs.x(), R.xs.y() };
                    if ((s & 1) == 1) ho_publish(xd, base + s + 1);
                }
                ho_publish(xd, base + kFT);      // lets the helper wave run its tile to the end
            }
            __syncthreads();
        }
        if (f_pos == 0 && live(f_c)) {
            p.fll_ph[ch0 + f_c] = R.ph;
            p.fll_fr[ch0 + f_c] = R.fr;
        }
    } else if (wave == kRoleH) {
        // ---- FLL helper wave: lane -> (row r = lane>>4, pos = (lane&15)>>2, c4 = lane&3), channel 4r + c4 -------
        // Its whole life is one assembly block (fll_asm.inc, generated by gen_fll_asm.py from the schedule of
        // FllFar4<float, 17>::step): three tiles that rebuild the in-flight far sums from the stored delay line (samples
        // -96 .. -1; x_i = 0 before -80, under zero taps) and leave F_0 .. F_15 in the ring, the barrier that ends the
        // prologue, epoch 0's barrier, then one tile + barrier per epoch.  A tile is always run to its 32nd sample (the loop
        // waves publish the whole tile at the end of a call's partial one; what the helper computes beyond is never used).
        const int h_pos = (lane & 15) >> 2;
        const int h_c = (lane >> 4) * 4 + (lane & 3);
        const unsigned x_row = lds_addr(&L.x_ring[h_c][0]);
        // only the head lanes hold completed far sums; the others' share of the ring store goes to the dump area
        const unsigned f_addr = h_pos == 0 ? lds_addr(&L.f_ring[h_c][0])
                                           : lds_addr(&L.dump[0]) + 16u * (unsigned)((lane >> 4) * 12 + (lane & 15) - 4);
        const unsigned xd_addr = lds_addr(&L.x_done[lane >> 5]);      // rows 0,1 follow loop wave 0, rows 2,3 loop wave 1
        const unsigned fd_addr = lane == 0 ? lds_addr(&L.f_done) : lds_addr(&L.dump[0]) + 4u * (unsigned)lane;          // publishing address
        const unsigned stuck_addr = lds_addr(&L.stuck);
        const unsigned tap_addr = lds_addr(&L.be84[0][kTH * (3 - h_pos)]);
#ifndef TETRA_FLL_CPP_HELPER
        {
            int base_, it_, st_, spins_;
            asm volatile(FLL_HELPER_ASM
                         : [base] "=&s"(base_), [it] "=&s"(it_), [st] "=&s"(st_), [spins] "=&s"(spins_)
                         : [iters] "s"(ntiles + 3), [x_row] "v"(x_row), [f_addr] "v"(f_addr), [xd_addr] "v"(xd_addr),
                           [fd_addr] "v"(fd_addr), [stuck_addr] "v"(stuck_addr), [tap_addr] "v"(tap_addr),
                           [one] "v"(1)
                         : "vcc", "scc", "memory", FLL_HELPER_CLOBBERS);
        }
        __syncthreads();
        __syncthreads();
        __syncthreads();
#else
        // Verification build (-DTETRA_FLL_CPP_HELPER): the same role from the C++ source that tests/emul compiles for the host.
        (void)tap_addr;
        FllFar4<float, kTH> F;
#pragma unroll
        for (int j = 0; j < kTH; j++) {
            const int kp = kTH * (3 - h_pos) + j;
            F.ta[j] = p.be_re84[kp];
            F.tb[j] = p.be_im84[kp];
        }
        lds_cfloat2* const xrow = (lds_cfloat2*)(size_t)x_row;
        lds_float4* const frow = (lds_float4*)(size_t)f_addr;
        F.clear_pipeline();
#define FUSED_H_STEP(S, I)                                                                         \
        {                                                                                          \
            const vfloat2 xv = xrow[(I) & (kFX - 1)];                                              \
            Pair<float> f14; Pair<float> f32;                                                      \
            F.template step<(S) & 15>(Pair<float>(xv.x, xv.y), f14, f32);                          \
            frow[((I) + kNearTaps) & (kFF - 1)] = vfloat4{ f14.x(), f14.y(), f32.x(), f32.y() };   \
        }
        for (int i0 = -kHist; i0 < 0; i0 += 16) {
            FUSED_H_STEP(0, i0) FUSED_H_STEP(1, i0 + 1) FUSED_H_STEP(2, i0 + 2) FUSED_H_STEP(3, i0 + 3)
            FUSED_H_STEP(4, i0 + 4) FUSED_H_STEP(5, i0 + 5) FUSED_H_STEP(6, i0 + 6) FUSED_H_STEP(7, i0 + 7)
            FUSED_H_STEP(8, i0 + 8) FUSED_H_STEP(9, i0 + 9) FUSED_H_STEP(10, i0 + 10) FUSED_H_STEP(11, i0 + 11)
            FUSED_H_STEP(12, i0 + 12) FUSED_H_STEP(13, i0 + 13) FUSED_H_STEP(14, i0 + 14) FUSED_H_STEP(15, i0 + 15)
        }
        ho_publish(fd_addr, kNearTaps);
        __syncthreads();
#define FUSED_H_PAIR(S)                                                                            \
        {                                                                                          \
            ho_wait(xd_addr, base + s0 + (S) + 2, stuck_addr);                                     \
            FUSED_H_STEP(S, base + s0 + (S))                                                       \
            FUSED_H_STEP((S) + 1, base + s0 + (S) + 1)                                             \
            ho_publish(fd_addr, base + s0 + (S) + 2 + kNearTaps);                                  \
        }
        FUSED_EPOCHS(
            const int t = e - 1;
            if (t >= 0 && t < ntiles) {
                const int base = t * kFT;
                for (int s0 = 0; s0 < kFT; s0 += 16) {
                    FUSED_H_PAIR(0) FUSED_H_PAIR(2) FUSED_H_PAIR(4) FUSED_H_PAIR(6)
                    FUSED_H_PAIR(8) FUSED_H_PAIR(10) FUSED_H_PAIR(12) FUSED_H_PAIR(14)
                }
            }
        )
#undef FUSED_H_PAIR
#undef FUSED_H_STEP
#endif
    } else if (wave == kRoleC) {
        // ---- RRC: lane -> (channel c = lane & 15, j = lane >> 4), outputs base + 8j + m; tile e-2 -------
        // The window of eight consecutive outputs starts at x_{i0-(nt-1)}; it is widened at the old end (under zero
        // taps) to start on a multiple of 8, so that no 8-sample chunk straddles the ring's wrap.
        const int c = lane & 15;
        const int rrc_pad = (8 - ((p.ntaps - 1) & 7)) & 7;
        const int rrc_chunks = (p.ntaps - 1 + rrc_pad) / 8 + 1;
        const unsigned x_base = pin_u32(lds_addr(&L.x_ring[c][0]));
        __syncthreads();
        FUSED_EPOCHS(
            const int t = e - 2;
            if (t >= 0 && t < ntiles) {
                const int i0 = t * kFT + 8 * (lane >> 4);
                if (i0 < n) {
                    const int start = i0 - (p.ntaps - 1) - rrc_pad;
                    Pair<float> out[kRrcOut];
                    rrc_direct8(rrc_chunks,
                                [&](int q) {      // q = 8*ck + j: chunk base wrapped, j added as an immediate offset
                                    lds_cfloat2* xw = (lds_cfloat2*)(size_t)(x_base + (((start + (q & ~7)) & (kFX - 1)) << 3));
                                    const vfloat2 v = xw[q & 7];
                                    return Pair<float>(v.x, v.y); },
                                [&](int q) { const float4 t4 = reinterpret_cast<const float4*>(L.rrc)[q];
                                             Tap4 r; r.v[0] = t4.x; r.v[1] = t4.y; r.v[2] = t4.z; r.v[3] = t4.w; return r; }, out);
                    _Pragma("unroll")
                    for (int m = 0; m < kRrcOut; m++) {
                        if (i0 + m < n) {
                            const float2 v = make_float2(out[m].x(), out[m].y());
                            y_ring_put(L, c, i0 + m, v);
                            if (p.y_dbg && live(c))
                                p.y_dbg[(long long)(kInterpTaps - 1 + i0 + m) * p.n_channels + ch0 + c] = v;
                        }
                    }
                }
            }
        )
    } else if (wave == kRoleD) {
        // ---- timing recovery: lane c < 16 owns channel c; consumes y of tiles <= e-3 ---------------------
        const bool on = lane < kFCh;
        const int c = on ? lane : 0;
        K2State st;
        st.mu = p.mu[chan(c)];
        st.omega = p.omega[chan(c)];
        st.offset = p.offset[chan(c)];
        st.cph = 0; st.cfr = 0; st.ph2 = 0; st.prev = 0;
        int S = 0;
        const int sym_cap = (int)(p.bits_stride / 2);
        const unsigned y_base = pin_u32(lds_addr(&L.y_ring[c][0]));
        const unsigned bank_base = pin_u32(lds_addr(&L.bank[0]));
        K2Consts k2 = p.k2;
        k2.tr_max_freq = v_pin(k2.tr_max_freq);
        __syncthreads();
        FUSED_EPOCHS(
            if (e >= 3 && on) {
                const int avail = (e - 2) * kFT;
                const int limit = avail < n ? avail : n;
                auto one_symbol = [&]() {
                    const int phase = k2_phase(st.mu);
                    // window buffer[offset .. offset+7] (contiguous thanks to the ring's mirror) and bank rows
                    // max(phase-1,0), phase, min(phase+1,127) = 24 contiguous floats of the padded table
                    lds_cfloat2* yw = (lds_cfloat2*)(size_t)(y_base + (((st.offset - (kInterpTaps - 1)) & (kFY - 1)) << 3));
                    lds_cfloat4* bk = (lds_cfloat4*)(size_t)(bank_base + (phase << 5));
                    Pair<float> w[kInterpTaps]; float t0[kInterpTaps]; float tm1[kInterpTaps]; float tp1[kInterpTaps];
                    _Pragma("unroll")
                    for (int j = 0; j < kInterpTaps; j++) {
                        const vfloat2 wv = yw[j];
                        w[j] = Pair<float>(wv.x, wv.y);
                    }
                    vfloat4 q;
                    q = bk[0]; tm1[0] = q.x; tm1[1] = q.y; tm1[2] = q.z; tm1[3] = q.w;
                    q = bk[1]; tm1[4] = q.x; tm1[5] = q.y; tm1[6] = q.z; tm1[7] = q.w;
                    q = bk[2]; t0[0] = q.x; t0[1] = q.y; t0[2] = q.z; t0[3] = q.w;
                    q = bk[3]; t0[4] = q.x; t0[5] = q.y; t0[6] = q.z; t0[7] = q.w;
                    q = bk[4]; tp1[0] = q.x; tp1[1] = q.y; tp1[2] = q.z; tp1[3] = q.w;
                    q = bk[5]; tp1[4] = q.x; tp1[5] = q.y; tp1[6] = q.z; tp1[7] = q.w;
                    float vr; float vi;
                    k2_timing(k2, st, phase, w, tm1, t0, tp1, &vr, &vi);
                    L.s_ring[c][S & (kFS - 1)] = make_float2(vr, vi);
                    S++;
                };
                // Output capacity guard.  Every symbol advances the offset by >= 1 sample (k2_timing), so this
                // epoch adds at most limit - offset symbols.  If even that fits the output row the loop runs
                // unchecked (always the case for a finite stream except near the end of very short calls);
                // otherwise it checks per symbol, and a NaN/Inf-poisoned channel whose loop has stopped advancing
                // properly is cut off at the row capacity instead of overrunning it.
                if (S + (limit - st.offset) <= sym_cap) {
                    while (st.offset < limit) one_symbol();
                } else {
                    while (st.offset < limit) {
                        if (S >= sym_cap) { st.offset = limit; break; }
                        one_symbol();
                    }
                }
                L.s_avail[c] = S;
            }
        )
        if (on && live(c)) {
            p.mu[ch0 + c] = st.mu;
            p.omega[ch0 + c] = st.omega;
            p.offset[ch0 + c] = st.offset - n;          // complex_fd.cpp:145
        }
    } else if (wave == kRoleNone) {
        __syncthreads();
        FUSED_EPOCHS()
    } else {
        // ---- kRoleE: Costas + slicer + differential decoder + bit unpacker; symbols published before e ----
        const bool on = lane < kFCh;
        const int c = on ? lane : 0;
        K2State st;
        st.mu = 0; st.omega = 0; st.offset = 0;
        st.cph = p.cph[chan(c)];
        st.cfr = p.cfr[chan(c)];
        st.ph2 = p.ph2[chan(c)];
        st.prev = p.prev[chan(c)];
        int S = 0;
        uint8_t* brow = p.bits + (long long)chan(c) * p.bits_stride;
        float2* srow = p.sym ? p.sym + (long long)chan(c) * (p.bits_stride / 2) : nullptr;
        const bool wr = on && live(c);
        const bool qon = QUALITY;   // compile-time: the statistic's code must not weigh on the default kernel
        QualityState q;
        q.sum = 0.0; q.ptr = 0; q.disp = 0; q.standarderr = 0.0f; q.sync = 0;
        float* qring = nullptr;
        if (qon) {
            q.sum = p.q_sum[chan(c)]; q.ptr = p.q_ptr[chan(c)]; q.disp = p.q_disp[chan(c)];
            q.standarderr = p.q_err[chan(c)]; q.sync = p.q_sync[chan(c)];
            qring = p.q_ring + (long long)chan(c) * 4096;
        }
        K2Consts k2 = p.k2;
        k2.costas_max_freq = v_pin(k2.costas_max_freq);
        __syncthreads();
        FUSED_EPOCHS(
            if (e >= 4 && on) {
                const int avail = L.s_avail[c];
                while (S < avail) {
                    const float2 v = L.s_ring[c][S & (kFS - 1)];
                    float zr; float zi;
                    const int d = k2_costas(k2, st, v.x, v.y, &zr, &zi);
                    if (wr) {
                        // bit_unpacker.cpp:6-7: byte 2S = MSB, byte 2S+1 = LSB
                        *reinterpret_cast<unsigned short*>(brow + 2 * S) = (unsigned short)(((d >> 1) & 1) | ((d & 1) << 8));
                        if (srow) srow[S] = make_float2(zr, zi);
                        if (qon) quality_step(q, qring, zr, zi);
                    }
                    S++;
                }
            }
        )
        if (wr) {
            p.cph[ch0 + c] = st.cph;
            p.cfr[ch0 + c] = st.cfr;
            p.ph2[ch0 + c] = st.ph2;
            p.prev[ch0 + c] = st.prev;
            p.n_bits[ch0 + c] = 2 * S;
            if (qon) {
                p.q_sum[ch0 + c] = q.sum; p.q_ptr[ch0 + c] = q.ptr; p.q_disp[ch0 + c] = q.disp;
                p.q_err[ch0 + c] = q.standarderr; p.q_sync[ch0 + c] = q.sync;
            }
        }
    }
#ifdef TETRA_DEMOD_DEBUG
    if (PROF && lane == 0) {
        p.prof[(long long)blockIdx.x * 8 + wave] = busy_;
        if (wave == 0) p.prof[(long long)blockIdx.x * 8 + 7] = __builtin_readcyclecounter() - t_entry_;
    }
#else
    (void)t_entry_;
#endif
    // delay lines: last 80 FLL outputs, last 7 RRC outputs (both rings still hold them; the loops end on a barrier)
    for (int i = tid; i < kFCh * kHist; i += kFThreads) {
        const int c = i / kHist, m = i % kHist;
        if (live(c)) p.hist[(long long)(ch0 + c) * kHist + m] = L.x_ring[c][(n - kHist + m) & (kFX - 1)];
    }
    for (int i = tid; i < kFCh * (kInterpTaps - 1); i += kFThreads) {
        const int c = i / (kInterpTaps - 1), m = i % (kInterpTaps - 1);
        if (live(c))
            p.ybuf[(long long)(ch0 + c) * (kInterpTaps - 1) + m] = L.y_ring[c][(n - (kInterpTaps - 1) + m) & (kFY - 1)];
    }
}
#undef FUSED_EPOCHS
#undef FUSED_PROF_T0
#undef FUSED_PROF_T1

}  // namespace
