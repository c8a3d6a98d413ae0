// kernel_fused.hpp -- the whole demodulator chain of 16 channels in ONE workgroup of six specialised
// wavefronts (included by tetra_demod.hip; device code only).
//
// Why this shape (measured on MI355X, profiles/r02/r02_a_issue_model.md, r02_b_*): a gfx950 wavefront issues one
// instruction of ANY kind (VALU, SALU, s_nop, s_waitcnt, LDS) per ~4.7 clocks whether or not it depends on the previous
// one, and a SIMD's issue capacity is shared by its waves (a packed-FP32 or DPP instruction occupies it for ~4.4 clocks,
// a plain VALU one for ~2.3).  A role's time is therefore its instruction count, the chain's pace is the instruction
// count of its busiest wave, and the sum over the waves of a SIMD must fit too.  With 4096 channels there are 16 channels
// per CU; the loop code (NCO sincos, error functions, loop filters) costs the same whether a wave carries 4 or 64
// channels.  A CU's 16 channels are therefore split by STAGE, each stage at the widest lane occupancy its recurrence
// allows:
//
//   wave  role                                  lanes/channel        instruction slots / sample       SIMD
//   F0,F1 FLL: NCO, band-edge FIRs, loop         8 (interleaved)      57 (hand-scheduled, fll_asm.inc) 2, 3: one each, alone
//   E     Costas + slicer + diff. decoder + out  1, then 4 (2 passes) ~42                              0 (older wave)
//   C     RRC matched filter (time-parallel)     4 x 8 outputs        ~24                              0
//   D     ML timing recovery                     4 (one row each)     ~35                              1 (older wave)
//   A     AGC                                    1                    ~25                              1
//
// The issue arbiter serves the OLDEST wave of a SIMD first, so the recurrence-bound roles take the lower wave index of
// their SIMD and the throughput roles fill the slots they leave.  (Cutting the FLL in two -- loop waves plus a helper wave
// for the far taps -- was built and measured in round 2: it shortens the longest wave but adds work and a seventh role that
// does not pack into four SIMDs, profiles/r02/r02_c_fll_split_experiment.md; round 3 bounded ANY re-homing of the FIR work --
// helper wave, matrix pipe -- by ablation: nothing to gain, profiles/r03/r03_b_matrix_pipe_and_coresidency.md; round 4 measured why the
// matrix pipe cannot help an exact-f32 chain: v_mfma_f32_*_f32 occupies the vector ALUs for its passes, profiles/r04/r04_l_*.)
// Two more shapes of the same template: 32 channels in eight waves (FLL rows of 4 lanes per channel: more than 16 channels per
// CU) and 4 channels (FLL rows of 16 lanes: at most 4 channels per CU; its LONG variant -- rows of 16 x 9 taps, 128 delay-line
// samples -- takes filters of 73 .. 129 taps at any channel count); see Roles<CH> below and DESIGN.md section 4.2.
//
// Stages are connected by LDS rings (AGC out -> FLL out x -> RRC out y -> symbols) and run as a software pipeline over
// 32-sample tiles with one workgroup barrier per tile: in epoch e, A works on tile e, F on e-1, C on e-2, D consumes y
// of tiles <= e-3 and E the symbols D published before the epoch.  No intermediate touches HBM: the kernel reads 8 B
// and writes 1 B per input sample.
#pragma once

#include "fll_asm.inc"
#include "fll4_asm.inc"
#include "fll16_asm.inc"
#include "fll16l_asm.inc"
#include "fll8l_asm.inc"

namespace {

constexpr int kFT = 32;                  // samples per tile (pipeline epoch)
constexpr int kFAS = kFT + 2;            // row stride of the AGC output buffer: 34 float2 = 68 dwords, so that the 8 (16, 4) channels an FLL
                                         // wave reads with one ds_read_b128 sit 4 banks apart instead of on the same 4 banks
constexpr int kFCh = 16;                 // channels per workgroup: the benchmark's shape (4096 channels = one workgroup per CU) ...
constexpr int kFChWide = 32;             // ... the wide shape for more than 16 channels per CU (FLL rows of 4 lanes per channel) ...
constexpr int kFChSmall = 4;             // ... and the small shape for at most 4 channels per CU (FLL rows of 16 lanes per channel)
#ifndef TETRA_WIDE_WAVES
#define TETRA_WIDE_WAVES 8
#endif
#ifndef TETRA_NARROW_WAVES
#define TETRA_NARROW_WAVES 6             // (experiment builds: 8 = two more waves without a role, so that the role table can leave SIMD slots empty)
#endif
constexpr int fused_threads(int ch) { return ch == 16 ? TETRA_NARROW_WAVES * 64 : ch == 4 ? 8 * 64 : TETRA_WIDE_WAVES * 64; }   // 16: six roles; 32, 4: see Roles
constexpr int kFThreads = fused_threads(kFCh);       // 384 = 6 waves
// Ring depths.  The TETRA_EXP_* overrides exist for TIMING-ONLY experiment builds (profiles/build_exp.sh: rings too short to
// hold the data, output garbage, same instruction streams); the product is built without them.
#ifndef TETRA_EXP_XRING
#define TETRA_EXP_XRING 256
#endif
#ifndef TETRA_EXP_YRING
#define TETRA_EXP_YRING 128
#endif
#ifndef TETRA_EXP_SRING
#define TETRA_EXP_SRING 64
#endif
constexpr int kFX = TETRA_EXP_XRING;     // x ring (FLL output) per channel ...
constexpr int kFXP = 8;                  // ... behind 8 slots of front padding: an FLL lane stores x_{i-pos} at slot i - pos of the
                                         // tile's window without wrapping (slots -7 .. -1 are never read, see fll_asm.inc)
constexpr int kFXS = kFXP + kFX + 1;     // row stride (odd: spreads channels over LDS banks)
constexpr int kFY = TETRA_EXP_YRING;     // y ring (RRC output) per channel ...
constexpr int kFYM = 8;                  // ... plus a mirror of the first slots so the interpolator window never wraps
constexpr int kFYS = kFY + kFYM + 1;
constexpr int kFS = TETRA_EXP_SRING;     // symbol ring per channel: two epochs' worth of symbols while every symbol advances >= 1 sample
constexpr int kFSDeep = 256;             // ... and for timing loops that may emit several symbols from one offset (floor(mu) = 0,
                                         // complex_fd.cpp:141-143): an epoch's 32 samples then carry up to 33 / min_step + 1 symbols, the
                                         // Costas wave runs one epoch behind, so the ring holds 2 (33 / min_step + 1) <= 256 for
                                         // min_step >= kMinStepDeep (design.hpp: 0.27 samples per symbol).  16- and 4-channel shapes only.
constexpr int kFSDeeper = 1024;          // ... and the 4-channel shape's second level: 2 (33 / min_step + 1) <= 1024 for min_step >= kMinStepDeeper
                                         // (0.07: up to fifteen symbols from one offset)
static_assert(kF8Pad == 72 && kF8Taps == 9, "fll_asm.inc is generated for 8 positions x 9 taps");
static_assert(kF4Pad == 68 && kF4Taps == 17, "fll4_asm.inc is generated for 4 positions x 17 taps");
static_assert(kF16Pad == 80 && kF16Taps == 5, "fll16_asm.inc is generated for 16 positions x 5 taps");
static_assert(kF16LPad == 144 && kF16LTaps == 9 && kBePadLong == 144, "fll16l_asm.inc is generated for 16 positions x 9 taps, tables of 144");
static_assert(kF8LPad == 136 && kF8LTaps == 17, "fll8l_asm.inc is generated for 8 positions x 17 taps, tables of 144");

// Wave index of each role, in the order E, D, F0, F1, A, C.  A workgroup's waves go to the CU's four SIMDs cyclically and
// the OLDER wave of a SIMD is served first, so this table decides who shares a SIMD with whom and who has priority there:
//   SIMD0 {E (wave 0), A (wave 4)}   SIMD1 {D (wave 1), C (wave 5)}   SIMD2 {F0}   SIMD3 {F1}
// The symbol-rate recurrences (E, D) keep priority.  Which of the two gets the packed-FP32-heavy RRC wave for a neighbour was
// measured twice on one box: in round 2 (Costas wave 52 slots per sample in one pass, timing wave 45 on one lane per channel:
// profiles/r02/r02_j_role_placement.md) {E,C}{D,A} 4.41 ms, {E,A}{D,C} 4.49, the pairs with priorities flipped 5.23,
// {E,D}{A,C} 5.02; on round 3's waves (Costas 42 in two passes, timing 35 on four lanes per channel) the RRC wave fits beside
// the lightened timing wave and the Costas wave keeps SIMD0 nearly to itself: {E,A}{D,C} 3.86 ms, {E,C}{D,A} 3.98
// (profiles/r03/r03_p2_exp.log) -- the launch then runs at the FLL waves' own pace.
#ifndef TETRA_ROLE_IDS
#define TETRA_ROLE_IDS 0, 1, 2, 3, 4, 5
#endif
namespace role_ids { constexpr int v[6] = { TETRA_ROLE_IDS }; }
// The wide workgroup (32 channels): the two FLL waves carry 16 channels each on rows of 4 lanes per channel (75 slots per
// sample for 16 channels instead of 59 for 8: the loop code -- NCO, error, loop filter -- is shared by twice the channels)
// and have a SIMD to themselves (waves 0, 1; waves 4 and 5 would land beside them and only keep the barriers).  The other
// four roles share the remaining two SIMDs: E (older) and D on one, the RRC wave (older, two passes per tile) and the AGC
// wave on the other.  Measured at 8192 x 36000, alternating on one box (profiles/r02/r02_q_wide_fll4.md): this placement
// 5.34 ms; {E,C}{D,A} 6.36; {E,A}{D,C} 6.41; D older than E 5.63; AGC older than RRC 6.00; RRC as two waves beside E and D
// with the AGC wave beside an FLL wave 5.75; twelve waves ({E,D,-}{C0,C1,A}) 5.36.  The FLL waves' floor is 75.19 x 4.65 =
// 350 clocks per sample = 5.25 ms.
#ifndef TETRA_ROLE_IDS_WIDE
#define TETRA_ROLE_IDS_WIDE 2, 6, 0, 1, 7, 3, -1       // E, D, F0, F1, A, C, C2 (second RRC wave: one pass each; -1 = none)
#endif
namespace role_ids { constexpr int w[7] = { TETRA_ROLE_IDS_WIDE }; }
// The small workgroup (4 channels; for at most 4 channels per CU, i.e. up to 1024 channels per GPU: BASELINE configs 2 and 5):
// ONE FLL wave whose rows spend all 16 lanes on a channel (16 positions x 5 taps: 10 tap FMAs per step instead of 18, the
// shortest FLL step, 52.56 slots) alone on SIMD0, the Costas wave alone on SIMD1, the timing wave alone on SIMD2, RRC (older)
// + AGC on SIMD3; waves 4..6 would land beside the three recurrences and only keep the barriers.
#ifndef TETRA_ROLE_IDS_SMALL
#define TETRA_ROLE_IDS_SMALL 1, 2, 0, -1, 7, 3        // E, D, F0, (no F1), A, C
#endif
namespace role_ids { constexpr int m[6] = { TETRA_ROLE_IDS_SMALL }; }
// LONG (4- and 16-channel shapes): FLL rows of 16 positions x 9 taps / 8 x 17 and deeper tables, for filters of 73 .. 129 taps.
template <int CH, bool LONG = false> struct Roles {
    static constexpr int E = CH == 16 ? role_ids::v[0] : CH == 4 ? role_ids::m[0] : role_ids::w[0],
                         D = CH == 16 ? role_ids::v[1] : CH == 4 ? role_ids::m[1] : role_ids::w[1],
                         F0 = CH == 16 ? role_ids::v[2] : CH == 4 ? role_ids::m[2] : role_ids::w[2],
                         A = CH == 16 ? role_ids::v[4] : CH == 4 ? role_ids::m[4] : role_ids::w[4],
                         C = CH == 16 ? role_ids::v[5] : CH == 4 ? role_ids::m[5] : role_ids::w[5],
                         C2 = CH == 32 ? role_ids::w[6] : -1, NF = CH == 4 ? 1 : 2;
    // FLL row geometry: lanes per channel, taps per lane, channels per FLL wave
    static constexpr int FL = CH == 16 ? kF8Lanes : CH == 4 ? kF16Lanes : kF4Lanes,
                         FT = CH == 16 ? (LONG ? kF8LTaps : kF8Taps) : CH == 4 ? (LONG ? kF16LTaps : kF16Taps) : kF4Taps, FCH = 64 / FL;
    static_assert(!LONG || CH == 4 || CH == 16, "the long rows exist for the 4- and 16-channel shapes");
    static_assert(NF * FCH == CH, "the FLL waves cover the workgroup's channels");
};

struct FusedParams {
    const float2* iq;
    long long in_ch_stride, in_t_stride;
    int n, n_channels;
    int ch_base;         // first channel of this launch (a call may be cut into a 32-channel-workgroup launch and a 16-channel one)
    // state
    float *agc_g, *fll_ph, *fll_fr;
    float2* hist;        // [C][kHist]
    int* rrc_valid;      // [C] how many of the newest delay-line samples the RRC may see (kHist = all; tetra_demod.h)
    float *mu, *omega;
    int* offset;
    float *cph, *cfr, *ph2;
    int* prev;
    float2* ybuf;        // [C][7]
    // tables
    const float* be_re80;   // band-edge taps zero-padded (old end) to kBePad = 80
    const float* be_im80;
    const float* rrc_ext;   // [kRrcExt] RRC taps as rrc_direct8 wants them: ext[7 + rrc_pad + k] = h[k], zero elsewhere
    int ntaps;
    const float* bank;
    // outputs
    uint8_t* bits;
    long long bits_stride;
    int* n_bits;
    float2* sym;         // optional [C][sym_stride]
    long long sym_stride;
    int* overruns;       // [1] channels whose output row filled up in this launch (the rest of their samples were dropped)
    float2* y_dbg;       // optional: time-major scratch [(7+n)][C], row 7+i = y_i
    K1Consts k1;
    K2Consts k2;
    long long* prof;     // instrumented (PROF) kernels of TETRA_DEMOD_DEBUG builds: [workgroups][8] = busy clocks of waves 0..5 inside
                         // their epoch bodies (barrier waits excluded), [7] = clocks from kernel entry to exit of wave 0; null = off.
                         // Every other kernel of the 16- and 32-channel shapes reads it as the CUT FLAG (below): one pointer, no new member
};
// The cut flag: optional int[1]; a cut-off channel also leaves a plain store of 1 there (the in-place host path points it into its
// mapped host block: a store crosses PCIe on every platform, an atomic may not).  Where it lives in the kernel arguments was
// chosen by measurement (profiles/r04/r04_f_param_layout.md): ANY member added in front of the loop constants k1 / k2 costs the
// 4-channel shape 2 % (3.40 against 3.33 ms per 36000 samples; same instruction counts, another register assignment); a member
// appended behind `prof` is the 4-channel shape's best (3.31) but costs the 16-channel shape 0.3-0.5 %; the 16- and 32-channel
// shapes are at their best with round 3's argument block unchanged.  So: the 4-channel kernels get the appended member, the
// others carry the pointer in `prof` (unused by every non-instrumented kernel).
template <int CH> struct FusedParamsT : FusedParams {};
template <> struct FusedParamsT<kFChSmall> : FusedParams { int* cut_flag4; };
// the long variant also carries the 48 delay-line samples in front of hist's 80 (tetra_demod_channel_state_t::hist_far)
template <int CH> struct FusedParamsLongT : FusedParamsT<CH> {
    float2* hist_far;    // [C][kHistLong - kHist]
    int far_valid;       // 0: not current (a kernel that does not carry them ran since they were written): zeros to every filter
};
template <int CH, bool LONG> struct FusedArgs { typedef FusedParamsT<CH> type; };
template <int CH> struct FusedArgs<CH, true> { typedef FusedParamsLongT<CH> type; };
template <int CH, bool PROF> __device__ __forceinline__ int* fused_cut_flag(const FusedParamsT<CH>& p) {
    if constexpr (CH == kFChSmall) return p.cut_flag4;
    else return PROF ? nullptr : reinterpret_cast<int*>(p.prof);
}

template <int CH, int DEEP = 0, bool LONG = false> struct FusedLdsT {
    static constexpr int kS = DEEP == 2 ? kFSDeeper : DEEP ? kFSDeep : kFS;
    static constexpr int kRE = LONG ? kRrcExtLong : kRrcExt, kBP = LONG ? kBePadLong : kBePad;
    float2 a_buf[2][CH][kFAS];
    float2 x_ring[CH][kFXS];
    float2 y_ring[CH][kFYS];
    float2 s_ring[CH][kS];      // (rows 512 bytes apart; padding them by one entry was measured: +1 % / +2 % at 4096 / 1024 channels,
                                 // -0.6 % at 8192, profiles/r03/r03_y_exp.log -- left as it is)
    int s_avail[CH];
    int e_span[2][CH];       // 4-channel workgroup: the symbols [first, end) the Costas wave's recurrence lane has just finished
    float2 e_last[2][CH];    // ... and, by epoch parity, z of the last symbol finished so far (the slicer's "previous symbol")
    // interpolator bank with row 0 repeated in front and row 127 behind: rows max(p-1,0), p, min(p+1,127) of
    // complex_fd.cpp:102-121 are then the 24 contiguous floats at bank[p * 8]
    __attribute__((aligned(16))) float bank[(kInterpPhases + 2) * kInterpTaps];
    __attribute__((aligned(16))) float rrc[kRE];           // zero-extended taps, see rrc_direct8
    float be80[2][kBP];      // band-edge taps (re, im), zero-padded at the old end: the FLL waves' assembly loads its taps from here
};
typedef FusedLdsT<kFCh> FusedLds;
static_assert(sizeof(FusedLdsT<kFCh>) <= 80 * 1024 && sizeof(FusedLdsT<kFChWide>) <= 160 * 1024 - 256 &&
              sizeof(FusedLdsT<kFChSmall>) <= 32 * 1024 && sizeof(FusedLdsT<kFCh, 1>) <= 104 * 1024 &&
              sizeof(FusedLdsT<kFChSmall, 1, true>) <= 40 * 1024 && sizeof(FusedLdsT<kFCh, 1, true>) <= 108 * 1024 &&
              sizeof(FusedLdsT<kFChSmall, 2, true>) <= 64 * 1024, "LDS budget of a CU");

// Typed LDS pointers built from a 32-bit LDS byte address.  Keeping the (loop-invariant) row base in one pinned vector
// register makes the compiler address a sliding window as `base register + immediate offsets` (ds_read2_b64 /
// ds_read_b128 with offset fields) instead of re-deriving every element's address from the LDS struct offset.
typedef float vfloat2 __attribute__((ext_vector_type(2)));
typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const vfloat2 lds_cfloat2;
typedef __attribute__((address_space(3))) const vfloat4 lds_cfloat4;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }
__device__ __forceinline__ unsigned pin_u32(unsigned x) { asm volatile("" : "+v"(x)); return x; }

template <class LDS> __device__ __forceinline__ void x_ring_put(LDS& L, int c, int i, float2 v) { L.x_ring[c][kFXP + (i & (kFX - 1))] = v; }
template <class LDS> __device__ __forceinline__ float2 x_ring_get(const LDS& L, int c, int i) { return L.x_ring[c][kFXP + (i & (kFX - 1))]; }
template <class LDS> __device__ __forceinline__ void y_ring_put(LDS& L, int c, int i, float2 v) {
    const int s = i & (kFY - 1);
    L.y_ring[c][s] = v;
    if (s < kFYM) L.y_ring[c][s + kFY] = v;
}

// LDS side of one FLL lane in the C++ form of the wave (see fll_tile / fll_replay in demod_core.hpp).
template <class LDS, class Row> struct FllDeviceIOT {
    LDS& L;
    const float2* a_tile;    // a_buf[parity][ch] of the tile being processed
    int c;                   // channel within the workgroup
    int pos;                 // position along the channel's lanes (0 = head)
    int tile_base;           // first sample index of the tile (replay: index of the first sample to come)

    // the delay line sits in the x ring: the Row::kReplay samples in front of tile_base
    __device__ __forceinline__ Pair<float> load_hist(int g) const {
        const float2 v = x_ring_get(L, c, tile_base - Row::kReplay + g * Row::kLanes + pos);
        return Pair<float>(v.x, v.y);
    }
    __device__ __forceinline__ Pair<float> sample(int s) const {
        const float2 v = a_tile[s];
        return Pair<float>(v.x, v.y);
    }
    __device__ __forceinline__ void xs_store(int iend, int cnt, Pair<float> xs) const {
        if (pos < cnt) x_ring_put(L, c, tile_base + iend - 1 - pos, make_float2(xs.x(), xs.y()));
    }
};

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

#ifndef TETRA_EXP_TWOPASS
#define TETRA_EXP_TWOPASS 0           // experiment builds: 1 = the two-pass Costas wave on the 32-channel shape too (product: 4 and 16 channels)
#endif
#ifndef TETRA_EXP_DPAIR
#define TETRA_EXP_DPAIR 1             // 32-channel shape: timing wave on two lanes per channel (0 = one lane per channel)
#endif
#ifndef TETRA_EXP_WAVES_PER_EU
#define TETRA_EXP_WAVES_PER_EU 1      // experiment builds: a larger value caps the VGPRs so that more waves fit a SIMD
#endif
// DEEP: the timing loop may emit several symbols from one offset (min_step < 1, see kFSDeep): deeper symbol ring, no forward-
// progress clamp in the timing step, the output-row check on every symbol.  Everything else is the same code.
// LONG (4- and 16-channel shapes): filters of 73 .. 129 taps -- FLL rows of 16 x 9 taps (fll16l_asm.inc) / 8 x 17 (fll8l_asm.inc), tap
// tables of 144 / 160 entries, 128 delay-line samples carried (hist + hist_far).  Everything else is the same code.
template <bool ALPHA0, bool PROF = false, int CH = kFCh, int DEEP = 0, bool LONG = false> __global__ __launch_bounds__(fused_threads(CH), TETRA_EXP_WAVES_PER_EU) void k_fused(typename FusedArgs<CH, LONG>::type p) {
    typedef FusedLdsT<CH, DEEP, LONG> Lds;
    constexpr int kH = LONG ? kHistLong : kHist;      // delay-line samples in front of the call
    constexpr int kSR = Lds::kS;          // symbol ring depth
    constexpr int kMinAdv = DEEP ? 0 : 1;
    static_assert(!DEEP || CH != kFChWide, "the deep symbol ring is instantiated for the 16- and 4-channel shapes");
    static_assert(DEEP < 2 || CH == kFChSmall, "... and its second level for the 4-channel shape");
    typedef FllRowT<float, Roles<CH, LONG>::FL, Roles<CH, LONG>::FT> FllRow;
    typedef FllDeviceIOT<Lds, FllRow> FllDeviceIO;
    typedef Roles<CH, LONG> R_;
    constexpr int kRoleE = R_::E, kRoleD = R_::D, kRoleF0 = R_::F0, kRoleA = R_::A, kRoleC = R_::C;
    constexpr int kThreadsCH = fused_threads(CH);
    __shared__ Lds L;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int ch0 = p.ch_base + blockIdx.x * CH;
    const int n = p.n;
    const int ntiles = (n + kFT - 1) / kFT;
    long long busy_ = 0;
    (void)busy_;
    const long long t_entry_ = PROF ? __builtin_readcyclecounter() : 0;
    auto chan = [&](int c) { const int ch = ch0 + c; return ch < p.n_channels ? ch : p.n_channels - 1; };
    auto live = [&](int c) { return ch0 + c < p.n_channels; };

    // ---- prologue, phase 1: tables and delay lines into LDS ----------------------------------------
    for (int i = tid; i < (kInterpPhases + 2) * kInterpTaps; i += kThreadsCH) {
        int row = i / kInterpTaps - 1;
        row = row < 0 ? 0 : (row > kInterpPhases - 1 ? kInterpPhases - 1 : row);
        L.bank[i] = p.bank[row * kInterpTaps + i % kInterpTaps];
    }
    if (tid < Lds::kRE) L.rrc[tid] = p.rrc_ext[tid];
    if (tid < Lds::kBP) { L.be80[0][tid] = p.be_re80[tid]; L.be80[1][tid] = p.be_im80[tid]; }
    // rings start at zero: FIR windows touch slots that were never written (weighted by zero taps, so they must be finite)
    for (int i = tid; i < CH * kFXS; i += kThreadsCH) (&L.x_ring[0][0])[i] = make_float2(0.f, 0.f);
    for (int i = tid; i < CH * kFYS; i += kThreadsCH) (&L.y_ring[0][0])[i] = make_float2(0.f, 0.f);
    if (tid < CH) L.s_avail[tid] = 0;
    __syncthreads();
    for (int i = tid; i < CH * kHist; i += kThreadsCH) {
        const int c = i / kHist, m = i % kHist;
        x_ring_put(L, c, m - kHist, p.hist[(long long)chan(c) * kHist + m]);
    }
    if constexpr (LONG) {
        constexpr int kFar = kHistLong - kHist;
        if (p.far_valid)
            for (int i = tid; i < CH * kFar; i += kThreadsCH) {
                const int c = i / kFar, m = i % kFar;
                x_ring_put(L, c, m - kHistLong, p.hist_far[(long long)chan(c) * kFar + m]);
            }
    }
    for (int i = tid; i < CH * (kInterpTaps - 1); i += kThreadsCH) {
        const int c = i / (kInterpTaps - 1), m = i % (kInterpTaps - 1);
        y_ring_put(L, c, m - (kInterpTaps - 1), p.ybuf[(long long)chan(c) * (kInterpTaps - 1) + m]);
    }
    __syncthreads();

    if (wave == kRoleA) {
        // ---- AGC: lane c < 16 owns channel c; tile e in epoch e --------------------------------------
        const bool on = lane < CH;
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
#if defined(TETRA_EXP_ABLATE) && TETRA_EXP_ABLATE == 3      // experiment builds only: AGC wave without its arithmetic
                        const Pair<float> a = x;
#else
                        const Pair<float> a = agc_step<float>(p.k1, x, g);
#endif
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
    } else if (wave >= kRoleF0 && wave < kRoleF0 + R_::NF) {
        // ---- FLL: lane -> (row r = lane>>4, pos = (lane&15) / hop, channel-in-row = lane % hop); tile e-1 in epoch e ---
        constexpr int kHop = FllRow::kHop;
        const int fw = wave - kRoleF0;
        const int f_pos = (lane & 15) / kHop;
        const int f_c = fw * R_::FCH + (lane >> 4) * kHop + (lane % kHop);
        float ph = p.fll_ph[chan(f_c)];
        float fr = p.fll_fr[chan(f_c)];
        K1Consts k1 = p.k1;
        k1.fll_max_freq = v_pin(k1.fll_max_freq);
        __syncthreads();
        __syncthreads();      // epoch 0: nothing to do yet
        // Every COMPLETE tile of the call runs in the assembly block of fll_asm.inc / fll4_asm.inc (generated by gen_fll_asm.py
        // from the schedule of FllRowT<float>::step): rebuild of the in-flight sums from the delay line in front of the call,
        // then 32 steps and one barrier per tile = epochs 1 .. nfull.  The alpha != 0 variant of the loop filter (never
        // produced by the reference's FLL, fll.cpp:25) takes the C++ form for every tile; the debug build does not
        // instrument the block.
        const int nfull = ALPHA0 ? n / kFT : 0;
        // padded tap kp of the row sits at be80[kp + 80 - LANES * TAPS] (both are padded at the old end)
        constexpr int kTapOff = Lds::kBP - FllRow::kLanes * FllRow::kTaps;
        if (nfull > 0) {
            int base_ = 0, tiles_ = nfull, st_;
            const unsigned a_addr = lds_addr(&L.a_buf[0][f_c][0]);
            const unsigned x_rowlane = lds_addr(&L.x_ring[f_c][kFXP]) - 8u * (unsigned)f_pos;
            const unsigned tap_addr = lds_addr(&L.be80[0][kTapOff + FllRow::kTaps * (FllRow::kLanes - 1 - f_pos)]);
            const unsigned hist_addr = lds_addr(&L.x_ring[f_c][kFXP + kFX - FllRow::kReplay]);
            const unsigned long long p4 = (unsigned long long)__builtin_bit_cast(unsigned, 0.4f);
            if constexpr (LONG && CH == 16) {
                asm volatile(FLL8L_WAVE_ASM
                             : [ph] "+v"(ph), [fr] "+v"(fr), [base] "+s"(base_), [tiles] "+s"(tiles_), [st] "=&s"(st_)
                             : [a_addr] "v"(a_addr), [x_rowlane] "v"(x_rowlane), [tap_addr] "v"(tap_addr), [hist_addr] "v"(hist_addr),
                               [maxf] "v"(k1.fll_max_freq),
                               [negc1] "s"(FLL8L_WAVE_NEGC1), [beta] "s"(k1.fll_beta), [minf] "s"(k1.fll_min_freq),
                               [p4] "s"(p4),
                               [k1] "s"(FLL8L_WAVE_K1), [k2] "s"(FLL8L_WAVE_K2), [k3] "s"(FLL8L_WAVE_K3), [k4] "s"(FLL8L_WAVE_K4),
                               [a_sum] "v"(2u * a_addr + (unsigned)(sizeof(float2) * CH * kFAS))
                             : "vcc", "scc", "memory", FLL8L_WAVE_CLOBBERS);
            } else if constexpr (LONG) {
                asm volatile(FLL16L_WAVE_ASM
                             : [ph] "+v"(ph), [fr] "+v"(fr), [base] "+s"(base_), [tiles] "+s"(tiles_), [st] "=&s"(st_)
                             : [a_addr] "v"(a_addr), [x_rowlane] "v"(x_rowlane), [tap_addr] "v"(tap_addr), [hist_addr] "v"(hist_addr),
                               [maxf] "v"(k1.fll_max_freq),
                               [negc1] "s"(FLL16L_WAVE_NEGC1), [beta] "s"(k1.fll_beta), [minf] "s"(k1.fll_min_freq),
                               [p4] "s"(p4),
                               [k1] "s"(FLL16L_WAVE_K1), [k2] "s"(FLL16L_WAVE_K2), [k3] "s"(FLL16L_WAVE_K3), [k4] "s"(FLL16L_WAVE_K4),
                               [a_sum] "v"(2u * a_addr + (unsigned)(sizeof(float2) * CH * kFAS))
                             : "vcc", "scc", "memory", FLL16L_WAVE_CLOBBERS);
            } else if constexpr (CH == 4) {
                // (negc1: -C1 of the phasor's Cody-Waite reduction, or -(C1 + C2) for a block generated with the first two steps
                // folded into one fma -- exact for every phase in [-pi, pi], checked exhaustively in tests/test_oracle.py; the
                // generator says which in <block>_NEGC1)
                asm volatile(FLL16_WAVE_ASM
                             : [ph] "+v"(ph), [fr] "+v"(fr), [base] "+s"(base_), [tiles] "+s"(tiles_), [st] "=&s"(st_)
                             : [a_addr] "v"(a_addr), [x_rowlane] "v"(x_rowlane), [tap_addr] "v"(tap_addr), [hist_addr] "v"(hist_addr),
                               [maxf] "v"(k1.fll_max_freq),
                               [negc1] "s"(FLL16_WAVE_NEGC1), [beta] "s"(k1.fll_beta), [minf] "s"(k1.fll_min_freq),
                               [p4] "s"(p4),
                               [k1] "s"(FLL16_WAVE_K1), [k2] "s"(FLL16_WAVE_K2), [k3] "s"(FLL16_WAVE_K3), [k4] "s"(FLL16_WAVE_K4),
                               [a_sum] "v"(2u * a_addr + (unsigned)(sizeof(float2) * CH * kFAS))
                             : "vcc", "scc", "memory", FLL16_WAVE_CLOBBERS);
            } else if constexpr (CH == 16) {
                asm volatile(FLL_WAVE_ASM
                             : [ph] "+v"(ph), [fr] "+v"(fr), [base] "+s"(base_), [tiles] "+s"(tiles_), [st] "=&s"(st_)
                             : [a_addr] "v"(a_addr), [x_rowlane] "v"(x_rowlane), [tap_addr] "v"(tap_addr), [hist_addr] "v"(hist_addr),
                               [maxf] "v"(k1.fll_max_freq),
                               [negc1] "s"(FLL_WAVE_NEGC1), [beta] "s"(k1.fll_beta), [minf] "s"(k1.fll_min_freq),
                               [p4] "s"(p4),
                               [k1] "s"(FLL_WAVE_K1), [k2] "s"(FLL_WAVE_K2), [k3] "s"(FLL_WAVE_K3), [k4] "s"(FLL_WAVE_K4),
                               [a_sum] "v"(2u * a_addr + (unsigned)(sizeof(float2) * CH * kFAS))
                             : "vcc", "scc", "memory", FLL_WAVE_CLOBBERS);
            } else {
                asm volatile(FLL4_WAVE_ASM
                             : [ph] "+v"(ph), [fr] "+v"(fr), [base] "+s"(base_), [tiles] "+s"(tiles_), [st] "=&s"(st_)
                             : [a_addr] "v"(a_addr), [x_rowlane] "v"(x_rowlane), [tap_addr] "v"(tap_addr), [hist_addr] "v"(hist_addr),
                               [maxf] "v"(k1.fll_max_freq),
                               [negc1] "s"(FLL4_WAVE_NEGC1), [beta] "s"(k1.fll_beta), [minf] "s"(k1.fll_min_freq),
                               [p4] "s"(p4),
                               [k1] "s"(FLL4_WAVE_K1), [k2] "s"(FLL4_WAVE_K2), [k3] "s"(FLL4_WAVE_K3), [k4] "s"(FLL4_WAVE_K4),
                               [a_sum] "v"(2u * a_addr + (unsigned)(sizeof(float2) * CH * kFAS))
                             : "vcc", "scc", "memory", FLL4_WAVE_CLOBBERS);
            }
        }
        // the tiles the block did not take (the partial tile at the end of the call) and the trailing epochs
        if (nfull < ntiles) {
            FllRow R;
#pragma unroll
            for (int j = 0; j < FllRow::kTaps; j++) {
                const int kp = kTapOff + FllRow::kTaps * (FllRow::kLanes - 1 - f_pos) + j;
                R.ta[j] = p.be_re80[kp];
                R.tb[j] = p.be_im80[kp];
            }
            R.ph = ph;
            R.fr = fr;
            {
                FllDeviceIO io{ L, nullptr, f_c, f_pos, nfull * kFT };
                fll_replay<FllRow, FllDeviceIO>(R, k1, io);
            }
            for (int e = nfull + 1; e < ntiles + 4; e++) {
                FUSED_PROF_T0
                const int t = e - 1;
                if (t < ntiles) {
                    const int base = t * kFT;
                    const int cnt = (n - base < kFT) ? (n - base) : kFT;
                    FllDeviceIO io{ L, &L.a_buf[t & 1][f_c][0], f_c, f_pos, base };
                    fll_tile<FllRow, FllDeviceIO, ALPHA0>(R, k1, io, cnt);
                }
                FUSED_PROF_T1
                __syncthreads();
            }
            ph = R.ph;
            fr = R.fr;
        } else {
            for (int e = nfull + 1; e < ntiles + 4; e++) __syncthreads();
        }
        if (f_pos == 0 && live(f_c)) {
            p.fll_ph[ch0 + f_c] = ph;
            p.fll_fr[ch0 + f_c] = fr;
        }
    } else if (wave == kRoleC || wave == R_::C2) {
        // ---- RRC: lane -> (channel c = lane % CH, j = lane / CH), outputs base + 8 (j + groups * pass) + m; tile e-2 -------
        // 16 channels: four lane groups cover the tile's 32 samples in one pass; 32 channels: two groups, two passes.
        // The window of eight consecutive outputs starts at x_{i0-(nt-1)}; it is widened at the old end (under zero
        // taps) to start on a multiple of 8, so that no 8-sample chunk straddles the ring's wrap.
        // (4 channels: four groups cover the tile, lanes 16..63 idle)
        constexpr int kGroups = 64 / CH < kFT / 8 ? 64 / CH : kFT / 8, kPasses = kFT / (8 * kGroups);
        const bool c_on = lane < CH * kGroups;
        const int c = lane % CH;
        const int rrc_pad = (8 - ((p.ntaps - 1) & 7)) & 7;
#if defined(TETRA_EXP_ABLATE) && TETRA_EXP_ABLATE == 2      // experiment builds only (profiles/r02): RRC wave with a ninth of its work
        const int rrc_chunks = 1;
#else
        const int rrc_chunks = (p.ntaps - 1 + rrc_pad) / 8 + 1;
#endif
        const bool rrc_tri = rrc_pad == 0 && rrc_chunks >= 2;      // nt = 8k + 1 taps (the reference's 65): no alignment pad
        const unsigned x_base = pin_u32(lds_addr(&L.x_ring[c][kFXP]));
        // Reference-style reset / tap-count growth (tetra_demod.h, rrc_valid): delay-line samples older than the newest
        // valid0 are zeros to the RRC (and only to it).  Rare, and only the tiles whose windows reach into the delay
        // line are affected: they take the masked copy of the loop, wave-uniformly.
        const int valid0 = p.rrc_valid[chan(c)];
        const bool blanked = __builtin_amdgcn_readfirstlane(__any(valid0 < kH) ? 1 : 0) != 0;
        __syncthreads();
        FUSED_EPOCHS(
            const int t = e - 2;
            if (t >= 0 && t < ntiles) {
              // two RRC waves share the passes of a tile, one does them all
              const int pass0 = R_::C2 >= 0 && wave == R_::C2 ? kPasses / 2 : 0;
              const int pass1 = R_::C2 >= 0 && wave == kRoleC ? kPasses / 2 : kPasses;
              for (int pass = pass0; pass < pass1; pass++) {
                const int i0 = t * kFT + 8 * (lane / CH + kGroups * pass);
                if (i0 < n && c_on) {
                    const int start = i0 - (p.ntaps - 1) - rrc_pad;
                    Pair<float> out[kRrcOut];
                    auto tap4 = [&](int q) { const float4 t4 = reinterpret_cast<const float4*>(L.rrc)[q];
                                             Tap4 r; r.v[0] = t4.x; r.v[1] = t4.y; r.v[2] = t4.z; r.v[3] = t4.w; return r; };
                    auto ldx = [&](int q) {      // q = 8*ck + j: chunk base wrapped, j added as an immediate offset
                        lds_cfloat2* xw = (lds_cfloat2*)(size_t)(x_base + (((start + (q & ~7)) & (kFX - 1)) << 3));
                        const vfloat2 v = xw[q & 7];
                        return Pair<float>(v.x, v.y); };
                    if (blanked && t * kFT < p.ntaps - 1 + rrc_pad) {
                        rrc_direct8<false>(rrc_chunks,
                                    [&](int q) {
                                        const float2 v = x_ring_get(L, c, start + q);
                                        const bool seen = start + q >= -valid0;
                                        return Pair<float>(seen ? v.x : 0.0f, seen ? v.y : 0.0f); },
                                    tap4, out);
                    } else if (rrc_tri) {
                        rrc_direct8<true>(rrc_chunks, ldx, tap4, out);      // 65 taps: triangular first and last chunk
                    } else {
                        rrc_direct8<false>(rrc_chunks, ldx, tap4, out);
                    }
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
            }
        )
        if (wave == kRoleC && lane < CH && live(c)) p.rrc_valid[ch0 + c] = valid0 + n >= kH ? kH : valid0 + n;
    } else if (wave == kRoleD) {
        // ---- timing recovery; consumes y of tiles <= e-3 ------------------------------------------------------
        // 16- and 4-channel workgroups: FOUR lanes per channel (lane = 4 c + kq), each holding one of the symbol's three
        // interpolator rows (k2_timing_quad; identical loop state in the four lanes); 32-channel workgroup: one lane per channel.
        constexpr int kDL = CH <= 16 ? 4 : TETRA_EXP_DPAIR ? 2 : 1;
        const bool on = lane < CH * kDL;
        const int c = on ? lane / kDL : 0;
        const int kq = lane % kDL;
        K2State st;
        st.mu = p.mu[chan(c)];
        st.omega = p.omega[chan(c)];
        st.offset = p.offset[chan(c)];
        st.cph = 0; st.cfr = 0; st.ph2 = 0; st.prev = 0;
        int S = 0;
        // symbols a row can take: the bits row, and the symbol row when one is written (the caller's, or the quality
        // statistic's scratch, whose rows may be shorter than the caller's bits rows)
        const long long cap_ = p.sym && p.sym_stride < p.bits_stride / 2 ? p.sym_stride : p.bits_stride / 2;
        const int sym_cap = (int)cap_;
        bool cut = false;
        const unsigned y_base = pin_u32(lds_addr(&L.y_ring[c][0]));
        const unsigned bank_base = pin_u32(lds_addr(&L.bank[0]));
        const unsigned row_off = kDL == 2 ? (kq ? 0u : 64u) : kq == 1 ? 64u : kq == 2 ? 0u : 32u;      // kDL 2: the lane's SECOND row
        (void)row_off;
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
                    Pair<float> w[kInterpTaps];
                    _Pragma("unroll")
                    for (int j = 0; j < kInterpTaps; j++) {
                        const vfloat2 wv = yw[j];
                        w[j] = Pair<float>(wv.x, wv.y);
                    }
                    float vr; float vi;
                    vfloat4 q;
                    if constexpr (kDL == 4) {
                        // this lane's row of the padded table: kq 0 -> row phase (+32 bytes), 1 -> phase+1 (+64), 2 -> phase-1 (+0)
                        lds_cfloat4* bk = (lds_cfloat4*)(size_t)(bank_base + (phase << 5) + row_off);
                        float tr[kInterpTaps];
                        q = bk[0]; tr[0] = q.x; tr[1] = q.y; tr[2] = q.z; tr[3] = q.w;
                        q = bk[1]; tr[4] = q.x; tr[5] = q.y; tr[6] = q.z; tr[7] = q.w;
#if defined(TETRA_EXP_ABLATE) && TETRA_EXP_ABLATE == 4      // experiment builds only: the timing wave without its arithmetic
                        vr = w[0].x() + tr[0]; vi = w[0].y(); st.offset += 2;
#else
                        k2_timing_quad<kMinAdv>(k2, st, phase, w, tr, &vr, &vi);
#endif
                    } else if constexpr (kDL == 2) {
                        lds_cfloat4* bk = (lds_cfloat4*)(size_t)(bank_base + (phase << 5) + 32u);
                        lds_cfloat4* b2 = (lds_cfloat4*)(size_t)(bank_base + (phase << 5) + row_off);
                        float t0[kInterpTaps]; float t2[kInterpTaps];
                        q = bk[0]; t0[0] = q.x; t0[1] = q.y; t0[2] = q.z; t0[3] = q.w;
                        q = bk[1]; t0[4] = q.x; t0[5] = q.y; t0[6] = q.z; t0[7] = q.w;
                        q = b2[0]; t2[0] = q.x; t2[1] = q.y; t2[2] = q.z; t2[3] = q.w;
                        q = b2[1]; t2[4] = q.x; t2[5] = q.y; t2[6] = q.z; t2[7] = q.w;
                        k2_timing_pair<kMinAdv>(k2, st, phase, w, t0, t2, kq != 0, &vr, &vi);
                    } else {
                        lds_cfloat4* bk = (lds_cfloat4*)(size_t)(bank_base + (phase << 5));
                        float t0[kInterpTaps]; float tm1[kInterpTaps]; float tp1[kInterpTaps];
                        q = bk[0]; tm1[0] = q.x; tm1[1] = q.y; tm1[2] = q.z; tm1[3] = q.w;
                        q = bk[1]; tm1[4] = q.x; tm1[5] = q.y; tm1[6] = q.z; tm1[7] = q.w;
                        q = bk[2]; t0[0] = q.x; t0[1] = q.y; t0[2] = q.z; t0[3] = q.w;
                        q = bk[3]; t0[4] = q.x; t0[5] = q.y; t0[6] = q.z; t0[7] = q.w;
                        q = bk[4]; tp1[0] = q.x; tp1[1] = q.y; tp1[2] = q.z; tp1[3] = q.w;
                        q = bk[5]; tp1[4] = q.x; tp1[5] = q.y; tp1[6] = q.z; tp1[7] = q.w;
                        k2_timing<kMinAdv>(k2, st, phase, w, tm1, t0, tp1, &vr, &vi);
                    }
                    L.s_ring[c][S & (kSR - 1)] = make_float2(vr, vi);
                    S++;
                };
                // Output capacity guard.  (DEEP: always the per-symbol check -- several symbols may share an offset.)  Otherwise
                // every symbol advances the offset by >= 1 sample (k2_timing), so this
                // epoch adds at most limit - offset symbols.  If even that fits the output row the loop runs
                // unchecked (always the case for a finite stream except near the end of very short calls);
                // otherwise it checks per symbol, and a NaN/Inf-poisoned channel whose loop has stopped advancing
                // properly is cut off at the row capacity instead of overrunning it.
                if (!DEEP && S + (limit - st.offset) <= sym_cap) {
                    while (st.offset < limit) one_symbol();
                } else {
                    while (st.offset < limit) {
                        if (S >= sym_cap) { st.offset = limit; cut = true; break; }
                        one_symbol();
                    }
                }
                L.s_avail[c] = S;
            }
        )
        if (on && kq == 0 && live(c)) {
            p.mu[ch0 + c] = st.mu;
            p.omega[ch0 + c] = st.omega;
            p.offset[ch0 + c] = st.offset - n;          // complex_fd.cpp:145
            if (cut) {                                  // never silently: tetra_demod_get_overruns / TETRA_ERR_OVERRUN
                atomicAdd(p.overruns, 1);
                int* const flag = fused_cut_flag<CH, PROF>(p);
                if (flag) *(volatile int*)flag = 1;
            }
        }
    } else if (wave == kRoleE && (CH != kFChWide || TETRA_EXP_TWOPASS)) {
        // ---- kRoleE, 4- and 16-channel workgroups: the recurrence -- Costas loop only -- runs on lanes 0..CH-1 and leaves z in
        // place of v in the symbol ring; then ALL 64 lanes (64 / CH per channel, one symbol each) slice, decode differentially
        // against the symbol before, and store: the part of the reference's per-symbol work that is not a recurrence
        // (dqpsk_sym_extr.cpp:32-52, bit_unpacker.cpp:6-7) costs one pass per epoch instead of ~35 instruction slots per symbol.
        // (4 channels: this wave has a SIMD to itself and would set the pace, -8 %; 16 channels: -1.1 % once the FLL stream was
        // down to 57 slots, neutral before; 32 channels, where the wave shares its SIMD with the timing wave: +1.8 %, so that
        // shape keeps the one-pass form below.  profiles/r03/r03_e, r03_l, r03_p.)
        const bool on = lane < CH;
        const int c = on ? lane : 0;
        K2State st;
        st.mu = 0; st.omega = 0; st.offset = 0; st.prev = 0;
        st.cph = p.cph[chan(c)];
        st.cfr = p.cfr[chan(c)];
        st.ph2 = p.ph2[chan(c)];
        int S = 0;
        float2 zlast = make_float2(0.f, 0.f);              // z of the newest finished symbol (none yet: symbol 0 uses prev0)
        constexpr int kFin = 64 / CH;                      // finishing lanes per channel
        const int fc = lane / kFin, fj = lane % kFin;
        uint8_t* brow = p.bits + (long long)chan(fc) * p.bits_stride;
        float2* srow = p.sym ? p.sym + (long long)chan(fc) * p.sym_stride : nullptr;
        const bool fwr = live(fc);
        const int prev0 = p.prev[chan(fc)];                // the slicer's carried previous symbol: for this call's symbol 0
        K2Consts k2 = p.k2;
        k2.costas_max_freq = v_pin(k2.costas_max_freq);
        __syncthreads();
        FUSED_EPOCHS(
            if (e >= 4) {
                if (on) {
                    const int avail = L.s_avail[c];
                    L.e_span[0][c] = S;
                    L.e_span[1][c] = avail;
                    while (S < avail) {
                        const float2 v = L.s_ring[c][S & (kSR - 1)];
                        float zr; float zi;
#if defined(TETRA_EXP_ABLATE) && TETRA_EXP_ABLATE == 1      // experiment builds only: the Costas recurrence without its arithmetic
                        zr = v.x; zi = v.y;
#else
                        k2_costas_rot(k2, st, v.x, v.y, &zr, &zi);
#endif
                        zlast = make_float2(zr, zi);
                        L.s_ring[c][S & (kSR - 1)] = zlast;
                        S++;
                    }
                    // The symbol before this epoch's first one is NOT taken from the ring by the finishing lanes: at close to
                    // one sample per symbol the timing wave, up to a tile ahead, may already be writing into that slot.
                    L.e_last[e & 1][c] = zlast;
                }
                // same wave: its LDS operations execute in order, the fence only keeps the compiler from moving them
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int f0 = L.e_span[0][fc], f1 = L.e_span[1][fc];
                for (int i = f0 + fj; i < f1; i += kFin) {
                    const float2 z = L.s_ring[fc][i & (kSR - 1)];
                    // z of the symbol before: this epoch's own (in the ring), or the one the previous epoch ended on
                    const float2* zpp = i == f0 ? &L.e_last[(e - 1) & 1][fc] : &L.s_ring[fc][(i - 1) & (kSR - 1)];
                    const float2 zp = *zpp;
                    const int prevq = i == 0 ? prev0 : k2_quadrant(zp.x, zp.y);
                    const int d = k2_dibit(k2_quadrant(z.x, z.y), prevq);
                    if (fwr) {
                        // bit_unpacker.cpp:6-7: byte 2i = MSB, byte 2i+1 = LSB
                        *reinterpret_cast<unsigned short*>(brow + 2 * i) = (unsigned short)(((d >> 1) & 1) | ((d & 1) << 8));
                        if (srow) srow[i] = z;
                    }
                }
            }
        )
        if (on && live(c)) {
            p.cph[ch0 + c] = st.cph;
            p.cfr[ch0 + c] = st.cfr;
            p.ph2[ch0 + c] = st.ph2;
            if (S > 0) p.prev[ch0 + c] = k2_quadrant(zlast.x, zlast.y);
            p.n_bits[ch0 + c] = 2 * S;
        }
    } else if (wave == kRoleE) {
        // ---- kRoleE: Costas + slicer + differential decoder + bit unpacker; symbols published before e ----
        const bool on = lane < CH;
        const int c = on ? lane : 0;
        K2State st;
        st.mu = 0; st.omega = 0; st.offset = 0;
        st.cph = p.cph[chan(c)];
        st.cfr = p.cfr[chan(c)];
        st.ph2 = p.ph2[chan(c)];
        st.prev = p.prev[chan(c)];
        int S = 0;
        uint8_t* brow = p.bits + (long long)chan(c) * p.bits_stride;
        float2* srow = p.sym ? p.sym + (long long)chan(c) * p.sym_stride : nullptr;
        const bool wr = on && live(c);
        K2Consts k2 = p.k2;
        k2.costas_max_freq = v_pin(k2.costas_max_freq);
        __syncthreads();
        FUSED_EPOCHS(
            if (e >= 4 && on) {
                const int avail = L.s_avail[c];
                while (S < avail) {
                    const float2 v = L.s_ring[c][S & (kSR - 1)];
                    float zr; float zi;
#if defined(TETRA_EXP_ABLATE) && TETRA_EXP_ABLATE == 1      // experiment builds only (profiles/r02): E without its arithmetic
                    zr = v.x; zi = v.y;
                    const int d = S & 3;
#else
                    const int d = k2_costas(k2, st, v.x, v.y, &zr, &zi);
#endif
                    if (wr) {
                        // bit_unpacker.cpp:6-7: byte 2S = MSB, byte 2S+1 = LSB
                        *reinterpret_cast<unsigned short*>(brow + 2 * S) = (unsigned short)(((d >> 1) & 1) | ((d & 1) << 8));
                        if (srow) srow[S] = make_float2(zr, zi);
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
        }
    } else {
        // ---- a wave without a role (wide workgroup: it would share a SIMD with an FLL wave): barriers only ----
        __syncthreads();
        for (int e = 0; e < ntiles + 4; e++) __syncthreads();
    }
#ifdef TETRA_DEMOD_DEBUG
    if (PROF && lane == 0) {
        int slot = 0;          // report in role order E, D, F0, F1, A, C whatever the wave indices are
        for (int r = 0; r < 6; r++) slot = (CH == 16 && role_ids::v[r] == wave) ? r : slot;
        p.prof[(long long)blockIdx.x * 8 + slot] = busy_;
        if (wave == 0) p.prof[(long long)blockIdx.x * 8 + 7] = __builtin_readcyclecounter() - t_entry_;
    }
#else
    (void)t_entry_;
#endif
    // delay lines: last 80 FLL outputs, last 7 RRC outputs (both rings still hold them; the loops end on a barrier)
    for (int i = tid; i < CH * kHist; i += kThreadsCH) {
        const int c = i / kHist, m = i % kHist;
        if (live(c)) p.hist[(long long)(ch0 + c) * kHist + m] = x_ring_get(L, c, n - kHist + m);
    }
    if constexpr (LONG) {
        constexpr int kFar = kHistLong - kHist;
        for (int i = tid; i < CH * kFar; i += kThreadsCH) {
            const int c = i / kFar, m = i % kFar;
            if (live(c)) p.hist_far[(long long)(ch0 + c) * kFar + m] = x_ring_get(L, c, n - kHistLong + m);
        }
    }
    for (int i = tid; i < CH * (kInterpTaps - 1); i += kThreadsCH) {
        const int c = i / (kInterpTaps - 1), m = i % (kInterpTaps - 1);
        if (live(c))
            p.ybuf[(long long)(ch0 + c) * (kInterpTaps - 1) + m] = L.y_ring[c][(n - (kInterpTaps - 1) + m) & (kFY - 1)];
    }
}
#undef FUSED_EPOCHS
#undef FUSED_PROF_T0
#undef FUSED_PROF_T1

}  // namespace
