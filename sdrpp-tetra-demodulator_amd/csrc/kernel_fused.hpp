// kernel_fused.hpp -- the whole demodulator chain of 16 channels in ONE workgroup of six specialised
// wavefronts (included by tetra_demod.hip; device code only).
//
// Why this shape (measured on MI355X, profiles/r01_*): the chain is a set of per-channel serial
// recurrences and the kernels are VALU-issue bound -- a lone wave per SIMD already keeps its SIMD ~90 %
// busy -- so throughput is set by the busiest SIMD's instruction count per sample.  With 4096 channels
// there are exactly 4 channels per SIMD; the loop code (sincos, AGC, error functions) costs the same
// whether a wave carries 4 or 64 channels.  This kernel therefore splits a CU's 16 channels by STAGE
// instead of by channel so that each stage runs at the widest lane occupancy it allows and the four
// SIMDs carry equal instruction loads:
//
//   wave  role                                  lanes/channel   VALU ops / sample (approx.)
//   F0,F1 FLL: NCO, band-edge FIRs, loop         8 (interleaved) ~80   <- sets the pace, alone on a SIMD each
//   A     AGC                                    1               ~22  \  share one SIMD
//   E     Costas + slicer + diff. decoder + out  1               ~45  /
//   C     RRC matched filter (time-parallel)     4 x 8 outputs   ~18  \  share one SIMD
//   D     ML timing recovery                     1               ~40  /
//
// Stages are connected by LDS rings (AGC out -> FLL out x -> RRC out y -> symbols) and run as a
// software pipeline over 32-sample tiles with one workgroup barrier per tile: in epoch e, A works on
// tile e, F on e-1, C on e-2, D consumes y of tiles <= e-3 and E the symbols D published before the
// epoch.  No intermediate touches HBM: the kernel reads 8 B and writes 1 B per input sample.
#pragma once

namespace {

constexpr int kFT = 32;                  // samples per tile (pipeline epoch)
constexpr int kFCh = 16;                 // channels per workgroup
constexpr int kFThreads = 384;           // 6 waves
constexpr int kFX = 256;                 // x ring (FLL output) per channel ...
constexpr int kFXM = 88;                 // ... plus a mirror of the first slots so RRC windows (<= 86 samples) never wrap
constexpr int kFXS = kFX + kFXM + 1;     // row stride (odd: spreads channels over LDS banks)
constexpr int kFY = 128;                 // y ring (RRC output) per channel
constexpr int kFYM = 8;
constexpr int kFYS = kFY + kFYM + 1;
constexpr int kFS = 64;                  // symbol ring per channel

// wave index -> role.  A workgroup's waves are placed on SIMDs cyclically, so waves w and w+4 share a SIMD:
// the two FLL waves get SIMDs of their own, {E, A} and {D, C} pair up.  The issue arbiter serves the OLDEST wave
// of a SIMD first (profiles/r02/r02_a_issue_model.md), so the recurrence-bound roles (Costas, timing recovery) take the
// lower wave index of their pair and the throughput roles (AGC, RRC) fill the slots they leave.
enum { kRoleE = 0, kRoleD = 1, kRoleF0 = 2, kRoleF1 = 3, kRoleA = 4, kRoleC = 5 };

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
    const float* be_re72;   // band-edge taps zero-padded (old end) to 72
    const float* be_im72;
    const float* rrc_ext;   // [kRrcExt] RRC taps as rrc_direct8 wants them: ext[7 + k] = h[k], zero elsewhere
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
    int ablate;          // debug/profiling only: bit r set = role r keeps its barriers but skips its work (results invalid)
    long long* prof;     // debug/profiling only (TETRA_DEMOD_PROFILE): [workgroups][8] = busy clocks of waves 0..5 inside their epoch
                         // bodies (barrier waits excluded), [6] = clocks from kernel entry to exit of wave 0; null = off
};

struct FusedLds {
    float2 a_buf[2][kFCh][kFT];
    float2 x_ring[kFCh][kFXS];
    float2 y_ring[kFCh][kFYS];
    float2 s_ring[kFCh][kFS];
    int s_avail[kFCh];
    // interpolator bank with row 0 repeated in front and row 127 behind: rows max(p-1,0), p, min(p+1,127) of
    // complex_fd.cpp:102-121 are then the 24 contiguous floats at bank[p * 8]
    __attribute__((aligned(16))) float bank[(kInterpPhases + 2) * kInterpTaps];
    __attribute__((aligned(16))) float rrc[kRrcExt];       // zero-extended taps, see rrc_direct8
};

// Typed LDS pointers built from a 32-bit LDS byte address.  Keeping the (loop-invariant) row base in one pinned vector
// register makes the compiler address a sliding window as `base register + immediate offsets` (ds_read2_b64 /
// ds_read_b128 with offset fields) instead of re-deriving every element's address from the LDS struct offset.
typedef float vfloat2 __attribute__((ext_vector_type(2)));
typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const vfloat2 lds_cfloat2;
typedef __attribute__((address_space(3))) const vfloat4 lds_cfloat4;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }
__device__ __forceinline__ unsigned pin_u32(unsigned x) { asm volatile("" : "+v"(x)); return x; }

__device__ __forceinline__ void x_ring_put(FusedLds& L, int c, int i, float2 v) {
    const int s = i & (kFX - 1);
    L.x_ring[c][s] = v;
    if (s < kFXM) L.x_ring[c][s + kFX] = v;
}
__device__ __forceinline__ void y_ring_put(FusedLds& L, int c, int i, float2 v) {
    const int s = i & (kFY - 1);
    L.y_ring[c][s] = v;
    if (s < kFYM) L.y_ring[c][s + kFY] = v;
}

// LDS side of one FLL lane (see fll8_tile / fll8_replay in demod_core.hpp).
struct FllDeviceIO {
    FusedLds& L;
    const float2* hist_ch;   // this lane's channel's stored delay line (global)
    const float2* a_tile;    // a_buf[parity][ch] of the tile being processed
    int c;                   // channel within the workgroup
    int pos;                 // position along the channel's 8 lanes (0 = head)
    int tile_base;           // first sample index of the tile

    __device__ __forceinline__ Pair<float> load_hist(int g) const {
        const float2 v = hist_ch[(kHist - kF8Pad) + g * 8 + pos];
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

// Every role runs its own copy of the epoch loop (same trip count, one workgroup barrier per epoch): the
// branch on the wave index is wave-uniform, and keeping the roles in separate code paths keeps each role's
// registers out of the others' live ranges.
#define FUSED_EPOCHS(...)                                                        \
    for (int e = 0; e < ntiles + 4; e++) {                                       \
        const long long tb_ = PROF ? __builtin_readcyclecounter() : 0;           \
        __VA_ARGS__                                                              \
        if (PROF) busy_ += __builtin_readcyclecounter() - tb_;                   \
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
    const long long t_entry_ = PROF ? __builtin_readcyclecounter() : 0;
    auto chan = [&](int c) { const int ch = ch0 + c; return ch < p.n_channels ? ch : p.n_channels - 1; };
    auto live = [&](int c) { return ch0 + c < p.n_channels; };

    // ---- prologue: tables and delay lines into LDS -------------------------------------------------
    for (int i = tid; i < (kInterpPhases + 2) * kInterpTaps; i += kFThreads) {
        int row = i / kInterpTaps - 1;
        row = row < 0 ? 0 : (row > kInterpPhases - 1 ? kInterpPhases - 1 : row);
        L.bank[i] = p.bank[row * kInterpTaps + i % kInterpTaps];
    }
    if (tid < kRrcExt) L.rrc[tid] = p.rrc_ext[tid];
    // rings start at zero: the RRC window may touch slots that were never written (weighted by zero taps)
    for (int i = tid; i < kFCh * kFXS; i += kFThreads) (&L.x_ring[0][0])[i] = make_float2(0.f, 0.f);
    for (int i = tid; i < kFCh * kFYS; i += kFThreads) (&L.y_ring[0][0])[i] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < kFCh * kHist; i += kFThreads) {
        const int c = i / kHist, m = i % kHist;
        x_ring_put(L, c, m - kHist, p.hist[(long long)chan(c) * kHist + m]);
    }
    for (int i = tid; i < kFCh * (kInterpTaps - 1); i += kFThreads) {
        const int c = i / (kInterpTaps - 1), m = i % (kInterpTaps - 1);
        y_ring_put(L, c, m - (kInterpTaps - 1), p.ybuf[(long long)chan(c) * (kInterpTaps - 1) + m]);
    }
    if (tid < kFCh) L.s_avail[tid] = 0;

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
            if (e < ntiles && on && !(p.ablate & 1)) {
                const int base = e * kFT;
                float2* dst = &L.a_buf[e & 1][c][0];
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
        )
        if (on && live(c)) p.agc_g[ch0 + c] = g;
    } else if (wave == kRoleF0 || wave == kRoleF1) {
        // ---- FLL: lane -> (row r = lane>>4, pos = (lane&15)>>1, parity = lane&1); tile e-1 in epoch e ---
        const int fw = wave - kRoleF0;
        const int f_pos = (lane & 15) >> 1;
        const int f_c = fw * 8 + (lane >> 4) * 2 + (lane & 1);
        FllRow8<float> R;
#pragma unroll
        for (int j = 0; j < kF8Taps; j++) {
            const int kp = kF8Taps * (kF8Lanes - 1 - f_pos) + j;
            R.ta[j] = p.be_re72[kp];
            R.tb[j] = p.be_im72[kp];
        }
        R.ph = p.fll_ph[chan(f_c)];
        R.fr = p.fll_fr[chan(f_c)];
        K1Consts k1 = p.k1;
        k1.fll_max_freq = v_pin(k1.fll_max_freq);
        {
            FllDeviceIO io{ L, p.hist + (long long)chan(f_c) * kHist, nullptr, f_c, f_pos, 0 };
            fll8_replay<float, FllDeviceIO>(R, k1, io);
        }
        __syncthreads();
        FUSED_EPOCHS(
            const int t = e - 1;
            if (t >= 0 && t < ntiles && !(p.ablate & 2)) {
                const int base = t * kFT;
                const int cnt = (n - base < kFT) ? (n - base) : kFT;
                FllDeviceIO io{ L, nullptr, &L.a_buf[t & 1][f_c][0], f_c, f_pos, base };
                fll8_tile<float, FllDeviceIO, ALPHA0>(R, k1, io, cnt);
            }
        )
        if (f_pos == 0 && live(f_c)) {
            p.fll_ph[ch0 + f_c] = R.ph;
            p.fll_fr[ch0 + f_c] = R.fr;
        }
    } else if (wave == kRoleC) {
        // ---- RRC: lane -> (channel c = lane & 15, j = lane >> 4), outputs base + 8j + m; tile e-2 -------
        const int c = lane & 15;
        const int rrc_chunks = (p.ntaps + 7 + 7) / 8;
        __syncthreads();
        FUSED_EPOCHS(
            const int t = e - 2;
            if (t >= 0 && t < ntiles && !(p.ablate & 4)) {
                const int i0 = t * kFT + 8 * (lane >> 4);
                if (i0 < n) {
                    // window x_{i0-(nt-1)} .. x_{i0+7} (+ up to 7 zero-weighted slots of chunk padding, inside the mirror)
                    const float2* xw = &L.x_ring[c][(i0 - (p.ntaps - 1)) & (kFX - 1)];
                    Pair<float> out[kRrcOut];
                    rrc_direct8(rrc_chunks,
                                [&](int q) { const float2 v = xw[q]; return Pair<float>(v.x, v.y); },
                                [&](int q) { const float4 t = reinterpret_cast<const float4*>(L.rrc)[q];
                                             Tap4 r; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r; }, out);
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
            if (e >= 3 && on && !(p.ablate & 8)) {
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
            if (e >= 4 && on && !(p.ablate & 16)) {
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
    if (PROF && lane == 0) {
        p.prof[(long long)blockIdx.x * 8 + wave] = busy_;
        if (wave == 0) p.prof[(long long)blockIdx.x * 8 + 6] = __builtin_readcyclecounter() - t_entry_;
    }
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

}  // namespace
