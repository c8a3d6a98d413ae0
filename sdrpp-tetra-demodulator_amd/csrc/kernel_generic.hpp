// kernel_generic.hpp -- the chain for the parameter sets the fused kernel's rings cannot hold (included by tetra_demod.hip; device
// code only): timing loops whose smallest step is below 0.07 samples per symbol (COMPLEX_FD::process, src/dsp/complex_fd.cpp:98-145,
// emits several symbols from one offset for as long as floor(mu) is 0) -- and, on request (TETRA_FLAG_GENERIC_KERNEL: tests, A/B
// runs), loops below 0.27 and RRC / band-edge filters of 73 .. 129 taps (the reference's PI4DQPSK::init and setRRCTapCount take any count,
// src/dsp/pi4dqpsk.cpp:11-30,56-70), which otherwise run in the fused kernel's LONG rows.  The plugin itself runs 65 taps at 2 samples per symbol (src/main.cpp:40,84) and never gets here: this kernel exists so
// that such parameters are IMPLEMENTED -- bit for bit the arithmetic contract, like the fused kernel -- instead of refused.  It is
// not fast and is not meant to be: ONE LANE PER CHANNEL walks the whole call stage by stage, direct-form FIRs over a delay line
// in an HBM scratch ([128 history | n new] FLL outputs and [7 | n] RRC outputs per channel), the same per-sample / per-symbol
// functions of demod_core.hpp the fused kernel's lanes run (agc_step, sincos_t, cmul_phasor, fll_error, pcl_advance, k2_timing,
// k2_costas), every FIR sum one fmaf chain in ascending tap order.
#pragma once

namespace {

constexpr int kGenHist = 128;            // delay-line samples kept per channel for this path: filters of up to 129 taps
constexpr int kGenMaxTaps = kGenHist + 1;

struct GenericParams {
    const float2* iq;
    long long in_ch_stride, in_t_stride;
    int n, n_channels;
    float *agc_g, *fll_ph, *fll_fr;
    float2* hist;          // [C][kHist]: the newest 80 FLL outputs (shared with the fused path) ...
    float2* hist_far;      // [C][kGenHist - kHist]: ... and the 48 before them (this path only)
    int far_valid;         // 0: hist_far is not current (the fused path ran since it was written): zeros to every filter
    int* rrc_valid;
    float *mu, *omega;
    int* offset;
    float *cph, *cfr, *ph2;
    int* prev;
    float2* ybuf;          // [C][7]
    const float *be_a, *be_b, *rrc;      // [ntaps_be], [ntaps_be], [ntaps]: un-padded tables
    int ntaps, ntaps_be;
    const float* bank;     // [128][8]
    float2* xs;            // scratch [C][kGenHist + n_max]
    float2* ys;            // scratch [C][7 + n_max]
    long long xs_stride, ys_stride;
    uint8_t* bits;
    long long bits_stride;
    int* n_bits;
    float2* sym;
    long long sym_stride;
    int* overruns;
    int* cut_flag;
    float2* y_dbg;         // optional: time-major [(7 + n)][C], row 7 + i = y_i (TETRA_FLAG_KEEP_RRC_OUT)
    K1Consts k1;
    K2Consts k2;
    int lanes;             // channels per wave (1 .. 64): every channel is one serial walk, so a launch of few channels spreads them
                           // over MANY waves (the latency of one walk is what it is; more waves per CU hide it)
};

__global__ __launch_bounds__(64) void k_generic(GenericParams p) {
    if ((int)threadIdx.x >= p.lanes) return;
    const int c = blockIdx.x * p.lanes + threadIdx.x;
    if (c >= p.n_channels) return;
    const int n = p.n, H = kGenHist;
    float2* xs = p.xs + (long long)c * p.xs_stride;
    float2* ys = p.ys + (long long)c * p.ys_stride;
    // delay line in front of the new samples: [far 48 | newest 80]
    for (int m = 0; m < H - kHist; m++) xs[m] = p.far_valid ? p.hist_far[(long long)c * (H - kHist) + m] : make_float2(0.f, 0.f);
    for (int m = 0; m < kHist; m++) xs[H - kHist + m] = p.hist[(long long)c * kHist + m];
    for (int m = 0; m < kInterpTaps - 1; m++) ys[m] = p.ybuf[(long long)c * (kInterpTaps - 1) + m];

    // ---- FastAGC::process + FLL::process, sample by sample (pi4dqpsk.cpp:134-135, fll.cpp:135-149) ----
    {
        float g = p.agc_g[c], ph = p.fll_ph[c], fr = p.fll_fr[c];
        const float2* in = p.iq + (long long)c * p.in_ch_stride;
        const int nb = p.ntaps_be;
        for (int i = 0; i < n; i++) {
            const float2 v = in[(long long)i * p.in_t_stride];
            const Pair<float> a = agc_step<float>(p.k1, Pair<float>(v.x, v.y), g);
            float s, co;
            sincos_t<float>(-ph, s, co);
            const Pair<float> x = cmul_phasor<float>(a, co, s);
            xs[H + i] = make_float2(x.x(), x.y());
            // the two band-edge FIRs over the newest nb samples as four fmaf chains, oldest sample first (conjugate tap pair)
            const float2* w = xs + H + i - (nb - 1);
            float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll 16          // sixteen window loads in flight per trip: the chains stay in tap order, the memory latency overlaps
            for (int k = 0; k < nb; k++) {
                const float2 xv = w[k];
                const float ta = p.be_a[k], tb = p.be_b[k];
                s1 = __builtin_fmaf(xv.x, ta, s1);
                s2 = __builtin_fmaf(xv.y, tb, s2);
                s3 = __builtin_fmaf(xv.x, tb, s3);
                s4 = __builtin_fmaf(xv.y, ta, s4);
            }
            const float err = fll_error<float>(Pair<float>(s1, s4), Pair<float>(s3, s2));
            pcl_advance<float, true, true>(err, ph, fr, p.k1.fll_alpha, p.k1.fll_beta, p.k1.fll_min_freq, p.k1.fll_max_freq);
        }
        p.agc_g[c] = g; p.fll_ph[c] = ph; p.fll_fr[c] = fr;
    }
    // ---- FIR<complex_t, float>::process: RRC (pi4dqpsk.cpp:136); delay-line samples older than the newest rrc_valid are zeros to it ----
    const int valid0 = p.rrc_valid[c];
    {
        const int nt = p.ntaps;
        for (int i = 0; i < n; i++) {
            const float2* w = xs + H + i - (nt - 1);
            const long long have = (long long)valid0 + i + 1;
            const int k0 = have >= nt ? 0 : (int)(nt - have);
            float ar = 0.f, ai = 0.f;
#pragma unroll 16
            for (int k = k0; k < nt; k++) {
                const float2 xv = w[k];
                const float t = p.rrc[k];
                ar = __builtin_fmaf(xv.x, t, ar);
                ai = __builtin_fmaf(xv.y, t, ai);
            }
            ys[kInterpTaps - 1 + i] = make_float2(ar, ai);
            if (p.y_dbg) p.y_dbg[(long long)(kInterpTaps - 1 + i) * p.n_channels + c] = make_float2(ar, ai);
        }
    }
    // ---- COMPLEX_FD::process -> PI4DQPSK_COSTAS::process -> slicer / differential decoder / bit unpacker, symbol by symbol ----
    {
        K2State st;
        st.mu = p.mu[c]; st.omega = p.omega[c]; st.offset = p.offset[c];
        st.cph = p.cph[c]; st.cfr = p.cfr[c]; st.ph2 = p.ph2[c]; st.prev = p.prev[c];
        uint8_t* brow = p.bits + (long long)c * p.bits_stride;
        float2* srow = p.sym ? p.sym + (long long)c * p.sym_stride : nullptr;
        const long long cap_ = p.sym && p.sym_stride < p.bits_stride / 2 ? p.sym_stride : p.bits_stride / 2;
        const int sym_cap = (int)cap_;
        int S = 0;
        bool cut = false;
        while (st.offset < n) {
            if (S >= sym_cap) { st.offset = n; cut = true; break; }
            const int phase = k2_phase(st.mu);
            const int pm = phase > 0 ? phase - 1 : 0, pp = phase < kInterpPhases - 1 ? phase + 1 : kInterpPhases - 1;
            Pair<float> w[kInterpTaps];
            float tm1[kInterpTaps], t0[kInterpTaps], tp1[kInterpTaps];
            for (int j = 0; j < kInterpTaps; j++) {
                const float2 yv = ys[st.offset + j];
                w[j] = Pair<float>(yv.x, yv.y);
                tm1[j] = p.bank[pm * kInterpTaps + j];
                t0[j] = p.bank[phase * kInterpTaps + j];
                tp1[j] = p.bank[pp * kInterpTaps + j];
            }
            float vr, vi, zr, zi;
            k2_timing<0>(p.k2, st, phase, w, tm1, t0, tp1, &vr, &vi);
            const int d = k2_costas(p.k2, st, vr, vi, &zr, &zi);
            brow[2 * S] = (uint8_t)((d >> 1) & 1);          // bit_unpacker.cpp:6-7
            brow[2 * S + 1] = (uint8_t)(d & 1);
            if (srow) srow[S] = make_float2(zr, zi);
            S++;
            // an offset that wrapped negative (mu = +Inf saturates floor(mu)) would index the scratch out of range: cut the channel
            if (st.offset < 0) { st.offset = n; cut = true; break; }
        }
        p.mu[c] = st.mu; p.omega[c] = st.omega; p.offset[c] = st.offset - n;          // complex_fd.cpp:145
        p.cph[c] = st.cph; p.cfr[c] = st.cfr; p.ph2[c] = st.ph2; p.prev[c] = st.prev;
        p.n_bits[c] = 2 * S;
        if (cut) {
            atomicAdd(p.overruns, 1);
            if (p.cut_flag) *(volatile int*)p.cut_flag = 1;
        }
    }
    // carried delay lines
    for (int m = 0; m < H - kHist; m++) p.hist_far[(long long)c * (H - kHist) + m] = xs[n + m];
    for (int m = 0; m < kHist; m++) p.hist[(long long)c * kHist + m] = xs[n + H - kHist + m];
    for (int m = 0; m < kInterpTaps - 1; m++) p.ybuf[(long long)c * (kInterpTaps - 1) + m] = ys[n + m];
    p.rrc_valid[c] = (long long)valid0 + n >= H ? H : valid0 + n;
}

}  // namespace
