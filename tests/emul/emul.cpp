// emul.cpp -- host emulation of the GPU kernel's lane-level algorithms (TEST TOOL).
//
// Compiles the product's shared kernel source (sdrpp-tetra-demodulator_amd/csrc/demod_core.hpp) with -DTETRA_HOST_EMUL:
// the building blocks of k_fused (agc_step, the FLL row FllRowT with fll_replay / fll_tile on a 16-lane Row16 with
// emulated DPP moves, rrc_direct8, k2_timing, k2_costas) run stage after stage over linear arrays.  The tests compare this
// against the CPU oracle so that the systolic schedule, the replay of the delay line and the tile bookkeeping are verified
// without a GPU.  It also exposes the product's host-side filter design (design.hpp) for comparison with the oracle's.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -mfma -mavx2 -DTETRA_HOST_EMUL -shared -fPIC
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/tetra_demod.h"
#include "../../sdrpp-tetra-demodulator_amd/csrc/demod_core.hpp"
#include "../../sdrpp-tetra-demodulator_amd/csrc/design.hpp"
#include "../../sdrpp-tetra-demodulator_amd/csrc/constellation_core.hpp"

using namespace tdm;

namespace {

constexpr int kYHist = kInterpTaps - 1;

}  // namespace

extern "C" {

struct emul_tables {
    int ntaps;
    float rrc[kBePadLong], be_re[kBePadLong], be_im[kBePadLong];  // unpadded, [ntaps] used (up to 129: the long rows)
    float bank[kInterpPhases * kInterpTaps];
    K1Consts k1;
    K2Consts k2;
    float tr_omega;
};

// Product host design (design.hpp) from a tetra_demod_config_t.
int emul_design(const tetra_demod_config_t* cfg, emul_tables* t) {
    host::DesignParams dp;
    dp.symbolrate = cfg->symbolrate; dp.samplerate = cfg->samplerate; dp.rrc_tap_count = cfg->rrc_tap_count;
    dp.rrc_beta = cfg->rrc_beta; dp.agc_rate = cfg->agc_rate; dp.costas_bandwidth = cfg->costas_bandwidth;
    dp.fll_bandwidth = cfg->fll_bandwidth; dp.omega_gain = cfg->omega_gain; dp.mu_gain = cfg->mu_gain;
    dp.omega_rel_limit = cfg->omega_rel_limit;
    host::Design d;
    if (!host::make_design(dp, cfg->rrc_taps, cfg->bandedge_taps, cfg->interp_bank, d)) return -1;
    std::memset(t, 0, sizeof(*t));
    t->ntaps = d.ntaps;
    std::memcpy(t->rrc, d.rrc.data(), sizeof(float) * d.ntaps);
    std::memcpy(t->be_re, d.be_re.data(), sizeof(float) * d.ntaps);
    std::memcpy(t->be_im, d.be_im.data(), sizeof(float) * d.ntaps);
    std::memcpy(t->bank, d.bank.data(), sizeof(t->bank));
    t->k1 = d.k1; t->k2 = d.k2; t->tr_omega = d.tr_omega;
    return 0;
}

// The product's output-row rule (design.hpp bits_stride_for = tetra_demod_bits_stride_for): -1 when the design is refused.
long long emul_bits_stride_for(const tetra_demod_config_t* cfg, long long n) {
    host::DesignParams dp;
    dp.symbolrate = cfg->symbolrate; dp.samplerate = cfg->samplerate; dp.rrc_tap_count = cfg->rrc_tap_count;
    dp.rrc_beta = cfg->rrc_beta; dp.agc_rate = cfg->agc_rate; dp.costas_bandwidth = cfg->costas_bandwidth;
    dp.fll_bandwidth = cfg->fll_bandwidth; dp.omega_gain = cfg->omega_gain; dp.mu_gain = cfg->mu_gain;
    dp.omega_rel_limit = cfg->omega_rel_limit;
    host::Design d;
    if (!host::make_design(dp, nullptr, nullptr, nullptr, d)) return -1;
    return host::bits_stride_for(d, n);
}

void emul_default_cfg(tetra_demod_config_t* cfg) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->n_channels = 1; cfg->max_samples = 65536; cfg->device = -1;
    cfg->symbolrate = 18000; cfg->samplerate = 36000; cfg->rrc_tap_count = 65; cfg->rrc_beta = 0.35f;
    cfg->agc_rate = 0.02f; cfg->costas_bandwidth = 0.01f; cfg->fll_bandwidth = 0.006f;
    host::default_timing_gains(cfg->omega_gain, cfg->mu_gain);
    cfg->omega_rel_limit = 0.02f;
}

void emul_reset_state(const emul_tables* t, tetra_demod_channel_state_t* st) {
    std::memset(st, 0, sizeof(*st));
    st->agc_gain = 1.0f;
    st->omega = t->tr_omega;
    st->rrc_valid = t->ntaps > kF8Pad ? kHistLong : kHist;      // "all of it": the long rows keep 128 delay-line samples
}

// Fused pipeline: the building blocks of csrc/kernel_fused.hpp (agc_step, FllRowT + fll_replay/fll_tile,
// rrc_direct8, k2_timing, k2_costas) run stage after stage over linear arrays -- a valid serialisation of the
// device's barrier-synchronised software pipeline.  (The LDS ring/epoch bookkeeping itself is device-only.)
// ---------------------------------------------------------------------------------------------------
}  // extern "C"
namespace {
// One DPP row of an FLL wave: Row::kHop channels interleaved on the 16 lanes, lane = kHop * pos + channel-in-row.
template <class Row> struct FllEmulIO {
    static constexpr int H = Row::kHop;
    const float* hist[H];   // per channel of the row: stored delay line [hist_len][2]
    int hist_len = kHist;   // 80; the long rows: 144 = 16 zeros (under zero taps) + the 128 samples they keep
    const float* a[H];      // AGC output of the whole chunk [n][2]
    float* x[H];            // FLL output [n][2]
    int tile_base = 0;
    int n = 0;

    Pair<Row16> load_hist(int g) const {
        Row16 re, im;
        for (int l = 0; l < 16; l++) {
            const int pos = l / H, par = l % H;
            const int m = (hist_len - Row::kReplay) + g * Row::kLanes + pos;
            re.l[l] = hist[par][2 * m];
            im.l[l] = hist[par][2 * m + 1];
        }
        return Pair<Row16>(re, im);
    }
    Pair<Row16> sample(int s) const {
        Row16 re, im;
        const int i = tile_base + s;
        for (int l = 0; l < 16; l++) {
            const int par = l % H;
            re.l[l] = i < n ? a[par][2 * i] : 0.f;
            im.l[l] = i < n ? a[par][2 * i + 1] : 0.f;
        }
        return Pair<Row16>(re, im);
    }
    void xs_store(int iend, int cnt, Pair<Row16> xs) {
        for (int l = 0; l < 16; l++) {
            const int pos = l / H, par = l % H;
            if (pos < cnt) {
                const int i = tile_base + iend - 1 - pos;
                x[par][2 * i] = xs.x().l[l];
                x[par][2 * i + 1] = xs.y().l[l];
            }
        }
    }
};

// F stage of emul_fused for one row geometry: FLL rows of Row::kHop interleaved channels.
// (LONG: the rows of filters beyond 72 taps -- tables padded to 144, delay line = [16 zeros | hist_far 48 | hist 80])
template <class Row, bool LONG = false> void emul_fll(const emul_tables* t, tetra_demod_channel_state_t* st, int C, int n, const float* re72,
                                   const float* im72, std::vector<float>& a, std::vector<float>& x) {
    constexpr int H = Row::kHop, tile = 32;
    constexpr int kLine = LONG ? kF16LPad : kHist;
    static_assert(Row::kReplay <= kLine, "the delay line holds the replayed samples");
    constexpr int tap_off = (LONG ? kBePadLong : kBePad) - Row::kLanes * Row::kTaps;       // padded tap kp of the row = entry kp + tap_off of the padded table
    std::vector<float> zeros((size_t)std::max(n, 1) * 2, 0.f), zhist(2 * kLine, 0.f), dump((size_t)std::max(n, 1) * 2);
    std::vector<std::vector<float>> lines;
    if (LONG)
        for (int c = 0; c < C; c++) {
            std::vector<float> ln(2 * kLine, 0.f);
            std::memcpy(ln.data() + 2 * (kLine - kHistLong), st[c].hist_far, sizeof(float) * 2 * (kHistLong - kHist));
            std::memcpy(ln.data() + 2 * (kLine - kHist), st[c].hist, sizeof(float) * 2 * kHist);
            lines.push_back(ln);
        }
    for (int c0 = 0; c0 < C; c0 += H) {
        Row R;
        for (int j = 0; j < Row::kTaps; j++) {
            Row16 ta, tb;
            for (int l = 0; l < 16; l++) {
                const int pos = l / H;
                const int kp = tap_off + Row::kTaps * (Row::kLanes - 1 - pos) + j;
                ta.l[l] = re72[kp];
                tb.l[l] = im72[kp];
            }
            R.ta[j] = ta;
            R.tb[j] = tb;
        }
        FllEmulIO<Row> io;
        io.hist_len = kLine;
        for (int par = 0; par < H; par++) {
            const bool have = c0 + par < C;
            const int c = have ? c0 + par : c0;
            io.hist[par] = !have ? zhist.data() : LONG ? lines[c].data() : st[c].hist;
            io.a[par] = have ? &a[(size_t)c * n * 2] : zeros.data();
            io.x[par] = have ? &x[(size_t)c * n * 2] : dump.data();
        }
        for (int l = 0; l < 16; l++) {
            const int c = c0 + l % H < C ? c0 + l % H : c0;
            R.ph.l[l] = st[c].fll_phase;
            R.fr.l[l] = st[c].fll_freq;
        }
        io.n = n;
        fll_replay<Row, FllEmulIO<Row>>(R, t->k1, io);
        for (int base = 0; base < n; base += tile) {
            io.tile_base = base;
            const int cnt = n - base < tile ? n - base : tile;
            if (t->k1.fll_alpha == 0.0f) fll_tile<Row, FllEmulIO<Row>, true>(R, t->k1, io, cnt);
            else fll_tile<Row, FllEmulIO<Row>, false>(R, t->k1, io, cnt);
        }
        for (int par = 0; par < H && c0 + par < C; par++) {
            st[c0 + par].fll_phase = R.ph.l[par];
            st[c0 + par].fll_freq = R.fr.l[par];
        }
    }
}
}  // namespace
extern "C" {

// C <= 64 channels (processed in rows of two).  iq [C][n] channel-major.  Outputs like emul_k2, plus y_out [C][n].
int emul_fused_shape(const emul_tables* t, tetra_demod_channel_state_t* st, int C, int n, const float* iq, float* y_out,
                     uint8_t* bits, int bits_stride, int32_t* n_bits, float* sym, int fll_lanes) {
    if (C < 1 || C > 64 || t->ntaps > kRrcMaxTapsLong) return -1;
    const bool long_rows = t->ntaps > kF8Pad;          // filters beyond 72 taps: the LONG variant of the 4- / 16-channel shape
    if (long_rows && fll_lanes != 16 && fll_lanes != 8) return -1;
    const int kH = long_rows ? kHistLong : kHist;      // delay-line samples carried
    float re72[kBePadLong] = { 0 }, im72[kBePadLong] = { 0 };
    float rrc_ext[kRrcExtLong] = { 0 };
    const int o72 = (long_rows ? kBePadLong : kBePad) - t->ntaps;
    const int rpad = (8 - ((t->ntaps - 1) & 7)) & 7;      // RRC windows start on a multiple of 8 (see kernel_fused.hpp)
    for (int k = 0; k < t->ntaps; k++) { re72[o72 + k] = t->be_re[k]; im72[o72 + k] = t->be_im[k]; rrc_ext[7 + rpad + k] = t->rrc[k]; }
    const int rrc_chunks = (t->ntaps - 1 + rpad) / 8 + 1;
    std::vector<float> a((size_t)C * n * 2), x((size_t)C * n * 2);
    // A: AGC
    for (int c = 0; c < C; c++) {
        float g = st[c].agc_gain;
        for (int i = 0; i < n; i++) {
            Pair<float> o = agc_step<float>(t->k1, Pair<float>(iq[((size_t)c * n + i) * 2], iq[((size_t)c * n + i) * 2 + 1]), g);
            a[((size_t)c * n + i) * 2] = o.x();
            a[((size_t)c * n + i) * 2 + 1] = o.y();
        }
        st[c].agc_gain = g;
    }
    // F: FLL rows (8 lanes per channel: the 16-channel workgroup; 4 lanes: the 32-channel one; taps beyond 68 need the former)
    if (long_rows && fll_lanes == 16) emul_fll<FllRow16L<Row16>, true>(t, st, C, n, re72, im72, a, x);
    else if (long_rows) emul_fll<FllRow8L<Row16>, true>(t, st, C, n, re72, im72, a, x);
    else if (fll_lanes == 4 && t->ntaps <= kF4Pad) emul_fll<FllRow4<Row16>>(t, st, C, n, re72, im72, a, x);
    else if (fll_lanes == 8) emul_fll<FllRow8<Row16>>(t, st, C, n, re72, im72, a, x);
    else if (fll_lanes == 16) emul_fll<FllRow16<Row16>>(t, st, C, n, re72, im72, a, x);
    else return -1;
    // C: RRC, eight outputs at a time, over [history | x]
    std::vector<float> y((size_t)C * n * 2);
    for (int c = 0; c < C; c++) {
        std::vector<float> xf0((size_t)(8 + kH + n + 16) * 2, 0.f);      // 8 zero samples in front: the aligned window may reach x_{-87} (x_{-135})
        float* const xfp = xf0.data() + 16;
        if (long_rows) std::memcpy(xfp, st[c].hist_far, sizeof(float) * 2 * (kHistLong - kHist));
        std::memcpy(xfp + 2 * (kH - kHist), st[c].hist, sizeof(float) * 2 * kHist);
        if (n) std::memcpy(xfp + 2 * kH, &x[(size_t)c * n * 2], sizeof(float) * 2 * n);
        struct { float* p; float* data() { return p; } } xf{ xfp };
        for (int i0 = 0; i0 < n; i0 += 8) {
            Pair<float> out[kRrcOut];
            const int start = i0 - (t->ntaps - 1) - rpad;
            const float* w = xf.data() + 2 * (kH + start);
            const int valid0 = st[c].rrc_valid;           // delay-line samples the RRC may see (tetra_demod.h)
            const bool tri = rpad == 0 && rrc_chunks >= 2 && st[c].rrc_valid >= kH;
            auto run = [&](auto T) { rrc_direct8<decltype(T)::value>(rrc_chunks, [&](int q) { const bool seen = start + q >= -valid0;
                                                 return Pair<float>(seen ? w[2 * q] : 0.0f, seen ? w[2 * q + 1] : 0.0f); },
                        [&](int q) { Tap4 r; for (int z = 0; z < 4; z++) r.v[z] = rrc_ext[4 * q + z]; return r; }, out); };
            // like the kernel: 8k+1 taps without hidden delay-line samples take the triangular end chunks
            if (tri) run(std::true_type{}); else run(std::false_type{});
            for (int m = 0; m < kRrcOut && i0 + m < n; m++) {
                y[((size_t)c * n + i0 + m) * 2] = out[m].x();
                y[((size_t)c * n + i0 + m) * 2 + 1] = out[m].y();
            }
        }
        // new delay line
        // (the regular rows carry the newest 80 samples: what lies before them reads as zeros afterwards, tetra_demod.h hist_far)
        std::vector<float> nh(2 * kH);
        std::memcpy(nh.data(), xf.data() + 2 * n, sizeof(float) * 2 * kH);
        std::memset(st[c].hist_far, 0, sizeof(st[c].hist_far));
        if (long_rows) std::memcpy(st[c].hist_far, nh.data(), sizeof(float) * 2 * (kHistLong - kHist));
        std::memcpy(st[c].hist, nh.data() + 2 * (kH - kHist), sizeof(float) * 2 * kHist);
        st[c].rrc_valid = st[c].rrc_valid + n >= kH ? kH : st[c].rrc_valid + n;
    }
    if (y_out && n) std::memcpy(y_out, y.data(), sizeof(float) * y.size());
    // D + E
    for (int c = 0; c < C; c++) {
        std::vector<float> yf((size_t)(kYHist + n) * 2);
        std::memcpy(yf.data(), st[c].ybuf, sizeof(float) * 2 * kYHist);
        if (n) std::memcpy(yf.data() + 2 * kYHist, &y[(size_t)c * n * 2], sizeof(float) * 2 * n);
        K2State ks;
        ks.mu = st[c].mu; ks.omega = st[c].omega; ks.offset = st[c].offset;
        ks.cph = st[c].costas_phase; ks.cfr = st[c].costas_freq; ks.ph2 = st[c].ph2; ks.prev = st[c].prev;
        int S = 0;
        while (ks.offset < n) {
            const int phase = k2_phase(ks.mu);
            const int pm = phase > 0 ? phase - 1 : 0;
            const int pp = phase < kInterpPhases - 1 ? phase + 1 : kInterpPhases - 1;
            Pair<float> w[kInterpTaps];
            for (int j = 0; j < kInterpTaps; j++) w[j] = Pair<float>(yf[2 * (ks.offset + j)], yf[2 * (ks.offset + j) + 1]);
            float vr, vi, zr, zi;
            k2_timing(t->k2, ks, phase, w, t->bank + pm * kInterpTaps, t->bank + phase * kInterpTaps,
                      t->bank + pp * kInterpTaps, &vr, &vi);
            const int d = k2_costas(t->k2, ks, vr, vi, &zr, &zi);
            if (2 * S + 2 > bits_stride) return -2;
            if (sym) { sym[((size_t)c * (bits_stride / 2) + S) * 2] = zr; sym[((size_t)c * (bits_stride / 2) + S) * 2 + 1] = zi; }
            bits[(size_t)c * bits_stride + 2 * S] = (uint8_t)((d >> 1) & 1);
            bits[(size_t)c * bits_stride + 2 * S + 1] = (uint8_t)(d & 1);
            S++;
        }
        n_bits[c] = 2 * S;
        st[c].mu = ks.mu; st[c].omega = ks.omega; st[c].offset = ks.offset - n;
        st[c].costas_phase = ks.cph; st[c].costas_freq = ks.cfr; st[c].ph2 = ks.ph2; st[c].prev = ks.prev;
        std::memcpy(st[c].ybuf, yf.data() + 2 * n, sizeof(float) * 2 * kYHist);
    }
    return 0;
}

int emul_fused(const emul_tables* t, tetra_demod_channel_state_t* st, int C, int n, const float* iq, float* y_out,
               uint8_t* bits, int bits_stride, int32_t* n_bits, float* sym) {
    return emul_fused_shape(t, st, C, n, iq, y_out, bits, bits_stride, n_bits, sym, 8);
}

// DQPSKSymbolExtractor's per-symbol angular distance as k_quality computes it (demod_core.hpp quality_distance): n symbols
// z[2i], z[2i+1] -> out[i]
void emul_quality_distance(int n, const float* z, float* out) {
    for (int i = 0; i < n; i++) out[i] = quality_distance(z[2 * i], z[2 * i + 1]);
}

// k_constellation for one channel and one call (constellation_core.hpp): phase 1 for every thread index, then phase 2 for every
// thread index (the kernel's barrier), then the two counters.  z[n] this call's symbols (re, im pairs), blk / part [1024] pairs.
void emul_constellation(int n, const float* z, float* blk, float* part, int* fill, int* blocks, int nthr) {
    struct Z { float re, im; };
    const Z* zz = reinterpret_cast<const Z*>(z);
    Z* B = reinterpret_cast<Z*>(blk);
    Z* P = reinterpret_cast<Z*>(part);
    const int f0 = *fill;
    const tetra_cd::Plan p = tetra_cd::plan(f0, n);
    for (int tid = 0; tid < nthr; tid++) tetra_cd::assemble_block(p, tid, nthr, zz, P, B);
    for (int tid = nthr - 1; tid >= 0; tid--) tetra_cd::carry_partial(p, f0, n, tid, nthr, zz, P);
    *fill = p.r;
    *blocks += p.nb;
}

}  // extern "C"
