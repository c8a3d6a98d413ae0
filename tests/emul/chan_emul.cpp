// chan_emul.cpp -- host emulation of the channeliser's FFT kernel (TEST TOOL): the product's lane-level source
// (sdrpp-tetra-demodulator_amd/csrc/chan_fft_core.hpp: fold, 32-point FFT, 5 x 5 DFT, LDS layouts, thread -> slot / lane maps) run
// thread by thread, phase by phase, exactly as k_channelise_fft arranges them between its barriers -- so the index maps and the
// transforms are checked against the double-precision definition (oracle/chan_oracle.c) without a GPU.
// Build: g++ -O2 -std=c++17 -shared -fPIC
#include <cmath>
#include <vector>

#include "../../sdrpp-tetra-demodulator_amd/csrc/chan_fft_core.hpp"

using namespace chanfft;

extern "C" {

// hist: the L - 1 samples before the call (oldest first); x: the call's n_in new samples -- two separate buffers, like the kernel sees them; out: [frames][800] complex.  P in {4, 6, 8}.  Returns frames.
// fmt: the format of x (kFmtC32 / kFmtCs16 / kFmtCs8: interleaved float / int16 / int8 I, Q pairs); hist is always complex64.
int chan_fft_emul_fmt(const float* hist, const void* x, int fmt, int n_in, int P, int ph0, long long abs0, const float* h, float* out) {
    const int L = kM * P, D = kM / 2;
    const int frames = (ph0 + n_in) / D;
    std::vector<c32> tw((size_t)kN1 * kN2);
    const double pi = 3.14159265358979323846;
    for (int n1 = 0; n1 < kN1; n1++)
        for (int k2 = 0; k2 < kN2; k2++) {
            const double a = -2.0 * pi * (double)((n1 * k2) % kM) / kM;
            tw[(size_t)n1 * kN2 + k2] = mk((float)std::cos(a), (float)std::sin(a));
        }
    BlockCtx c;
    c.x = x;
    c.hist = reinterpret_cast<const c32*>(hist);
    c.n_in = n_in;
    c.out = reinterpret_cast<c32*>(out);
    c.h = h; c.tw = tw.data(); c.frames = frames; c.ph0 = ph0; c.abs0 = abs0; c.L = L;
    std::vector<c32> lds((size_t)kBlockFrames * kFrameLds);
    const int blocks = (frames + kBlockFrames - 1) / kBlockFrames;
    auto run = [&](auto Ptag) {
        constexpr int PP = decltype(Ptag)::value;
        std::vector<float> ht((size_t)2 * kM * PP);
        fold_transpose_prototype(h, PP, ht.data());
        c.h = ht.data();
        struct Regs { c32 x[32]; };
        std::vector<Regs> regs(256);
        std::vector<char> live(256);
        for (int blk = 0; blk < blocks; blk++) {
            for (int tid = 0; tid < 256; tid++) {
                if (fmt == kFmtCs16) phase_fold<PP, kFmtCs16>(c, blk, tid, lds.data());
                else if (fmt == kFmtCs8) phase_fold<PP, kFmtCs8>(c, blk, tid, lds.data());
                else phase_fold<PP, kFmtC32>(c, blk, tid, lds.data());
            }
            for (int tid = 0; tid < 256; tid++) live[tid] = phase_fft32_compute(tid, lds.data(), regs[tid].x);      // every lane reads ...
            for (int tid = 0; tid < 256; tid++) if (live[tid]) phase_fft32_store(tid, lds.data(), regs[tid].x);       // ... before any lane writes
            for (int tid = 0; tid < 256; tid++) { c32 tw[kN1 - 1]; load_twiddles(c, tid, tw); phase_dft25_store(c, (long long)kBlockFrames * blk, tid, lds.data(), tw); }
        }
    };
    if (P == 8) run(std::integral_constant<int, 8>());
    else if (P == 6) run(std::integral_constant<int, 6>());
    else if (P == 4) run(std::integral_constant<int, 4>());
    else return -1;
    return frames;
}

int chan_fft_emul(const float* hist, const float* x, int n_in, int P, int ph0, long long abs0, const float* h, float* out) {
    return chan_fft_emul_fmt(hist, x, kFmtC32, n_in, P, ph0, abs0, h, out);
}

// the two register-level transforms on their own (natural order in and out)
void chan_fft32(const float* in, float* out) {
    c32 x[32];
    for (int i = 0; i < 32; i++) x[i] = mk(in[2 * i], in[2 * i + 1]);
    fft32_dif(x);
    for (int p = 0; p < 32; p++) { out[2 * bitrev5(p)] = x[p].x; out[2 * bitrev5(p) + 1] = x[p].y; }
}
void chan_dft25(const float* in, float* out) {
    c32 x[25];
    for (int i = 0; i < 25; i++) x[i] = mk(in[2 * i], in[2 * i + 1]);
    dft25(x);
    for (int i = 0; i < 25; i++) { out[2 * i] = x[i].x; out[2 * i + 1] = x[i].y; }
}
}

