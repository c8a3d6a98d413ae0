// resamp_emul.cpp -- host emulation of the rational resampler's kernels (TEST TOOL): the product's thread-level source
// (sdrpp-tetra-demodulator_amd/csrc/resamp_core.hpp: group / unit maps, the interior and the careful row accessors, the predicated
// stores, the phase table) run thread by thread over exactly the thread range tetra_resamp_process_device launches -- so the index
// arithmetic is checked against the double-precision definition (oracle/chan_oracle.c) without a GPU.
// Build: g++ -O2 -std=c++17 -shared -fPIC
#include <cstddef>
#include <vector>

#include "../../sdrpp-tetra-demodulator_amd/csrc/resamp_core.hpp"

using namespace resamp;

namespace {
template <int I, int DN, int T> bool run_fixed(const Ctx& c, int W, long long threads) {
    for (long long t = 0; t < threads; t++) {
        if (W == 4) thread_fixed<I, DN, T, 4>(c, t);
        else thread_fixed<I, DN, T, 2>(c, t);
    }
    return true;
}
}  // namespace

extern "C" {

// hist: [T - 1][C] complex before the call; x: [n_in][C]; out: [m1 - m0][C]; proto: [I T].  generic != 0: the run-time-ratio kernel.
// Returns the number of output frames, -1 if the (I, DN, T) has no specialised kernel and generic == 0.
int resamp_emul(int I, int DN, int T, int C, int W, int generic, const float* proto, const float* hist, const float* x, int n_in,
                long long n_total, long long m_next, float* out) {
    const long long m1 = outputs_after(n_total + n_in, I, DN);
    Ctx c;
    c.x = x; c.hist = hist; c.out = out;
    c.n0 = n_total; c.m0 = m_next; c.m1 = m1; c.n_in = n_in; c.units = 2 * C / W;
    c.I = I; c.DN = DN; c.T = T;
    if (m1 == m_next) return 0;
    constexpr int kThreads = 256;
    if (generic) {
        c.coef = proto;
        const long long blocks = ((m1 - m_next) * c.units + kThreads - 1) / kThreads;
        for (long long t = 0; t < blocks * kThreads; t++) {
            if (W == 4) thread_generic<4>(c, t);
            else thread_generic<2>(c, t);
        }
        return (int)(m1 - m_next);
    }
    std::vector<float> coef((size_t)I * T);
    phase_table(proto, I, DN, T, coef.data());
    c.coef = coef.data();
    const long long threads0 = ((m1 + I - 1) / I - m_next / I) * (long long)c.units;
    const long long threads = (threads0 + kThreads - 1) / kThreads * kThreads;
    bool ok = false;
    if (I == 18 && DN == 25 && T == 8) ok = run_fixed<18, 25, 8>(c, W, threads);
    else if (I == 18 && DN == 25 && T == 12) ok = run_fixed<18, 25, 12>(c, W, threads);
    else if (I == 18 && DN == 25 && T == 16) ok = run_fixed<18, 25, 16>(c, W, threads);
    else if (I == 18 && DN == 25 && T == 24) ok = run_fixed<18, 25, 24>(c, W, threads);
    else if (I == 2 && DN == 3 && T == 8) ok = run_fixed<2, 3, 8>(c, W, threads);
    else if (I == 3 && DN == 2 && T == 8) ok = run_fixed<3, 2, 8>(c, W, threads);
    else if (I == 1 && DN == 2 && T == 8) ok = run_fixed<1, 2, 8>(c, W, threads);
    return ok ? (int)(m1 - m_next) : -1;
}
}
