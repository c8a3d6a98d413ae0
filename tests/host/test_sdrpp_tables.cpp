// test_sdrpp_tables.cpp -- the dsp::block mirror built the way an SDR++ module build builds it: -DTETRA_WITH_SDRPP with SDR++'s core
// headers on the include path.  Here those headers are tests/refshim/ (our stand-ins; -DREFSHIM_FLOAT_PI selects the variant that
// spells pi as the float macro FL_M_PI), so what this program shows is the ROUTE: PI4DQPSK::init and every re-designing setter
// call the INCLUDED headers' generators (host/sdrpp_tables.h) and the kernels run exactly those tables -- whatever the headers
// compute.  In a real SDR++ tree the same code picks up upstream's arithmetic.  tests/test_sdrpp_tables.py drives it.
//
//   test_sdrpp_tables tables <out.f32> <count> <beta> <symbolrate> <samplerate>
//       no GPU: writes rrc[count], band-edge re[count], im[count], bank[1024] as the included headers design them
//   test_sdrpp_tables mirror <iq.f32> <chunk> <out_prefix>
//       GPU: the plugin's init (src/main.cpp:78-84), then the setters one after the other; after each step the handle's tables
//       (tetra_demod_get_tables) must equal the headers' output BIT FOR BIT (exit code 10 + step otherwise); every step processes
//       the stream in `chunk`-sample calls and writes <out_prefix>.<step>.sym / .bits / .tables for the oracle comparison
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../sdrpp-tetra-demodulator_amd/host/pi4dqpsk_gpu.h"

#ifndef TETRA_WITH_SDRPP
#error "build with -DTETRA_WITH_SDRPP -I tests/refshim"
#endif

namespace st = dsp::demod::sdrpp_tables;

static bool same_bits(const float* a, const float* b, size_t n) { return std::memcmp(a, b, n * sizeof(float)) == 0; }

static void write_floats(const std::string& path, const std::vector<float>& v) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) std::exit(2);
    std::fwrite(v.data(), sizeof(float), v.size(), f);
    std::fclose(f);
}

// the handle's tables against what the included headers design for (count, beta, rates) and -- band-edge filters and bank -- for
// the construction-time parameters: no PI4DQPSK setter re-designs those (pi4dqpsk.cpp:32-118)
static int check_tables(dsp::demod::PI4DQPSK& d, int count, double beta, double symbolrate, double samplerate, int count0, double beta0,
                        double symbolrate0, double samplerate0, std::vector<float>* all) {
    int nt = 0, nbe = 0;
    std::vector<float> rrc(TETRA_DEMOD_MAX_TAPS), re(TETRA_DEMOD_MAX_TAPS), im(TETRA_DEMOD_MAX_TAPS), bank(128 * 8);
    if (tetra_demod_get_tables(d.handle(), &nt, rrc.data(), &nbe, re.data(), im.data(), bank.data()) != TETRA_OK) return 1;
    const std::vector<float> wr = st::rrc(count, beta, symbolrate, samplerate);
    const std::vector<float> wb = st::bandedge(count0, (float)beta0, (int)symbolrate0, (int)samplerate0);
    const std::vector<float> wk = st::interpBank();
    if (nt != count || nbe != count0) return 2;
    if (!same_bits(rrc.data(), wr.data(), (size_t)count)) return 3;
    if (!same_bits(re.data(), wb.data(), (size_t)count0) || !same_bits(im.data(), wb.data() + count0, (size_t)count0)) return 4;
    if (!same_bits(bank.data(), wk.data(), bank.size())) return 5;
    if (all) {
        all->assign(rrc.begin(), rrc.begin() + nt);
        all->insert(all->end(), re.begin(), re.begin() + nbe);
        all->insert(all->end(), im.begin(), im.begin() + nbe);
        all->insert(all->end(), bank.begin(), bank.end());
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 7 && !std::strcmp(argv[1], "tables")) {
        const int count = std::atoi(argv[3]);
        const double beta = std::atof(argv[4]), symbolrate = std::atof(argv[5]), samplerate = std::atof(argv[6]);
        std::vector<float> all = st::rrc(count, beta, symbolrate, samplerate);
        const std::vector<float> be = st::bandedge(count, (float)beta, (int)symbolrate, (int)samplerate), bank = st::interpBank();
        all.insert(all.end(), be.begin(), be.end());
        all.insert(all.end(), bank.begin(), bank.end());
        write_floats(argv[2], all);
        std::printf("tables from the included headers: %s, %d taps\n", dsp::demod::PI4DQPSK::tablesFromSdrpp() ? "yes" : "NO", count);
        return dsp::demod::PI4DQPSK::tablesFromSdrpp() ? 0 : 1;
    }
    if (argc < 5 || std::strcmp(argv[1], "mirror")) return 2;
    FILE* f = std::fopen(argv[2], "rb");
    if (!f) return 2;
    std::vector<float> iq;
    float tmp[4096];
    size_t r;
    while ((r = std::fread(tmp, sizeof(float), 4096, f)) > 0) iq.insert(iq.end(), tmp, tmp + r);
    std::fclose(f);
    const int n = (int)(iq.size() / 2), chunk = std::atoi(argv[3]);
    const std::string prefix = argv[4];

    tetra_demod_config_t cfg;
    tetra_demod_default_config(&cfg);
    const int count0 = cfg.rrc_tap_count;
    const double beta0 = cfg.rrc_beta, sr0 = cfg.symbolrate, fs0 = cfg.samplerate;
    dsp::demod::PI4DQPSK d;
    d.init(nullptr, sr0, fs0, count0, beta0, cfg.agc_rate, cfg.costas_bandwidth, cfg.fll_bandwidth, cfg.omega_gain, cfg.mu_gain,
           cfg.omega_rel_limit);      // src/main.cpp:84
    if (d.lastStatus() != TETRA_OK) { std::fprintf(stderr, "init failed: %s\n", tetra_demod_strerror(d.lastStatus())); return 3; }

    int count = count0;
    double beta = beta0, sr = sr0, fs = fs0;
    std::vector<dsp::complex_t> out((size_t)n + 16);
    for (int step = 0; step < 7; step++) {
        switch (step) {
        case 0: break;                                              // as initialised
        case 1: d.setSymbolrate(17000); sr = 17000; break;          // pi4dqpsk.cpp:32-42
        case 2: d.setSamplerate(34000); fs = 34000; break;          // :44-54
        case 3: d.setRRCParams(49, 0.5); count = 49; beta = 0.5; break;   // :56-66
        case 4: d.setRRCTapCount(71); count = 71; break;            // :68-70 (grows: FIR::setTaps' history rule)
        case 5: d.setRRCBeta(1); beta = 1; break;                   // :72-74 (an int)
        case 6: d.setSymbolrate(18000); d.setSamplerate(36000); d.setRRCParams(65, 0.35); sr = 18000; fs = 36000; count = 65; beta = 0.35; break;
        }
        if (d.lastStatus() != TETRA_OK) { std::fprintf(stderr, "step %d: %s\n", step, tetra_demod_strerror(d.lastStatus())); return 4; }
        std::vector<float> tables;
        const int bad = check_tables(d, count, beta, sr, fs, count0, beta0, sr0, fs0, &tables);
        if (bad) { std::fprintf(stderr, "step %d: handle tables differ from the headers' (%d)\n", step, bad); return 10 + step; }
        std::vector<float> sym;
        std::vector<uint8_t> bits;
        for (int pos = 0; pos < n; pos += chunk) {
            const int c = n - pos < chunk ? n - pos : chunk;
            const int ns = d.process(c, reinterpret_cast<const dsp::complex_t*>(iq.data()) + pos, out.data());
            if (ns < 0) { std::fprintf(stderr, "step %d: process failed: %s\n", step, tetra_demod_strerror(d.lastStatus())); return 5; }
            sym.insert(sym.end(), reinterpret_cast<float*>(out.data()), reinterpret_cast<float*>(out.data()) + 2 * (size_t)ns);
            bits.insert(bits.end(), d.lastBits().begin(), d.lastBits().end());
        }
        const std::string base = prefix + "." + std::to_string(step);
        write_floats(base + ".sym", sym);
        write_floats(base + ".tables", tables);
        FILE* fb = std::fopen((base + ".bits").c_str(), "wb");
        std::fwrite(bits.data(), 1, bits.size(), fb);
        std::fclose(fb);
        std::printf("step %d: %d taps, tables == headers, %zu symbols\n", step, count, sym.size() / 2);
    }
    return 0;
}
