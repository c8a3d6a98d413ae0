// C entry points around the REFERENCE's own PI4DQPSK -> DQPSKSymbolExtractor -> BitUnpacker objects, for
// tests/test_reference_shim.py.  The reference sources are compiled where they lie (-I /root/reference/src) against the
// stand-in core headers in this directory; see README.md.  Chain order as the reference wires it (main.cpp:84-92).
#include <dsp/pi4dqpsk.h>
#include <dsp/dqpsk_sym_extr.h>
#include <dsp/bit_unpacker.h>

#include <memory>

namespace {
struct Chain {
    dsp::demod::PI4DQPSK demod;
    dsp::DQPSKSymbolExtractor extractor;
    dsp::BitUnpacker unpacker;
    std::vector<dsp::complex_t> sym;
    std::vector<uint8_t> dibits;
};
}

extern "C" {

void* ref_create(double symbolrate, double samplerate, int rrc_taps, double rrc_beta, double agc_rate, double costas_bw,
                 double fll_bw, double omega_gain, double mu_gain, double omega_rel_limit) {
    Chain* c = new Chain();
    c->demod.init(nullptr, symbolrate, samplerate, rrc_taps, rrc_beta, agc_rate, costas_bw, fll_bw, omega_gain, mu_gain,
                  omega_rel_limit);
    c->extractor.init(nullptr);
    c->unpacker.init(nullptr);
    c->sym.resize(STREAM_BUFFER_SIZE);
    c->dibits.resize(STREAM_BUFFER_SIZE);
    return c;
}

void ref_destroy(void* h) { delete (Chain*)h; }

// count <= STREAM_BUFFER_SIZE interleaved complex samples in; symbols (interleaved re,im), one bit per byte out.
// Returns the number of symbols produced (bits = 2 x symbols).
int ref_process(void* h, int count, const float* iq, float* symbols, uint8_t* bits) {
    Chain* c = (Chain*)h;
    if (count < 0 || count > STREAM_BUFFER_SIZE) { return -1; }
    const int ns = c->demod.process(count, (const dsp::complex_t*)iq, c->sym.data());
    memcpy(symbols, c->sym.data(), (size_t)ns * sizeof(dsp::complex_t));
    const int nd = c->extractor.process(ns, c->sym.data(), c->dibits.data());
    c->unpacker.process(nd, c->dibits.data(), bits);
    return ns;
}

// id follows include/tetra_demod.h's TETRA_PARAM_* numbering.
int ref_set_param(void* h, int id, double v) {
    Chain* c = (Chain*)h;
    switch (id) {
    case 0: c->demod.setSymbolrate(v); break;
    case 1: c->demod.setSamplerate(v); break;
    case 2: c->demod.setRRCTapCount((int)v); break;
    case 3: c->demod.setRRCBeta(v); break;  // int parameter in the reference: truncates
    case 4: c->demod.setAGCRate(v); break;
    case 5: c->demod.setCostasBandwidth(v); break;
    case 6: c->demod.setFllBandwidth(v); break;
    case 7: c->demod.setOmegaGain(v); break;
    case 8: c->demod.setMuGain(v); break;
    case 9: c->demod.setOmegaRelLimit(v); break;
    default: return -1;
    }
    return 0;
}

// PI4DQPSK::setRRCParams (pi4dqpsk.cpp:56-66): tap count and the (untruncated, double) roll-off in one call.
void ref_set_rrc_params(void* h, int taps, double beta) { ((Chain*)h)->demod.setRRCParams(taps, beta); }

void ref_reset(void* h) { ((Chain*)h)->demod.reset(); }

int ref_sync(void* h) { return ((Chain*)h)->extractor.sync ? 1 : 0; }

// The tables the reference's objects designed FOR THEMSELVES with the core headers they were compiled against (read-out; the members
// are protected): rrc[count], lower band-edge filter re / im [be_count], interpolator bank [128][8].  Returns the RRC tap count;
// *be_count receives the band-edge filters' length (the FLL keeps its construction-time filters across PI4DQPSK's setters).
int ref_get_tables(void* h, float* rrc, float* be_re, float* be_im, float* bank, int* be_count) {
    Chain* c = (Chain*)h;
    const int n = c->demod._rrcTapCount, nb = c->demod.fll._filt_size;
    for (int i = 0; i < n; i++) { rrc[i] = c->demod.rrcTaps.taps[i]; }
    for (int i = 0; i < nb; i++) { be_re[i] = c->demod.fll.lbandedgerrcTaps.taps[i].re; be_im[i] = c->demod.fll.lbandedgerrcTaps.taps[i].im; }
    for (int p = 0; p < 128; p++) { for (int k = 0; k < 8; k++) { bank[p * 8 + k] = c->demod.recov.interpBank.phases[p][k]; } }
    *be_count = nb;
    return n;
}

// The loop state of the reference's objects, for the bit-exact comparison with the oracle's reference-float mode.  The members
// are protected / private in the reference's headers; this file (and only this file) is compiled with g++ -fno-access-control
// so that they can be READ without touching or wrapping the reference's sources.
// out[0..8] = FastAGC gain, FLL pcl.phase, pcl.freq, COMPLEX_FD pcl.phase (mu), pcl.freq (omega), PLL pcl.phase, pcl.freq, ph2,
// DQPSKSymbolExtractor::standarderr; iout[0..1] = COMPLEX_FD offset, DQPSKSymbolExtractor prev.
void ref_get_state(void* h, float* out, int* iout) {
    Chain* c = (Chain*)h;
    out[0] = c->demod.agc._gain;
    out[1] = c->demod.fll.pcl.phase;
    out[2] = c->demod.fll.pcl.freq;
    out[3] = c->demod.recov.pcl.phase;
    out[4] = c->demod.recov.pcl.freq;
    out[5] = c->demod.costas.pcl.phase;
    out[6] = c->demod.costas.pcl.freq;
    out[7] = c->demod.costas.ph2;
    out[8] = c->extractor.standarderr;
    iout[0] = c->demod.recov.offset;
    iout[1] = c->extractor.prev;
}
}
