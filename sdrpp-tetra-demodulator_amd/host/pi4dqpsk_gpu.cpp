// pi4dqpsk_gpu.cpp -- see pi4dqpsk_gpu.h.  Thin glue over the C ABI; mirrors src/dsp/pi4dqpsk.cpp's control flow
// (ctrlMtx + tempStop()/tempStart() around anything that re-designs filters, src/dsp/pi4dqpsk.cpp:32-42).
#include "pi4dqpsk_gpu.h"

#include <cassert>

namespace dsp {
namespace demod {

PI4DQPSK::~PI4DQPSK() {
    if (base_type::_block_init) base_type::stop();
    if (h_) tetra_demod_destroy(h_);
}

void PI4DQPSK::init(stream<complex_t>* in, double symbolrate, double samplerate, int rrcTapCount, double rrcBeta,
                    double agcRate, double costasBandwidth, double fllBandwidth, double omegaGain, double muGain,
                    double omegaRelLimit) {
    tetra_demod_config_t cfg;
    tetra_demod_default_config(&cfg);
    cfg.n_channels = 1;
    cfg.max_samples = STREAM_BUFFER_SIZE;   // the reference's count <= STREAM_BUFFER_SIZE
    cfg.symbolrate = symbolrate;
    cfg.samplerate = samplerate;
    cfg.rrc_tap_count = rrcTapCount;
    cfg.rrc_beta = rrcBeta;
    cfg.agc_rate = agcRate;
    cfg.costas_bandwidth = costasBandwidth;
    cfg.fll_bandwidth = fllBandwidth;
    cfg.omega_gain = omegaGain;
    cfg.mu_gain = muGain;
    cfg.omega_rel_limit = omegaRelLimit;
    cfg.flags |= TETRA_FLAG_REFERENCE_QUIRKS;   // this class IS the reference's block: reset() and the RRC setters behave like pi4dqpsk.cpp
    if (h_) { tetra_demod_destroy(h_); h_ = nullptr; }
    status_ = tetra_demod_create(&cfg, &h_);
    const int stride = tetra_demod_bits_stride(STREAM_BUFFER_SIZE);
    bitbuf_.assign(stride, 0);
    symbuf_.assign((size_t)stride, 0.f);   // stride/2 complex
    base_type::init(in);
}

void PI4DQPSK::set(int id, double v) {
    assert(base_type::_block_init);
    std::lock_guard<std::recursive_mutex> lck(base_type::ctrlMtx);
    base_type::tempStop();
    status_ = tetra_demod_set_param(h_, id, v);
    base_type::tempStart();
}
void PI4DQPSK::setSymbolrate(double v) { set(TETRA_PARAM_SYMBOLRATE, v); }
void PI4DQPSK::setSamplerate(double v) { set(TETRA_PARAM_SAMPLERATE, v); }
void PI4DQPSK::setRRCParams(int n, double beta) { set(TETRA_PARAM_RRC_TAP_COUNT, n); set(TETRA_PARAM_RRC_BETA, beta); }
void PI4DQPSK::setRRCTapCount(int n) { set(TETRA_PARAM_RRC_TAP_COUNT, n); }
void PI4DQPSK::setRRCBeta(double beta) { set(TETRA_PARAM_RRC_BETA, beta); }
void PI4DQPSK::setAGCRate(double v) { set(TETRA_PARAM_AGC_RATE, v); }
void PI4DQPSK::setCostasBandwidth(double v) { set(TETRA_PARAM_COSTAS_BANDWIDTH, v); }
void PI4DQPSK::setFllBandwidth(double v) { set(TETRA_PARAM_FLL_BANDWIDTH, v); }
void PI4DQPSK::setMMParams(double og, double mg, double lim) {
    set(TETRA_PARAM_OMEGA_GAIN, og); set(TETRA_PARAM_MU_GAIN, mg); set(TETRA_PARAM_OMEGA_REL_LIMIT, lim);
}
void PI4DQPSK::setOmegaGain(double v) { set(TETRA_PARAM_OMEGA_GAIN, v); }
void PI4DQPSK::setMuGain(double v) { set(TETRA_PARAM_MU_GAIN, v); }
void PI4DQPSK::setOmegaRelLimit(double v) { set(TETRA_PARAM_OMEGA_REL_LIMIT, v); }

void PI4DQPSK::reset() {
    assert(base_type::_block_init);
    std::lock_guard<std::recursive_mutex> lck(base_type::ctrlMtx);
    base_type::tempStop();
    status_ = tetra_demod_reset(h_, -1);
    base_type::tempStart();
}

int PI4DQPSK::process(int count, const complex_t* in, complex_t* out) {
    if (!h_) return -1;
    const int stride = tetra_demod_bits_stride(count);
    int32_t nb = 0;
    status_ = tetra_demod_process(h_, reinterpret_cast<const float*>(in), count, bitbuf_.data(), stride, &nb,
                                  symbuf_.data());
    if (status_ != TETRA_OK) return -1;
    const int nsym = nb / 2;
    std::memcpy(out, symbuf_.data(), sizeof(complex_t) * (size_t)nsym);
    bits_.assign(bitbuf_.begin(), bitbuf_.begin() + nb);
    return nsym;
}

PI4DQPSKBank::~PI4DQPSKBank() {
    if (h_) tetra_demod_destroy(h_);
}
int PI4DQPSKBank::init(const tetra_demod_config_t& cfg) {
    if (h_) { tetra_demod_destroy(h_); h_ = nullptr; }
    channels_ = cfg.n_channels;
    return tetra_demod_create(&cfg, &h_);
}
int PI4DQPSKBank::process(int count, const complex_t* in, uint8_t* bits, int32_t* nBits, complex_t* symbols) {
    if (!h_) return TETRA_ERR_ARG;
    return tetra_demod_process(h_, reinterpret_cast<const float*>(in), count, bits, tetra_demod_bits_stride(count), nBits,
                               reinterpret_cast<float*>(symbols));
}
int PI4DQPSKBank::reset(int channel) { return h_ ? tetra_demod_reset(h_, channel) : TETRA_ERR_ARG; }
int PI4DQPSKBank::setParam(int id, double v) { return h_ ? tetra_demod_set_param(h_, id, v) : TETRA_ERR_ARG; }

}  // namespace demod
}  // namespace dsp
