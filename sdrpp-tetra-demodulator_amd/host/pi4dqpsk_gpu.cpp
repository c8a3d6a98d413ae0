// pi4dqpsk_gpu.cpp -- see pi4dqpsk_gpu.h.  Thin glue over the C ABI; mirrors src/dsp/pi4dqpsk.cpp's control flow
// (ctrlMtx + tempStop()/tempStart() around anything that re-designs filters, src/dsp/pi4dqpsk.cpp:32-42).
#include "pi4dqpsk_gpu.h"

#include <cassert>
#include <cstring>

namespace dsp {
namespace demod {

void DecisionTap::push(const uint8_t* bits, int nBits) {
    std::lock_guard<std::mutex> l(m_);
    for (int i = 0; i + 1 < nBits; i += 2) q_.push_back((uint8_t)((bits[i] << 1) | bits[i + 1]));      // re-packing of the GPU's decisions, no arithmetic on samples
    // A consumer that was attached but is not running would let the queue grow for ever (SDR++'s streams block their writer
    // instead; this side channel must not).  Beyond kMaxQueuedSymbols the OLDEST decisions are dropped and counted as consumed,
    // so positions (statistic marks) stay aligned; a consumer that comes back late realigns on the newest ones (popAligned).
    if (q_.size() > kMaxQueuedSymbols) {
        const size_t drop = q_.size() - kMaxQueuedSymbols;
        q_.erase(q_.begin(), q_.begin() + (std::ptrdiff_t)drop);
        consumed_ += (long long)drop;
        dropped_ += (long long)drop;
        while (!marks_.empty() && marks_.front().pos <= consumed_) marks_.pop_front();
    }
}
void DecisionTap::mark(long long pos, float err, bool sync) {
    std::lock_guard<std::mutex> l(m_);
    marks_.push_back(Mark{ pos, err, sync });
}
void DecisionTap::take(int n, uint8_t* dibits, uint8_t* bits, float* standarderr, bool* sync) {
    for (int i = 0; i < n; i++) {
        const uint8_t d = q_[(size_t)i];
        if (dibits) dibits[i] = d;
        if (bits) { bits[2 * i] = (uint8_t)(d >> 1); bits[2 * i + 1] = (uint8_t)(d & 1); }
    }
    q_.erase(q_.begin(), q_.begin() + n);
    consumed_ += n;
    while (!marks_.empty() && marks_.front().pos <= consumed_) {
        if (standarderr) *standarderr = marks_.front().err;
        if (sync) *sync = marks_.front().sync;
        marks_.pop_front();
    }
}
int DecisionTap::pop(int nSym, uint8_t* dibits, uint8_t* bits, float* standarderr, bool* sync) {
    std::lock_guard<std::mutex> l(m_);
    const long long have = (long long)q_.size();
    const int n = have < nSym ? (int)have : nSym;
    take(n, dibits, bits, standarderr, sync);
    return n;
}
int DecisionTap::popAligned(int nSym, const uint8_t* expect, bool firstIsReliable, uint8_t* dibits, uint8_t* bits, float* standarderr,
                            bool* sync) {
    std::lock_guard<std::mutex> l(m_);
    if (nSym <= 0) return 0;
    const size_t have = q_.size(), n = (size_t)nSym;
    if (have < n) return -1;
    if (!firstIsReliable && nSym < 2) return -1;      // nothing to verify the position with: never pop blindly
    auto matches = [&](size_t s, size_t from) {
        for (size_t i = from; i < n; i++)
            if (q_[s + i] != expect[i]) return false;
        return true;
    };
    if (matches(0, firstIsReliable ? 0 : 1)) { take(nSym, dibits, bits, standarderr, sync); return nSym; }
    if (nSym < kMinMatch) return -1;            // too short to tell a realignment from a coincidence
    const size_t last = have - n < kSearchWindow ? have - n : kSearchWindow;
    for (size_t s = 0; s <= last; s++) {
        if (!matches(s, 1)) continue;
        // the symbols this consumer was handed sit s decisions into the queue: what lies before them never reached it
        // (s = 0: only the first dibit disagreed -- the consumer's own idea of the symbol before this buffer was stale)
        if (s > 0) {
            q_.erase(q_.begin(), q_.begin() + (std::ptrdiff_t)s);
            consumed_ += (long long)s;
            skipped_ += (long long)s;
            resyncs_++;
        }
        take(nSym, dibits, bits, standarderr, sync);
        return nSym;
    }
    return -1;
}
long long DecisionTap::queuedSymbols() {
    std::lock_guard<std::mutex> l(m_);
    return (long long)q_.size();
}
void DecisionTap::clear() {
    std::lock_guard<std::mutex> l(m_);
    q_.clear();
    marks_.clear();
    consumed_ = dropped_ = resyncs_ = skipped_ = 0;
}

std::shared_ptr<DecisionTap> PI4DQPSK::openTap() {
    std::lock_guard<std::mutex> l(tapMtx_);
    taps_.push_back(std::make_shared<DecisionTap>());
    return taps_.back();
}

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
    // DQPSKSymbolExtractor's statistic is kept on the GPU for the mirror of that block (dqpsk_sym_extr_gpu.h).  A create-time
    // flag, and the plugin attaches the extractor AFTER this init (src/main.cpp:84,90), so it is always on: one small k_quality
    // launch per call; the read-back (tetra_demod_get_quality) happens only while a tap is open (process()).
    cfg.flags |= TETRA_FLAG_QUALITY;
    _symbolrate = symbolrate;
    _samplerate = samplerate;
    _rrcTapCount = rrcTapCount;
    _rrcBeta = rrcBeta;
#ifdef TETRA_WITH_SDRPP
    // SDR++'s own generators design the tables (sdrpp_tables.h); the library's restatement (csrc/design.hpp) is not consulted
    const std::vector<float> rrcTaps = sdrpp_tables::rrc(_rrcTapCount, _rrcBeta, _symbolrate, _samplerate);              // pi4dqpsk.cpp:18
    const std::vector<float> beTaps = sdrpp_tables::bandedge(_rrcTapCount, (float)_rrcBeta, (int)_symbolrate, (int)_samplerate);   // pi4dqpsk.cpp:17 -> fll.cpp:10-18
    const std::vector<float> bank = sdrpp_tables::interpBank();                                                       // pi4dqpsk.cpp:22 -> complex_fd.cpp:23
    if ((int)rrcTaps.size() == rrcTapCount && bank.size() == 128 * 8) {
        cfg.rrc_taps = rrcTaps.data();
        cfg.bandedge_taps = beTaps.data();
        cfg.interp_bank = bank.data();
    }
#endif
    {   // decisions of the old handle are not this stream's: every open tap starts over with the new handle
        std::lock_guard<std::mutex> l(tapMtx_);
        symbols_ = 0;
        for (auto& t : taps_) t->clear();
    }
    if (h_) { tetra_demod_destroy(h_); h_ = nullptr; }
    status_ = tetra_demod_create(&cfg, &h_);
    maxStride_ = 0;
    resizeBuffers();
    base_type::init(in);
}

// Output rows for the largest call (count <= STREAM_BUFFER_SIZE) at the handle's current rates: grown after every setter
// that can slow the timing loop down (more symbols per sample).
void PI4DQPSK::resizeBuffers() {
    if (!h_) return;
    const int stride = tetra_demod_bits_stride_for(h_, STREAM_BUFFER_SIZE);
    if (stride > maxStride_) {
        maxStride_ = stride;
        bitbuf_.assign((size_t)stride, 0);
        symbuf_.assign((size_t)stride, 0.f);   // stride/2 complex
    }
}

void PI4DQPSK::set(int id, double v) {
    assert(base_type::_block_init);
    std::lock_guard<std::recursive_mutex> lck(base_type::ctrlMtx);
    base_type::tempStop();
    status_ = tetra_demod_set_param(h_, id, v);
    resizeBuffers();
    base_type::tempStart();
}
bool PI4DQPSK::tablesFromSdrpp() {
#ifdef TETRA_WITH_SDRPP
    return true;
#else
    return false;
#endif
}

// The tail of the three re-designing setters (pi4dqpsk.cpp:37-39, 49-51, 62-64): new RRC taps, FIR::setTaps.  In an SDR++ build
// the taps come from SDR++'s taps::rootRaisedCosine and go in as a caller table; called with ctrlMtx held and the block stopped.
int PI4DQPSK::redesignRRC(int tapCount, double beta) {
#ifdef TETRA_WITH_SDRPP
    const std::vector<float> rrcTaps = sdrpp_tables::rrc(tapCount, beta, _symbolrate, _samplerate);
    const int rc = tetra_demod_set_tables(h_, rrcTaps.data(), (int)rrcTaps.size(), nullptr, 0, nullptr);
    if (status_ == TETRA_OK) status_ = rc;
    return rc;
#else
    (void)tapCount; (void)beta;
    return TETRA_OK;
#endif
}

// pi4dqpsk.cpp:32-54: RRC re-design + COMPLEX_FD::setOmega.  (With a caller's RRC table on the handle -- the SDR++ build -- the
// library's rate setter does the timing-loop half and leaves the table to redesignRRC(); outside SDR++ it re-designs itself.)
void PI4DQPSK::setSymbolrate(double v) {
    assert(base_type::_block_init);
    std::lock_guard<std::recursive_mutex> lck(base_type::ctrlMtx);
    base_type::tempStop();
    status_ = tetra_demod_set_param(h_, TETRA_PARAM_SYMBOLRATE, v);
    if (status_ == TETRA_OK) { _symbolrate = v; redesignRRC(_rrcTapCount, _rrcBeta); }
    resizeBuffers();
    base_type::tempStart();
}
void PI4DQPSK::setSamplerate(double v) {
    assert(base_type::_block_init);
    std::lock_guard<std::recursive_mutex> lck(base_type::ctrlMtx);
    base_type::tempStop();
    status_ = tetra_demod_set_param(h_, TETRA_PARAM_SAMPLERATE, v);
    if (status_ == TETRA_OK) { _samplerate = v; redesignRRC(_rrcTapCount, _rrcBeta); }
    resizeBuffers();
    base_type::tempStart();
}
// pi4dqpsk.cpp:56-66: tap count and roll-off (the double, untruncated) in one re-design of the RRC
void PI4DQPSK::setRRCParams(int n, double beta) {
    assert(base_type::_block_init);
    std::lock_guard<std::recursive_mutex> lck(base_type::ctrlMtx);
    base_type::tempStop();
#ifdef TETRA_WITH_SDRPP
    if (n < 2 || n > TETRA_DEMOD_MAX_TAPS) status_ = TETRA_ERR_UNSUPPORTED;      // (the library's own limit; nothing changes)
    else {
        // the mirror's members follow the handle: committed only once the handle has taken the new table
        status_ = TETRA_OK;
        if (redesignRRC(n, beta) == TETRA_OK) { _rrcTapCount = n; _rrcBeta = beta; }
    }
#else
    status_ = tetra_demod_set_rrc_params(h_, n, beta);
    if (status_ == TETRA_OK) { _rrcTapCount = n; _rrcBeta = beta; }
#endif
    base_type::tempStart();
}
// pi4dqpsk.cpp:68-74
void PI4DQPSK::setRRCTapCount(int n) { setRRCParams(n, _rrcBeta); }
// ... through the int parameter of pi4dqpsk.h:56: setRRCBeta(0.35) designs with roll-off 0, like the reference
void PI4DQPSK::setRRCBeta(int beta) { setRRCParams(_rrcTapCount, beta); }
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
    const int stride = tetra_demod_bits_stride_for(h_, count);
    if (stride < 0 || stride > maxStride_) { status_ = TETRA_ERR_SIZE; return -1; }
    int32_t nb = 0;
    status_ = tetra_demod_process(h_, reinterpret_cast<const float*>(in), count, bitbuf_.data(), stride, &nb,
                                  symbuf_.data());
    // TETRA_ERR_OVERRUN: the symbols up to the row's capacity were delivered (a NaN/Inf-poisoned stream); the reference's
    // process() cannot fail, so the block keeps running and lastStatus() tells
    if (status_ != TETRA_OK && status_ != TETRA_ERR_OVERRUN) return -1;
    const int nsym = nb / 2;
    std::memcpy(out, symbuf_.data(), sizeof(complex_t) * (size_t)nsym);
    bits_.assign(bitbuf_.begin(), bitbuf_.begin() + nb);
    // the downstream mirrors: the decisions of these symbols, and -- when the call crossed a 256-symbol boundary, where the
    // reference publishes its statistic (dqpsk_sym_extr.cpp:17-30) -- the statistic the GPU has brought up to that boundary
    {
        std::lock_guard<std::mutex> l(tapMtx_);
        const long long before = symbols_;
        symbols_ += nsym;
        if (!taps_.empty()) {
            const long long boundary = symbols_ / 256 * 256;
            float err = 0.f;
            uint8_t sy = 0;
            const bool fresh = boundary > before && tetra_demod_get_quality(h_, &err, &sy) == TETRA_OK;
            for (auto& t : taps_) {
                t->push(bitbuf_.data(), nb);
                if (fresh) t->mark(boundary, err, sy != 0);
            }
        }
    }
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
    return tetra_demod_process(h_, reinterpret_cast<const float*>(in), count, bits, tetra_demod_bits_stride_for(h_, count), nBits,
                               reinterpret_cast<float*>(symbols));
}
int PI4DQPSKBank::reset(int channel) { return h_ ? tetra_demod_reset(h_, channel) : TETRA_ERR_ARG; }
int PI4DQPSKBank::setParam(int id, double v) { return h_ ? tetra_demod_set_param(h_, id, v) : TETRA_ERR_ARG; }
int PI4DQPSKBank::quality(float* standarderr, uint8_t* sync) { return h_ ? tetra_demod_get_quality(h_, standarderr, sync) : TETRA_ERR_ARG; }

int PI4DQPSKBank::constellation(int first, int count, complex_t* blocks, int32_t* nBlocks) {
    return h_ ? tetra_demod_get_constellation(h_, first, count, reinterpret_cast<float*>(blocks), nBlocks) : TETRA_ERR_ARG;
}

PI4DQPSKMultiBank::~PI4DQPSKMultiBank() { shutdown(); }

void PI4DQPSKMultiBank::shutdown() {
    {
        std::lock_guard<std::mutex> l(m_);
        quit_ = true;
    }
    cv_.notify_all();
    for (auto& s : shards_)
        if (s->worker.joinable()) s->worker.join();
    for (auto& s : shards_)
        if (s->h) tetra_demod_destroy(s->h);
    shards_.clear();
    quit_ = false;
    epoch_ = 0;
    pending_ = 0;
}

int PI4DQPSKMultiBank::init(const tetra_demod_config_t& cfg, const std::vector<int>& devices) {
    shutdown();
    const int G = (int)devices.size();
    if (G < 1 || cfg.n_channels < G) return TETRA_ERR_ARG;
    if (cfg.layout != TETRA_LAYOUT_CHANNEL_MAJOR) return TETRA_ERR_UNSUPPORTED;   // a shard's rows must be one contiguous block
    channels_ = cfg.n_channels;
    const int q = channels_ / G, r = channels_ % G;
    for (int g = 0; g < G; g++) {
        std::unique_ptr<Shard> s(new Shard());
        s->device = devices[g];
        s->first = g * q + (g < r ? g : r);
        s->count = q + (g < r ? 1 : 0);
        tetra_demod_config_t c = cfg;
        c.n_channels = s->count;
        c.device = s->device;
        const int rc = tetra_demod_create(&c, &s->h);
        if (rc != TETRA_OK) { shutdown(); return rc; }
        shards_.push_back(std::move(s));
    }
    for (auto& s : shards_) s->worker = std::thread([this, p = s.get()] { workerLoop(p); });
    return TETRA_OK;
}

void PI4DQPSKMultiBank::workerLoop(Shard* s) {
    long long seen = 0;
    for (;;) {
        Job j;
        int shard = 0;
        {
            std::unique_lock<std::mutex> l(m_);
            cv_.wait(l, [&] { return quit_ || epoch_ != seen; });
            if (quit_) return;
            seen = epoch_;
            j = job_;
            while (shards_[(size_t)shard].get() != s) shard++;
        }
        // this shard's rows of the caller's buffers; its own handle, device and streams
        const int stride = tetra_demod_bits_stride_for(s->h, j.count);
        int rc;
        if (j.dIn) {
            // samples already on this shard's GPU: one launch on the handle's own stream, waited for here
            rc = tetra_demod_process_resident(s->h, reinterpret_cast<const float*>(j.dIn[shard]), j.count, j.dBits[shard], stride,
                                              j.dNBits[shard], nullptr);
        } else {
            const size_t elem = j.format == TETRA_IQ_CS16 ? 2 * sizeof(int16_t) : sizeof(complex_t);
            rc = stride < 0 ? stride
                            : tetra_demod_process_async(s->h, static_cast<const uint8_t*>(j.in) + elem * (size_t)s->first * (size_t)j.count,
                                                        j.format, j.count, j.bits + (size_t)s->first * (size_t)stride, stride,
                                                        j.nBits + s->first);
            if (rc == TETRA_OK) rc = tetra_demod_wait(s->h);
        }
        {
            std::lock_guard<std::mutex> l(m_);
            s->status = rc;
            pending_--;
        }
        cv_.notify_all();
    }
}

int PI4DQPSKMultiBank::run(const Job& j) {
    std::unique_lock<std::mutex> l(m_);
    job_ = j;
    pending_ = (int)shards_.size();
    epoch_++;
    cv_.notify_all();
    cv_.wait(l, [&] { return pending_ == 0; });
    int overrun = TETRA_OK;
    for (auto& s : shards_) {
        if (s->status == TETRA_ERR_OVERRUN) overrun = TETRA_ERR_OVERRUN;      // every shard delivered; report it last
        else if (s->status != TETRA_OK) return s->status;
    }
    return overrun;
}

int PI4DQPSKMultiBank::process(int count, const complex_t* in, uint8_t* bits, int32_t* nBits) {
    if (shards_.empty() || !in || !bits || !nBits) return TETRA_ERR_ARG;
    Job j;
    j.count = count; j.format = TETRA_IQ_CF32; j.in = in; j.bits = bits; j.nBits = nBits;
    return run(j);
}

int PI4DQPSKMultiBank::processCS16(int count, const int16_t* in, uint8_t* bits, int32_t* nBits) {
    if (shards_.empty() || !in || !bits || !nBits) return TETRA_ERR_ARG;
    Job j;
    j.count = count; j.format = TETRA_IQ_CS16; j.in = in; j.bits = bits; j.nBits = nBits;
    return run(j);
}

int PI4DQPSKMultiBank::processDevice(int count, const complex_t* const* in, uint8_t* const* bits, int32_t* const* nBits) {
    if (shards_.empty() || !in || !bits || !nBits) return TETRA_ERR_ARG;
    for (size_t g = 0; g < shards_.size(); g++)
        if (!in[g] || !bits[g] || !nBits[g]) return TETRA_ERR_ARG;
    Job j;
    j.count = count; j.dIn = in; j.dBits = bits; j.dNBits = nBits;
    return run(j);
}

// DQPSKSymbolExtractor's standarderr / sync of every channel, shard by shard into the caller's arrays
int PI4DQPSKMultiBank::quality(float* standarderr, uint8_t* sync) {
    if (shards_.empty()) return TETRA_ERR_ARG;
    for (auto& s : shards_) {
        const int rc = tetra_demod_get_quality(s->h, standarderr ? standarderr + s->first : nullptr, sync ? sync + s->first : nullptr);
        if (rc != TETRA_OK) return rc;
    }
    return TETRA_OK;
}

// the constellation blocks of channels [first, first + count): each shard's part of the range into the caller's arrays
int PI4DQPSKMultiBank::constellation(int first, int count, complex_t* blocks, int32_t* nBlocks) {
    if (shards_.empty() || first < 0 || count < 0 || first > channels_ || count > channels_ - first) return TETRA_ERR_ARG;
    for (auto& s : shards_) {
        const int lo = first > s->first ? first : s->first;
        const int hi = first + count < s->first + s->count ? first + count : s->first + s->count;
        if (hi <= lo) continue;
        const int rc = tetra_demod_get_constellation(s->h, lo - s->first, hi - lo,
                                                     blocks ? reinterpret_cast<float*>(blocks + (size_t)(lo - first) * TETRA_CONSTELLATION_SYMBOLS) : nullptr,
                                                     nBlocks ? nBlocks + (lo - first) : nullptr);
        if (rc != TETRA_OK) return rc;
    }
    return TETRA_OK;
}

int PI4DQPSKMultiBank::reset() {
    if (shards_.empty()) return TETRA_ERR_ARG;
    for (auto& s : shards_) { const int rc = tetra_demod_reset(s->h, -1); if (rc != TETRA_OK) return rc; }
    return TETRA_OK;
}

int PI4DQPSKMultiBank::setParam(int id, double v) {
    if (shards_.empty()) return TETRA_ERR_ARG;
    for (auto& s : shards_) { const int rc = tetra_demod_set_param(s->h, id, v); if (rc != TETRA_OK) return rc; }
    return TETRA_OK;
}

void PI4DQPSKMultiBank::shardInfo(int shard, int& first, int& count, int& device) const {
    const Shard& s = *shards_.at((size_t)shard);
    first = s.first; count = s.count; device = s.device;
}

}  // namespace demod
}  // namespace dsp
