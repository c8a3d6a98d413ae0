// pi4dqpsk_gpu.h -- host-side mirror of the reference's demodulator block, backed by the HIP kernels
// through the C ABI (include/tetra_demod.h).  No DSP arithmetic happens on the host.
//
// dsp::demod::PI4DQPSK        same class name, init() signature, run()/process() contract, twelve setters and
//                             reset() as src/dsp/pi4dqpsk.h:27-81 / src/dsp/pi4dqpsk.cpp:11-140 -- replace that
//                             pair of files with this header + pi4dqpsk_gpu.cpp and the rest of the plugin
//                             (src/main.cpp:84-114) builds unchanged: `out` still carries the Costas-locked
//                             symbols that DQPSKSymbolExtractor slices (sign tests only, so its dibits equal
//                             the GPU's).  One instance = one channel = a C = 1 handle.
// dsp::demod::PI4DQPSKBank    the batched form the GPU is built for: C channels per call, returns the unpacked
//                             bit streams tetra_burst_sync_in() eats (src/decoder/src/phy/tetra_burst_sync.c:54),
//                             i.e. PI4DQPSK + DQPSKSymbolExtractor + BitUnpacker for every channel at once.
// dsp::demod::PI4DQPSKMultiBank  a bank spread over the GPUs of one node from ONE process: one handle, host thread and
//                             stream set per device, channel ranges, nothing exchanged.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/tetra_demod.h"
#include "dsp_compat.h"
#include "sdrpp_tables.h"

namespace dsp {
namespace demod {

// The GPU's per-symbol decisions on their way to the blocks downstream of PI4DQPSK.  In the reference the symbols leave
// PI4DQPSK through a stream and DQPSKSymbolExtractor / BitUnpacker -- separate blocks with their own worker threads -- slice
// and unpack them on the CPU (src/main.cpp:84-91).  The kernels have already made those decisions for the very same symbols
// (and keep DQPSKSymbolExtractor's statistic), so the GPU-backed mirrors of the two blocks (dqpsk_sym_extr_gpu.h,
// bit_unpacker_gpu.h) take them from here instead of recomputing: a thread-safe FIFO written by PI4DQPSK::process (the
// demodulator's worker thread), read by one consumer block's process() (that block's worker thread).  Decisions are stored as
// dibits (bit 1 = first bit; what DQPSKSymbolExtractor writes); the statistic as (symbol position, value) marks that a consumer
// applies once it has passed the position -- the reference updates standarderr / sync inside process() every 256 symbols
// (dqpsk_sym_extr.cpp:17-30).
//
// The FIFO is a side channel next to SDR++'s streams, and those can lose, repeat or hold back a buffer between
// mainDemodulator.out and the consumer (the splitter unbound / rebound, src/main.cpp:85-90; disable() / enable(), :130-167; a
// consumer that is not running while this queue overflows).  A positional FIFO would then hand every later symbol the decision
// of another one, silently and for ever.  So the consumers do not pop blindly: popAligned() is given what the consumer can
// tell about each symbol it was handed WITHOUT any DSP (the extractor: the dibit two sign tests imply; the unpacker: the dibit
// byte itself), verifies the queue head against it, and on a mismatch searches the queue for the offset at which the handed
// symbols line up again, discards what lies before it (counted: resyncs(), skippedSymbols()) and carries on; if the symbols
// are nowhere in the queue (a repeated buffer, a foreign stream) it consumes nothing and says so -- the consumer then falls
// back to its own sign tests for that buffer.
class DecisionTap {
public:
    void push(const uint8_t* bits, int nBits);      // the kernels' output row of one call: one bit per byte, MSB of each dibit first
    void mark(long long symbolPosition, float standarderr, bool sync);
    // Takes the decisions of the next nSym symbols: dibits[nSym] (bit 1 = first bit; what DQPSKSymbolExtractor writes) and/or
    // bits[2 nSym] (what BitUnpacker writes); applies every statistic mark passed on the way to *standarderr / *sync.
    // Returns nSym, or the (smaller) number available if the source has not produced that many -- a wiring error.
    int pop(int nSym, uint8_t* dibits, uint8_t* bits, float* standarderr, bool* sync);
    // pop() for a consumer that knows what to expect: expect[nSym] = the dibit each handed symbol implies (values 0..3);
    // expect[0] takes part in the head check only when firstIsReliable (the extractor's first dibit is a difference against the
    // last symbol it saw, meaningless right after a gap).  Queue head matches -> pops nSym, returns nSym.  Otherwise the first
    // offset s <= kSearchWindow with queue[s + i] == expect[i] for all i >= 1 (needs nSym >= kMinMatch to be trusted) -> discards
    // s symbols, pops nSym, counts one resync, returns nSym.  No such offset, too few decisions queued, or nothing to verify
    // the position with (one symbol whose dibit is not reliable) -> consumes nothing, returns -1.
    int popAligned(int nSym, const uint8_t* expect, bool firstIsReliable, uint8_t* dibits, uint8_t* bits, float* standarderr, bool* sync);
    long long consumedSymbols() const { return consumed_; }     // popped + discarded + dropped: the stream position of the queue head
    long long droppedSymbols() const { return dropped_; }      // decisions discarded because nobody took them (see push)
    long long resyncs() const { return resyncs_; }             // realignments popAligned() has made
    long long skippedSymbols() const { return skipped_; }      // decisions it discarded doing so
    long long queuedSymbols();
    void clear();      // empties the queue and restarts every counter: the source starts a new stream (PI4DQPSK::init)
    static constexpr size_t kMaxQueuedSymbols = (size_t)1 << 23;   // 8 Mi symbols = 7.8 minutes of one TETRA channel
    static constexpr int kMinMatch = 16;                             // symbols a realignment must agree on (chance match: 4^-15)
    static constexpr size_t kSearchWindow = kMaxQueuedSymbols;       // how far ahead popAligned() looks: the whole queue

private:
    struct Mark { long long pos; float err; bool sync; };
    void take(int n, uint8_t* dibits, uint8_t* bits, float* standarderr, bool* sync);      // m_ held
    std::mutex m_;
    std::deque<uint8_t> q_;      // dibits, oldest first
    std::deque<Mark> marks_;
    long long consumed_ = 0, dropped_ = 0, resyncs_ = 0, skipped_ = 0;
};

class PI4DQPSK : public Processor<complex_t, complex_t> {
    using base_type = Processor<complex_t, complex_t>;

public:
    PI4DQPSK() {}
    // Like the reference's constructor (src/dsp/pi4dqpsk.h:32), this one does NOT forward omegaRelLimit: init() runs with
    // its default 0.01 whatever is passed here.  (The plugin default-constructs and calls init, src/main.cpp:84.)
    PI4DQPSK(stream<complex_t>* in, double symbolrate, double samplerate, int rrcTapCount, double rrcBeta, double agcRate,
             double costasBandwidth, double fllBandwidth, double omegaGain, double muGain, double omegaRelLimit = 0.01) {
        (void)omegaRelLimit;
        init(in, symbolrate, samplerate, rrcTapCount, rrcBeta, agcRate, costasBandwidth, fllBandwidth, omegaGain, muGain);
    }
    ~PI4DQPSK();

    // src/dsp/pi4dqpsk.h:36.  Throws nothing; a failed GPU set-up leaves the block un-initialised (lastStatus() < 0).
    void init(stream<complex_t>* in, double symbolrate, double samplerate, int rrcTapCount, double rrcBeta, double agcRate,
              double costasBandwidth, double fllBandwidth, double omegaGain, double muGain, double omegaRelLimit = 0.01);

    // src/dsp/pi4dqpsk.h:38-50
    int run() override {
        int count = base_type::_in->read();
        if (count < 0) { return -1; }
        int outCount = process(count, base_type::_in->readBuf, base_type::out.writeBuf);
        base_type::_in->flush();
        if (outCount) {
            if (!base_type::out.swap(outCount)) { return -1; }
        }
        return outCount;
    }

    void setSymbolrate(double symbolrate);
    void setSamplerate(double samplerate);
    void setRRCParams(int rrcTapCount, double rrcBeta);
    void setRRCTapCount(int rrcTapCount);
    void setRRCBeta(int rrcBeta);      // an int, like src/dsp/pi4dqpsk.h:56: a fractional roll-off is truncated by the call itself
    void setAGCRate(double agcRate);
    void setCostasBandwidth(double bandwidth);
    void setFllBandwidth(double fllBandwidth);
    void setMMParams(double omegaGain, double muGain, double omegaRelLimit = 0.01);
    void setOmegaGain(double omegaGain);
    void setMuGain(double muGain);
    void setOmegaRelLimit(double omegaRelLimit);
    void reset();

    // src/dsp/pi4dqpsk.h:67: count input samples -> returns the number of symbols written to out
    // (in == out allowed, like the reference).  < 0 only if the GPU call failed (lastStatus()).
    int process(int count, const complex_t* in, complex_t* out);

    // Extras the GPU path gives for free: the unpacked bits of the last process() call
    // (= DQPSKSymbolExtractor + BitUnpacker output for the same symbols).
    const std::vector<uint8_t>& lastBits() const { return bits_; }
    // TETRA_OK, or the status of the last failing C-ABI call.  TETRA_ERR_OVERRUN after process() means the call delivered
    // its symbols but a NaN/Inf-poisoned stream filled the output row and the rest of the call's samples were dropped.
    int lastStatus() const { return status_; }
    // A FIFO of this demodulator's decisions for ONE downstream block (DQPSKSymbolExtractor::attach / BitUnpacker::attach call
    // this); every process() from then on feeds it.  Open taps before start().
    std::shared_ptr<DecisionTap> openTap();
    tetra_demod_t* handle() { return h_; }

    // Which code designed the tables the kernels run: true in a TETRA_WITH_SDRPP build -- SDR++'s own generators, called from
    // the included core headers (sdrpp_tables.h) in init() and in every re-designing setter, handed over as caller tables --,
    // false outside SDR++, where the library's restatement of them (csrc/design.hpp) is all there is.
    static bool tablesFromSdrpp();

private:
    void set(int id, double v);
    void resizeBuffers();
    int redesignRRC(int tapCount, double beta);          // returns the status of the table hand-over; setSymbolrate / setSamplerate / setRRCParams: taps::rootRaisedCosine + rrc.setTaps (pi4dqpsk.cpp:38-39,50-51,63-64)
    // what the reference keeps for its re-designs (pi4dqpsk.h:76-79)
    double _symbolrate = 0, _samplerate = 0, _rrcBeta = 0;
    int _rrcTapCount = 0;
    int maxStride_ = 0;
    tetra_demod_t* h_ = nullptr;
    int status_ = TETRA_ERR_ARG;
    std::vector<uint8_t> bitbuf_, bits_;
    std::vector<float> symbuf_;
    std::vector<std::shared_ptr<DecisionTap>> taps_;
    std::mutex tapMtx_;
    long long symbols_ = 0;      // symbols produced since init (the statistic's 256-symbol boundaries are counted from there)
};

class PI4DQPSKBank {
public:
    PI4DQPSKBank() {}
    ~PI4DQPSKBank();
    PI4DQPSKBank(const PI4DQPSKBank&) = delete;
    PI4DQPSKBank& operator=(const PI4DQPSKBank&) = delete;

    // cfg: tetra_demod_default_config() + n_channels / max_samples / layout / device.  Returns a TETRA_* status.
    int init(const tetra_demod_config_t& cfg);
    // in: n_channels x count complex samples in cfg.layout; bits: [n_channels][bitsStride(count)];
    // nBits: [n_channels].  Returns a TETRA_* status.
    int process(int count, const complex_t* in, uint8_t* bits, int32_t* nBits, complex_t* symbols = nullptr);
    // row length for calls of `count` samples with this bank's rates and timing-loop limits (tetra_demod_bits_stride_for)
    int bitsStride(int count) const { return tetra_demod_bits_stride_for(h_, count); }
    int reset(int channel = -1);
    int setParam(int paramId, double value);
    // DQPSKSymbolExtractor's public standarderr / sync (src/dsp/dqpsk_sym_extr.h:36-37) for every channel; needs
    // TETRA_FLAG_QUALITY in cfg.flags (TETRA_ERR_UNSUPPORTED otherwise).  Either pointer may be null.
    int quality(float* standarderr, uint8_t* sync);
    // The plugin's constellation tap (src/main.cpp:85-89 Reshaper keep 1024 / skip 0, :376-383 sink) for channels first .. first +
    // count - 1: blocks[count][1024] = each channel's last complete block of 1024 consecutive symbols (what the GUI's diagram would
    // hold), nBlocks[count] = blocks completed so far.  Needs TETRA_FLAG_CONSTELLATION in cfg.flags.  Either pointer may be null.
    int constellation(int first, int count, complex_t* blocks, int32_t* nBlocks);
    int channels() const { return channels_; }
    tetra_demod_t* handle() { return h_; }

private:
    tetra_demod_t* h_ = nullptr;
    int channels_ = 0;
};

// One node, G GPUs, one process: the channel axis cut into G contiguous ranges (sizes differ by at most one, the rule of
// shard.channel_range), each range a tetra_demod handle on its own device driven by its own host thread and its own HIP
// streams (tetra_demod_process_async).  Channels are independent chains, so nothing is exchanged between the shards -- the
// in-process counterpart of bench.py's one-rank-per-GPU launch.  `devices` may name a device more than once (two shards on
// one GPU: how the single-GPU test exercises two handles concurrently from two threads).
class PI4DQPSKMultiBank {
public:
    PI4DQPSKMultiBank() {}
    ~PI4DQPSKMultiBank();
    PI4DQPSKMultiBank(const PI4DQPSKMultiBank&) = delete;
    PI4DQPSKMultiBank& operator=(const PI4DQPSKMultiBank&) = delete;

    // cfg.n_channels = ALL channels (>= devices.size()); cfg.device is ignored; channel-major layout only.
    int init(const tetra_demod_config_t& cfg, const std::vector<int>& devices);
    // in: [n_channels][count] (page-locked memory makes the shards' copies overlap); bits: [n_channels][bitsStride(count)];
    // nBits: [n_channels].  Runs every shard concurrently and returns when all are done: TETRA_OK or the first failure.
    int process(int count, const complex_t* in, uint8_t* bits, int32_t* nBits);
    // The same with interleaved int16 IQ as SDR hardware delivers it (converted on the GPUs as x / 32768; half the PCIe bytes).
    int processCS16(int count, const int16_t* in, uint8_t* bits, int32_t* nBits);
    // Input and output already on the GPUs (what a channeliser or a capture DMA leaves there; the path the throughput metric
    // is quoted on): per shard s, in[s] = device pointer on devices[s] to that shard's [count_s channels][count] samples,
    // bits[s] / nBits[s] = device rows [count_s][bitsStride(count)] / [count_s].  Every shard's launch goes onto its own
    // stream on its own device from its own host thread; returns when all have finished (TETRA_OK or the first failure).
    int processDevice(int count, const complex_t* const* in, uint8_t* const* bits, int32_t* const* nBits);
    int bitsStride(int count) const { return shards_.empty() ? TETRA_ERR_ARG : tetra_demod_bits_stride_for(shards_[0]->h, count); }
    int reset();
    int setParam(int paramId, double value);
    // DQPSKSymbolExtractor's public standarderr / sync (src/dsp/dqpsk_sym_extr.h:36-37) for every channel; needs
    // TETRA_FLAG_QUALITY in cfg.flags (TETRA_ERR_UNSUPPORTED otherwise).  Either pointer may be null.
    int quality(float* standarderr, uint8_t* sync);
    // PI4DQPSKBank::constellation over the shards (channels first .. first + count - 1 of the whole bank)
    int constellation(int first, int count, complex_t* blocks, int32_t* nBlocks);
    int channels() const { return channels_; }
    int shards() const { return (int)shards_.size(); }
    // channel range [first, first + count) and device of a shard
    void shardInfo(int shard, int& first, int& count, int& device) const;

private:
    struct Shard {
        tetra_demod_t* h = nullptr;
        int device = 0, first = 0, count = 0;
        std::thread worker;
        int status = TETRA_OK;
    };
    struct Job {
        int count = 0, format = TETRA_IQ_CF32;
        const void* in = nullptr; uint8_t* bits = nullptr; int32_t* nBits = nullptr;                              // host buffers, all channels
        const complex_t* const* dIn = nullptr; uint8_t* const* dBits = nullptr; int32_t* const* dNBits = nullptr;   // or per-shard device buffers
    };
    int run(const Job& j);
    void workerLoop(Shard* s);
    void shutdown();
    std::vector<std::unique_ptr<Shard>> shards_;
    int channels_ = 0;
    std::mutex m_;
    std::condition_variable cv_;
    Job job_;
    long long epoch_ = 0;      // bumped per process() call; each worker runs every epoch once
    int pending_ = 0;
    bool quit_ = false;
};

}  // namespace demod
}  // namespace dsp
