// tetra_rx_bank.h -- C++ face of the receive chain behind one handle (include/tetra_rx.h), beside PI4DQPSKBank (pi4dqpsk_gpu.h):
// where PI4DQPSKBank stops at bits, TetraRxBank delivers what the reference's decoder block delivers to its upper MAC --
// tetra_burst_sync_in -> tetra_burst_rx_cb -> tp_sap_udata_ind (src/decoder/src/phy/tetra_burst_sync.c:54-155,
// phy/tetra_burst.c:343-393, lower_mac/tetra_lower_mac.c:148-275) -- for C channels on one GPU: decoded type-1 blocks with CRC
// verdict, TDMA time and channel, plus the per-channel cell state (tcd / t_phy_state).  Header-only; links libtetra_demod_hip.so.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "tetra_rx.h"

namespace dsp {
namespace demod {

class TetraRxBank {
public:
    struct Blocks {                       // one kind's hand-overs of one call, in (channel, frame) order
        std::vector<tetra_rx_block_t> info;
        std::vector<uint8_t> type1;       // [info.size()][bitsPerBlock], one bit per byte
        int bitsPerBlock = 0;
        const uint8_t* bits(size_t row) const { return type1.data() + row * (size_t)bitsPerBlock; }
    };

    TetraRxBank() {}
    ~TetraRxBank() { if (h_) tetra_rx_destroy(h_); }
    TetraRxBank(const TetraRxBank&) = delete;
    TetraRxBank& operator=(const TetraRxBank&) = delete;

    // cfg: tetra_rx_default_config() + demod.n_channels / max_samples / layout / device (+ kinds, flags).  Returns a TETRA_* status.
    int init(const tetra_rx_config_t& cfg) {
        if (h_) { tetra_rx_destroy(h_); h_ = nullptr; }
        channels_ = cfg.demod.n_channels;
        return tetra_rx_create(&cfg, &h_);
    }
    // in: n_channels x count complex samples (host) in cfg.demod.layout.  Enqueues the whole chain for this block and returns; the
    // blocks of the call BEFORE it are then complete or about to be (fetch(.., 1) waits for exactly that call's tail).
    int process(int count, const float* iq) { return tetra_rx_process(h_, iq, count); }
    // the same with the samples already on the GPU (what a channeliser leaves there), enqueued on the caller's stream
    int processDevice(int count, const float* dIq, void* hipStream) { return tetra_rx_process_device(h_, dIq, count, hipStream); }
    int wait() { return tetra_rx_wait(h_); }
    int reset() { return tetra_rx_reset(h_); }
    // Decoded blocks of one kind (TETRA_RX_KIND_*) of the latest call (which = 0) or the one before it (1).
    int fetch(int kind, Blocks& out, int which = 0) {
        const int nb = tetra_rx_type1_bits(kind);
        if (nb < 0) return nb;
        int n = 0;
        int rc = tetra_rx_fetch(h_, which, kind, nullptr, nullptr, 0, 0, &n);
        if (rc != TETRA_OK) return rc;
        out.bitsPerBlock = nb;
        out.info.resize((size_t)n);
        out.type1.resize((size_t)n * (size_t)nb);
        if (n == 0) return TETRA_OK;
        return tetra_rx_fetch(h_, which, kind, out.info.data(), out.type1.data(), nb, n, &n);
    }
    // tcd / t_phy_state of every channel (tetra_lower_mac.c:116, tetra_burst_sync.c:34 -- one per channel here)
    int cells(std::vector<tetra_lmac_cell_state_t>& out) {
        out.resize((size_t)channels_);
        return tetra_rx_get_cell(h_, 0, channels_, out.data());
    }
    int syncStates(std::vector<tetra_bsync_state_t>& out) {
        out.resize((size_t)channels_);
        return tetra_rx_get_sync_state(h_, 0, channels_, out.data());
    }
    // the PI4DQPSK setters of the demodulators inside (tetra_demod_set_param ids); waits for work in flight first
    int setParam(int paramId, double value) {
        const int rc = tetra_rx_wait(h_);
        if (rc != TETRA_OK && rc != TETRA_ERR_OVERRUN) return rc;
        return tetra_demod_set_param(tetra_rx_demod(h_), paramId, value);
    }
    int channels() const { return channels_; }
    tetra_rx_t* handle() { return h_; }

private:
    tetra_rx_t* h_ = nullptr;
    int channels_ = 0;
};

// The chain over several GPUs of a node, beside PI4DQPSKMultiBank: channels are independent receivers, so GPU g takes the channel
// range [g C / G, (g + 1) C / G) (the split of shard.py / bench.py --gpus N) and nothing crosses between the GPUs.  tetra_rx_process*
// only ENQUEUES (every handle has its own streams on its own device), so one host thread drives all shards; blocks come back with
// their channel numbers in the whole bank's numbering, shard after shard = (channel, frame) order.
class TetraRxMultiBank {
public:
    TetraRxMultiBank() {}
    TetraRxMultiBank(const TetraRxMultiBank&) = delete;
    TetraRxMultiBank& operator=(const TetraRxMultiBank&) = delete;

    // cfg.demod.n_channels = ALL channels (>= devices.size()); cfg.demod.device is ignored; one shard per entry of `devices` (an
    // ordinal may repeat: two shards on one GPU, e.g. for tests on a one-GPU box).
    int init(const tetra_rx_config_t& cfg, const std::vector<int>& devices) {
        shards_.clear();
        channels_ = cfg.demod.n_channels;
        const int G = (int)devices.size();
        if (G < 1 || channels_ < G) return TETRA_ERR_ARG;
        for (int g = 0; g < G; g++) {
            std::unique_ptr<Shard> s(new Shard());
            s->first = (int)((long long)g * channels_ / G);
            s->count = (int)((long long)(g + 1) * channels_ / G) - s->first;
            tetra_rx_config_t c = cfg;
            c.demod.n_channels = s->count;
            c.demod.device = devices[(size_t)g];
            const int rc = s->bank.init(c);
            if (rc != TETRA_OK) { shards_.clear(); return rc; }
            shards_.push_back(std::move(s));
        }
        return TETRA_OK;
    }
    // Samples already on the GPUs: dIq[s] = device pointer on shard s's device to its [count_s channels x count] samples (in
    // cfg.demod.layout), streams[s] = a HIP stream of that device (or null).  Enqueues on every shard and returns.
    int processDevice(int count, const float* const* dIq, void* const* streams) {
        for (size_t s = 0; s < shards_.size(); s++) {
            const int rc = shards_[s]->bank.processDevice(count, dIq[s], streams ? streams[s] : nullptr);
            if (rc != TETRA_OK) return rc;
        }
        return TETRA_OK;
    }
    // Host samples, channel major [n_channels][count] complex64: every shard copies its rows in and enqueues its chain.
    int process(int count, const float* iq) {
        for (auto& s : shards_) {
            const int rc = s->bank.process(count, iq + (size_t)2 * (size_t)s->first * (size_t)count);
            if (rc != TETRA_OK) return rc;
        }
        return TETRA_OK;
    }
    int wait() {
        int first = TETRA_OK;
        for (auto& s : shards_) { const int rc = s->bank.wait(); if (first == TETRA_OK) first = rc; }
        return first;
    }
    int reset() {
        for (auto& s : shards_) { const int rc = s->bank.reset(); if (rc != TETRA_OK) return rc; }
        return TETRA_OK;
    }
    int fetch(int kind, TetraRxBank::Blocks& out, int which = 0) {
        out.info.clear(); out.type1.clear();
        TetraRxBank::Blocks part;
        for (auto& s : shards_) {
            const int rc = s->bank.fetch(kind, part, which);
            if (rc != TETRA_OK) return rc;
            out.bitsPerBlock = part.bitsPerBlock;
            for (auto& b : part.info) b.channel += s->first;
            out.info.insert(out.info.end(), part.info.begin(), part.info.end());
            out.type1.insert(out.type1.end(), part.type1.begin(), part.type1.end());
        }
        return TETRA_OK;
    }
    int cells(std::vector<tetra_lmac_cell_state_t>& out) {
        out.clear();
        std::vector<tetra_lmac_cell_state_t> part;
        for (auto& s : shards_) {
            const int rc = s->bank.cells(part);
            if (rc != TETRA_OK) return rc;
            out.insert(out.end(), part.begin(), part.end());
        }
        return TETRA_OK;
    }
    int channels() const { return channels_; }
    int shards() const { return (int)shards_.size(); }
    void shardInfo(int shard, int& first, int& count) const { first = shards_[(size_t)shard]->first; count = shards_[(size_t)shard]->count; }
    TetraRxBank& shard(int s) { return shards_[(size_t)s]->bank; }

private:
    struct Shard {
        TetraRxBank bank;
        int first = 0, count = 0;
    };
    std::vector<std::unique_ptr<Shard>> shards_;
    int channels_ = 0;
};

}  // namespace demod
}  // namespace dsp
