// tap_selftest.h -- the DecisionTap alignment scenarios (host logic only, no GPU), shared by tests/host/test_block.cpp and the
// sanitizer build tests/san/san_host.cpp.
#pragma once
#include <cstdio>
#include <memory>
#include <vector>

#include "../../sdrpp-tetra-demodulator_amd/host/pi4dqpsk_gpu.h"
#include "../../sdrpp-tetra-demodulator_amd/host/dqpsk_sym_extr_gpu.h"
#include "../../sdrpp-tetra-demodulator_amd/host/bit_unpacker_gpu.h"

// The side channel checks itself (host logic only, no GPU): a DecisionTap fed by hand with the decisions of a synthetic symbol
// stream, the extractor / unpacker mirrors handed that stream in buffers with one LOST, one handed TWICE, the consumer stopped
// for a while, the queue overflowed -- the dibits / bits that leave the blocks must be the stream's own for every buffer they
// were handed (the first dibit of a repeated buffer excepted: a difference against that buffer's own last symbol, what the
// reference's block computes there too), and the statistic marks must arrive at their positions.
namespace {
struct SynthStream {
    std::vector<dsp::complex_t> sym;
    std::vector<uint8_t> dibit;      // the reference's decisions, dqpsk_sym_extr.cpp:32-52 (prev starts at 0)
    explicit SynthStream(size_t n, unsigned seed = 12345) : sym(n), dibit(n) {
        unsigned x = seed;
        auto rnd = [&] { x = x * 1664525u + 1013904223u; return x >> 8; };
        int prev = 0;
        for (size_t i = 0; i < n; i++) {
            const unsigned r = rnd(), r2 = rnd();      // (the top bits: an LCG's low bits have short periods and the stream must not repeat)
            const float re = ((r & 0x800000) ? -0.7071f : 0.7071f) + 0.2f * ((float)((r >> 2) & 1023) / 1023.f - 0.5f);
            const float im = ((r2 & 0x800000) ? -0.7071f : 0.7071f) + 0.2f * ((float)((r >> 12) & 1023) / 1023.f - 0.5f);
            sym[i] = dsp::complex_t{ re, im };
            const int a = im < 0, b = re < 0, q = (a << 1) | (a != b);
            static const uint8_t remap[4] = { 0, 1, 3, 2 };
            dibit[i] = remap[(q - prev + 4) & 3];
            prev = q;
        }
    }
    void pushTo(dsp::demod::DecisionTap& t, size_t from, size_t to) const {
        std::vector<uint8_t> b(2 * (to - from));
        for (size_t i = from; i < to; i++) { b[2 * (i - from)] = (uint8_t)(dibit[i] >> 1); b[2 * (i - from) + 1] = (uint8_t)(dibit[i] & 1); }
        t.push(b.data(), (int)b.size());
    }
};
}  // namespace

static inline int tap_alignment_selftest() {
    int fails = 0;
    int curB = 0;
    auto check = [&](bool ok, const char* what) { if (!ok) { std::printf("decision tap alignment FAILED (buffers of %d): %s\n", curB, what); fails++; } };
    // --- buffers of 180 symbols; buffer 3 lost, buffer 6 handed twice, buffers 9..11 never delivered (consumer stopped) ---
    for (int B : { 180, 8, 1 }) {
        const size_t nbuf = 16;
        curB = B;
        SynthStream st(nbuf * (size_t)B);
        auto tap = std::make_shared<dsp::demod::DecisionTap>();
        auto tap2 = std::make_shared<dsp::demod::DecisionTap>();
        dsp::DQPSKSymbolExtractor ex;
        dsp::BitUnpacker un;
        ex.attachTap(tap);
        un.attachTap(tap2);
        st.pushTo(*tap, 0, st.sym.size());
        st.pushTo(*tap2, 0, st.sym.size());
        tap->mark(256, 0.125f, true);
        tap->mark((long long)(nbuf - 1) * B, 0.5f, false);
        std::vector<uint8_t> dib((size_t)B), bits(2 * (size_t)B);
        std::vector<int> order;
        for (int k = 0; k < (int)nbuf; k++) {
            if (k == 3 || (k >= 9 && k <= 11)) continue;
            order.push_back(k);
            if (k == 6) order.push_back(k);
        }
        bool all = true, dupSeen = false;
        int lastStatusBad = 0;
        for (size_t j = 0; j < order.size(); j++) {
            const int k = order[j];
            const bool isDup = j > 0 && order[j - 1] == k;
            ex.process(B, st.sym.data() + (size_t)k * B, dib.data());
            un.process(B, dib.data(), bits.data());
            for (int i = 0; i < B; i++) {
                const uint8_t want = st.dibit[(size_t)k * B + i];
                const bool exempt = (isDup || (B < dsp::demod::DecisionTap::kMinMatch && j > 0 && order[j - 1] != k - 1)) && i == 0;
                if (!exempt && dib[i] != want) all = false;
                if (bits[2 * i] != (dib[i] >> 1) || bits[2 * i + 1] != (dib[i] & 1)) all = false;
            }
            if (isDup) dupSeen = ex.lastStatus() == TETRA_ERR_ARG;
            else if (B >= dsp::demod::DecisionTap::kMinMatch && ex.lastStatus() != TETRA_OK) lastStatusBad++;
        }
        check(all, "dibits / bits of every handed buffer");
        if (B >= dsp::demod::DecisionTap::kMinMatch) {
            check(dupSeen && ex.fallbacks() == 1, "a repeated buffer is sliced locally and flagged");
            check(lastStatusBad == 0, "every other buffer comes from the queue");
            check(ex.resyncs() == 2 && ex.skippedSymbols() == 4 * B, "two gaps found: 1 + 3 buffers skipped");
            check(ex.standarderr == 0.5f && !ex.sync, "statistic marks applied at their stream positions across the gaps");
            check(tap->queuedSymbols() == 0 && tap->consumedSymbols() == (long long)st.sym.size(), "queue drained in step with the stream");
        } else if (B == 8) {
            // buffers shorter than kMinMatch realign once enough locally sliced dibits have accumulated
            check(ex.resyncs() >= 1 && tap->queuedSymbols() == 0, "short buffers realign on the accumulated run");
        }
    }
    // --- the consumer sleeps while the queue overflows: the oldest decisions are dropped, the consumer realigns on what it is handed ---
    {
        const int B = 4096;
        curB = B;
        const size_t total = dsp::demod::DecisionTap::kMaxQueuedSymbols + 5 * (size_t)B;
        SynthStream st(total, 777);
        auto tap = std::make_shared<dsp::demod::DecisionTap>();
        dsp::DQPSKSymbolExtractor ex;
        ex.attachTap(tap);
        std::vector<uint8_t> dib((size_t)B);
        st.pushTo(*tap, 0, (size_t)B);
        ex.process(B, st.sym.data(), dib.data());                       // buffer 0 consumed normally
        for (size_t pos = (size_t)B; pos < total; pos += 65536) st.pushTo(*tap, pos, pos + 65536 < total ? pos + 65536 : total);
        check(tap->droppedSymbols() == (long long)(4 * B), "overflow drops the oldest decisions");
        // SDR++'s stream held buffer 1 back all the while (its decisions are gone): sliced locally; then the newest buffer arrives
        ex.process(B, st.sym.data() + (size_t)B, dib.data());
        bool ok1 = ex.lastStatus() == TETRA_ERR_ARG;
        for (int i = 1; i < B; i++) ok1 = ok1 && dib[i] == st.dibit[(size_t)B + i];
        ex.process(B, st.sym.data() + total - (size_t)B, dib.data());
        bool ok2 = ex.lastStatus() == TETRA_OK && ex.resyncs() == 1;
        for (int i = 1; i < B; i++) ok2 = ok2 && dib[i] == st.dibit[total - (size_t)B + i];
        check(ok1 && ok2 && tap->queuedSymbols() == 0, "consumer realigns after the queue overflowed");
    }
    // --- PI4DQPSK::init() starts a new stream: open taps are emptied and their counters restart (no GPU needed: a failed create leaves the block uninitialised) ---
    std::printf("decision tap alignment %s\n", fails ? "FAILED" : "ok");
    return fails;
}

