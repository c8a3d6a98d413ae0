// test_block.cpp -- exercises the dsp::block mirror (host/pi4dqpsk_gpu.h) the way the reference plugin drives its
// blocks: a source thread swaps IQ chunks into a stream, PI4DQPSK's worker thread run()s, a sink reads `out`.
// Usage: test_block <iq.f32 (interleaved re,im)> <chunk> <out_symbols.f32> <out_bits.u8>   (chunk must be >= 2 so
// that every chunk yields at least one symbol: like the reference, run() only swaps when symbols were produced)
// Without arguments: only constructs the classes (compile/link check; needs no GPU).
#include <hip/hip_runtime_api.h>      // only the "multibank-device" mode stages its own device buffers (the plugin side never needs HIP)

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <string>
#include <vector>

#include "../../sdrpp-tetra-demodulator_amd/host/pi4dqpsk_gpu.h"
#include "../../sdrpp-tetra-demodulator_amd/host/dqpsk_sym_extr_gpu.h"
#include "../../sdrpp-tetra-demodulator_amd/host/bit_unpacker_gpu.h"
#include "tap_selftest.h"

// test_block multibank|multibank-cs16|multibank-device <iq.f32 [C][n]> <C> <n> <calls> <out_bits.u8> <out_nbits.i32> [dev0 dev1 ...]
// PI4DQPSKMultiBank over the given devices (default: 0 0 = two shards, two host threads, two handles on ONE GPU), the
// stream cut into `calls` equal calls; writes every call's bit rows [calls][C][stride] and counts [calls][C].
//   multibank         process(): float IQ from page-locked host memory
//   multibank-cs16    processCS16(): the same samples as interleaved int16 (x * 32768 rounded; the caller compares with the
//                     oracle on the dequantised samples)
//   multibank-device  processDevice(): every shard's samples, bit rows and counts live on that shard's GPU
static int multibank_main(int argc, char** argv) {
    const bool cs16 = !std::strcmp(argv[1], "multibank-cs16"), resident = !std::strcmp(argv[1], "multibank-device");
    const int C = std::atoi(argv[3]), n = std::atoi(argv[4]), calls = std::atoi(argv[5]);
    std::vector<int> devs;
    for (int i = 8; i < argc; i++) devs.push_back(std::atoi(argv[i]));
    if (devs.empty()) devs = { 0, 0 };
    std::vector<float> iq((size_t)C * n * 2);
    FILE* f = std::fopen(argv[2], "rb");
    if (!f || std::fread(iq.data(), sizeof(float), iq.size(), f) != iq.size()) return 2;
    std::fclose(f);
    const int per = n / calls;
    tetra_demod_config_t cfg;
    tetra_demod_default_config(&cfg);
    cfg.n_channels = C;
    cfg.max_samples = per;
    cfg.flags |= TETRA_FLAG_CONSTELLATION;      // the plugin's constellation tap, kept per channel on each shard's GPU
    dsp::demod::PI4DQPSKMultiBank mb;
    int rc = mb.init(cfg, devs);
    if (rc != TETRA_OK) { std::fprintf(stderr, "init failed: %s\n", tetra_demod_strerror(rc)); return 3; }
    const int stride = mb.bitsStride(per);
    std::printf("stride %d\n", stride);
    std::vector<int16_t> q;
    // per-shard device buffers (multibank-device)
    std::vector<dsp::complex_t*> dIn((size_t)mb.shards(), nullptr);
    std::vector<uint8_t*> dBits((size_t)mb.shards(), nullptr);
    std::vector<int32_t*> dNb((size_t)mb.shards(), nullptr);
    if (resident)
        for (int g = 0; g < mb.shards(); g++) {
            int first, count, dev;
            mb.shardInfo(g, first, count, dev);
            if (hipSetDevice(dev) != hipSuccess) return 7;
            if (hipMalloc((void**)&dIn[g], sizeof(dsp::complex_t) * (size_t)count * per) != hipSuccess) return 7;
            if (hipMalloc((void**)&dBits[g], (size_t)count * stride) != hipSuccess) return 7;
            if (hipMalloc((void**)&dNb[g], sizeof(int32_t) * count) != hipSuccess) return 7;
        }
    // page-locked buffers: the shards' copies then really run side by side
    dsp::complex_t* in = (dsp::complex_t*)tetra_demod_host_alloc(sizeof(dsp::complex_t) * (size_t)C * per);
    uint8_t* bits = (uint8_t*)tetra_demod_host_alloc((size_t)C * stride);
    int32_t* nb = (int32_t*)tetra_demod_host_alloc(sizeof(int32_t) * C);
    if (!in || !bits || !nb) return 4;
    FILE* fb = std::fopen(argv[6], "wb");
    FILE* fn = std::fopen(argv[7], "wb");
    for (int k = 0; k < calls; k++) {
        for (int c = 0; c < C; c++)
            std::memcpy(in + (size_t)c * per, iq.data() + 2 * ((size_t)c * n + (size_t)k * per), sizeof(dsp::complex_t) * per);
        if (cs16) {
            q.resize((size_t)C * per * 2);
            const float* fp = reinterpret_cast<const float*>(in);
            for (size_t i = 0; i < q.size(); i++) {
                float v = std::nearbyint(fp[i] * 32768.0f);
                q[i] = (int16_t)(v > 32767.f ? 32767.f : v < -32768.f ? -32768.f : v);
            }
            rc = mb.processCS16(per, q.data(), bits, nb);
        } else if (resident) {
            for (int g = 0; g < mb.shards(); g++) {
                int first, count, dev;
                mb.shardInfo(g, first, count, dev);
                if (hipSetDevice(dev) != hipSuccess ||
                    hipMemcpy(dIn[g], in + (size_t)first * per, sizeof(dsp::complex_t) * (size_t)count * per, hipMemcpyHostToDevice) != hipSuccess) return 8;
            }
            rc = mb.processDevice(per, dIn.data(), dBits.data(), dNb.data());
            for (int g = 0; g < mb.shards() && rc == TETRA_OK; g++) {
                int first, count, dev;
                mb.shardInfo(g, first, count, dev);
                if (hipSetDevice(dev) != hipSuccess ||
                    hipMemcpy(bits + (size_t)first * stride, dBits[g], (size_t)count * stride, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(nb + first, dNb[g], sizeof(int32_t) * count, hipMemcpyDeviceToHost) != hipSuccess) return 8;
            }
        } else {
            rc = mb.process(per, in, bits, nb);
        }
        if (rc != TETRA_OK) { std::fprintf(stderr, "process failed: %s\n", tetra_demod_strerror(rc)); return 5; }
        std::fwrite(bits, 1, (size_t)C * stride, fb);
        std::fwrite(nb, sizeof(int32_t), C, fn);
        if (k == calls / 2 && mb.setParam(TETRA_PARAM_AGC_RATE, 0.02) != TETRA_OK) return 6;   // a setter reaches every shard
    }
    std::fclose(fb);
    std::fclose(fn);
    for (int g = 0; g < mb.shards(); g++) {
        int first, count, dev;
        mb.shardInfo(g, first, count, dev);
        std::printf("shard %d: channels [%d, %d) on device %d\n", g, first, first + count, dev);
    }
    {   // MultiBank::constellation: every channel's last complete 1024-symbol block -> <out_bits>.cd / .cdn for the caller; a range
        // that starts and ends inside shards must be the same rows
        std::vector<dsp::complex_t> blk((size_t)C * TETRA_CONSTELLATION_SYMBOLS), mid(blk.size());
        std::vector<int32_t> nblk((size_t)C), nmid((size_t)C);
        if (mb.constellation(0, C, blk.data(), nblk.data()) != TETRA_OK) return 10;
        if (C > 2) {
            if (mb.constellation(1, C - 2, mid.data(), nmid.data()) != TETRA_OK) return 10;
            if (std::memcmp(mid.data(), blk.data() + TETRA_CONSTELLATION_SYMBOLS, sizeof(dsp::complex_t) * (size_t)(C - 2) * TETRA_CONSTELLATION_SYMBOLS) ||
                std::memcmp(nmid.data(), nblk.data() + 1, sizeof(int32_t) * (size_t)(C - 2))) return 11;
        }
        if (mb.constellation(C, 1, blk.data(), nullptr) != TETRA_ERR_ARG || mb.constellation(-1, 1, nullptr, nullptr) != TETRA_ERR_ARG) return 12;
        const std::string base = argv[6];
        FILE* fc = std::fopen((base + ".cd").c_str(), "wb");
        FILE* fcn = std::fopen((base + ".cdn").c_str(), "wb");
        if (!fc || !fcn) return 13;
        std::fwrite(blk.data(), sizeof(dsp::complex_t), blk.size(), fc);
        std::fwrite(nblk.data(), sizeof(int32_t), nblk.size(), fcn);
        std::fclose(fc);
        std::fclose(fcn);
    }
    if (cfg.flags & TETRA_FLAG_QUALITY) {      // not set by this driver today; exercises the link of MultiBank::quality
        std::vector<float> e((size_t)C);
        std::vector<uint8_t> sy((size_t)C);
        if (mb.quality(e.data(), sy.data()) != TETRA_OK) return 9;
    }
    for (int g = 0; g < mb.shards(); g++) { (void)hipFree(dIn[g]); (void)hipFree(dBits[g]); (void)hipFree(dNb[g]); }
    tetra_demod_host_free(in); tetra_demod_host_free(bits); tetra_demod_host_free(nb);
    return 0;
}

// test_block chain3 <iq.f32> <chunk> <out_bits.u8> <out_dibits.u8> <out_symbols.f32>
// The plugin's three blocks as src/main.cpp:84-91 wires them, all three GPU-backed mirrors, each with its own worker thread:
//   source thread -> PI4DQPSK -> relay thread (stands where SDR++'s splitter is, src/main.cpp:85-87) -> DQPSKSymbolExtractor
//   -> BitUnpacker -> sink (this thread).  No DSP arithmetic on the host anywhere: the sink only stores what arrives.
// Prints "standarderr <float> sync <0|1>" = the extractor's public members after the last chunk.
static int chain3_main(int argc, char** argv) {
    (void)argc;
    tetra_demod_config_t cfg;
    tetra_demod_default_config(&cfg);
    FILE* f = std::fopen(argv[2], "rb");
    if (!f) return 2;
    std::vector<float> iq;
    float tmp[4096];
    size_t r;
    while ((r = std::fread(tmp, sizeof(float), 4096, f)) > 0) iq.insert(iq.end(), tmp, tmp + r);
    std::fclose(f);
    const int n = (int)(iq.size() / 2), chunk = std::atoi(argv[3]);
    // optional: the relay LOSES symbol chunk `drop` and hands chunk `dup` on twice (-1 = neither), and writes the symbol count of
    // every chunk it forwarded to <counts file> -- what a splitter that is re-bound mid-stream does to the blocks behind it
    const int drop = argc > 7 ? std::atoi(argv[7]) : -1, dup = argc > 8 ? std::atoi(argv[8]) : -1;
    const char* countsPath = argc > 9 ? argv[9] : nullptr;

    dsp::stream<dsp::complex_t> src, demodStream;
    dsp::demod::PI4DQPSK mainDemodulator;
    dsp::DQPSKSymbolExtractor symbolExtractor;
    dsp::BitUnpacker bitsUnpacker;
    mainDemodulator.init(&src, cfg.symbolrate, cfg.samplerate, cfg.rrc_tap_count, cfg.rrc_beta, cfg.agc_rate, cfg.costas_bandwidth,
                         cfg.fll_bandwidth, cfg.omega_gain, cfg.mu_gain, cfg.omega_rel_limit);
    if (mainDemodulator.lastStatus() != TETRA_OK) { std::fprintf(stderr, "init failed: %s\n", tetra_demod_strerror(mainDemodulator.lastStatus())); return 3; }
    symbolExtractor.init(&demodStream);                // src/main.cpp:90
    bitsUnpacker.init(&symbolExtractor.out);           // src/main.cpp:91
    symbolExtractor.attach(&mainDemodulator);          // the one added line per block of the GPU drop-in
    bitsUnpacker.attach(&mainDemodulator);
    mainDemodulator.start();
    symbolExtractor.start();
    bitsUnpacker.start();
    const int chunks = (n + chunk - 1) / chunk;
    std::vector<float> syms;
    std::thread feeder([&] {
        for (int pos = 0; pos < n; pos += chunk) {
            const int c = n - pos < chunk ? n - pos : chunk;
            std::memcpy(src.writeBuf, iq.data() + 2 * (size_t)pos, sizeof(float) * 2 * (size_t)c);
            if (!src.swap(c)) return;
        }
    });
    std::vector<int> forwarded;                       // (chunk index, symbols) pairs in the order the extractor saw them
    std::thread relay([&] {                            // the splitter's place: every symbol chunk goes on unchanged
        std::vector<dsp::complex_t> keep;
        for (int k = 0; k < chunks; k++) {
            const int c = mainDemodulator.out.read();
            if (c < 0) return;
            const float* p = reinterpret_cast<const float*>(mainDemodulator.out.readBuf);
            syms.insert(syms.end(), p, p + 2 * (size_t)c);
            keep.assign(mainDemodulator.out.readBuf, mainDemodulator.out.readBuf + c);
            mainDemodulator.out.flush();
            for (int rep = 0; rep < (k == drop ? 0 : k == dup ? 2 : 1); rep++) {
                std::memcpy(demodStream.writeBuf, keep.data(), sizeof(dsp::complex_t) * (size_t)c);
                forwarded.push_back(k);
                forwarded.push_back(c);
                if (!demodStream.swap(c)) return;
            }
        }
    });
    std::vector<uint8_t> bits;
    const int sinkChunks = chunks - (drop >= 0 && drop < chunks ? 1 : 0) + (dup >= 0 && dup < chunks && dup != drop ? 1 : 0);
    for (int got = 0; got < sinkChunks; got++) {
        const int c = bitsUnpacker.out.read();
        if (c < 0) break;
        bits.insert(bits.end(), bitsUnpacker.out.readBuf, bitsUnpacker.out.readBuf + c);
        bitsUnpacker.out.flush();
    }
    feeder.join();
    relay.join();
    std::printf("standarderr %.9g sync %d status %d %d %d\n", (double)symbolExtractor.standarderr, symbolExtractor.sync ? 1 : 0,
                mainDemodulator.lastStatus(), symbolExtractor.lastStatus(), bitsUnpacker.lastStatus());
    std::printf("extractor resyncs %lld skipped %lld fallbacks %lld unpacker resyncs %lld fallbacks %lld\n", symbolExtractor.resyncs(),
                symbolExtractor.skippedSymbols(), symbolExtractor.fallbacks(), bitsUnpacker.resyncs(), bitsUnpacker.fallbacks());
    if (countsPath) {
        FILE* fc = std::fopen(countsPath, "wb");
        std::fwrite(forwarded.data(), sizeof(int), forwarded.size(), fc);
        std::fclose(fc);
    }
    bitsUnpacker.stop();
    symbolExtractor.stop();
    mainDemodulator.stop();
    FILE* fb = std::fopen(argv[4], "wb");
    std::fwrite(bits.data(), 1, bits.size(), fb);
    std::fclose(fb);
    FILE* fs = std::fopen(argv[6], "wb");
    std::fwrite(syms.data(), sizeof(float), syms.size(), fs);
    std::fclose(fs);
    // the extractor's own process() signature, called directly the way a unit test of the reference's block would: dibits of a
    // second, short stream through the same demodulator
    if (drop < 0 && dup < 0) {
        std::vector<dsp::complex_t> out((size_t)chunk);
        std::vector<uint8_t> dib((size_t)chunk);
        const int ns = mainDemodulator.process(chunk, reinterpret_cast<const dsp::complex_t*>(iq.data()), out.data());
        if (ns < 0) return 4;
        uint8_t ub[4096];
        const int nd = symbolExtractor.process(ns, out.data(), dib.data());
        const int nbit = bitsUnpacker.process(nd, dib.data(), ub);
        if (nd != ns || nbit != 2 * ns || symbolExtractor.lastStatus() != TETRA_OK || bitsUnpacker.lastStatus() != TETRA_OK) return 5;
        FILE* fd = std::fopen(argv[5], "wb");
        std::fwrite(dib.data(), 1, (size_t)nd, fd);
        std::fwrite(ub, 1, (size_t)nbit, fd);
        std::fclose(fd);
        // a stream that does NOT come from the attached demodulator is flagged, not silently sliced
        if (symbolExtractor.process(3, out.data(), dib.data()) != 3 || symbolExtractor.lastStatus() != TETRA_ERR_ARG) return 6;
    }
    std::printf("symbols %zu bits %zu\n", syms.size() / 2, bits.size());
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 8 && !std::strncmp(argv[1], "multibank", 9)) return multibank_main(argc, argv);
    if (argc >= 7 && !std::strcmp(argv[1], "chain3")) return chain3_main(argc, argv);
    tetra_demod_config_t cfg;
    tetra_demod_default_config(&cfg);
    if (argc < 5) {
        dsp::demod::PI4DQPSKBank bank;
        std::printf("abi %d, devices %d, default taps %d, bank channels %d\n", tetra_demod_abi_version(),
                    tetra_demod_device_count(), cfg.rrc_tap_count, bank.channels());
        // the decision FIFO between the demodulator mirror and the extractor / unpacker mirrors (host logic only, no GPU):
        // stream order, dibit packing, statistic marks applied when the consumer has passed their symbol position, underflow
        {
            dsp::demod::DecisionTap tap;
            const uint8_t b0[8] = { 0, 1, 1, 0, 1, 1, 0, 0 };          // dibits 1, 2, 3, 0
            tap.push(b0, 8);
            tap.mark(3, 0.25f, true);                                  // published when 3 symbols have been consumed
            uint8_t dib[8], bits[16];
            float err = -1.f;
            bool sync = false;
            int ok = tap.pop(2, dib, bits, &err, &sync) == 2 && dib[0] == 1 && dib[1] == 2 && bits[0] == 0 && bits[1] == 1 && bits[2] == 1 &&
                     bits[3] == 0 && err == -1.f && !sync;             // two symbols consumed: the mark at 3 is not yet due
            ok = ok && tap.pop(1, dib, nullptr, &err, &sync) == 1 && dib[0] == 3 && err == 0.25f && sync && tap.consumedSymbols() == 3;
            ok = ok && tap.pop(5, dib, bits, &err, &sync) == 1 && dib[0] == 0;          // only one symbol left: the caller sees the shortfall
            tap.push(b0, 4);
            tap.clear();
            ok = ok && tap.pop(1, dib, bits, nullptr, nullptr) == 0;
            std::printf("decision tap %s\n", ok ? "ok" : "FAILED");
            if (!ok) return 1;
        }
        if (tap_alignment_selftest() != 0) return 1;
        return 0;
    }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<float> iq;
    float tmp[4096];
    size_t r;
    while ((r = std::fread(tmp, sizeof(float), 4096, f)) > 0) iq.insert(iq.end(), tmp, tmp + r);
    std::fclose(f);
    const int n = (int)(iq.size() / 2), chunk = std::atoi(argv[2]);

    dsp::stream<dsp::complex_t> src;
    dsp::demod::PI4DQPSK dem;
    dem.init(&src, cfg.symbolrate, cfg.samplerate, cfg.rrc_tap_count, cfg.rrc_beta, cfg.agc_rate, cfg.costas_bandwidth,
             cfg.fll_bandwidth, cfg.omega_gain, cfg.mu_gain, cfg.omega_rel_limit);
    if (dem.lastStatus() != TETRA_OK) { std::fprintf(stderr, "init failed: %s\n", tetra_demod_strerror(dem.lastStatus())); return 3; }
    dem.start();
    std::vector<float> syms;
    std::vector<uint8_t> bits;
    std::thread feeder([&] {
        for (int pos = 0; pos < n; pos += chunk) {
            const int c = n - pos < chunk ? n - pos : chunk;
            std::memcpy(src.writeBuf, iq.data() + 2 * (size_t)pos, sizeof(float) * 2 * (size_t)c);
            if (!src.swap(c)) return;
        }
    });
    // sink = what the plugin's own DQPSKSymbolExtractor + BitUnpacker do with `out` (sign tests only,
    // src/dsp/dqpsk_sym_extr.cpp:6-7,32-52 and src/dsp/bit_unpacker.cpp:6-7), restated here for the test
    int chunks = (n + chunk - 1) / chunk, got = 0, prev = 0;
    while (got < chunks) {
        int c = dem.out.read();
        if (c < 0) break;
        const float* p = reinterpret_cast<const float*>(dem.out.readBuf);
        syms.insert(syms.end(), p, p + 2 * (size_t)c);
        for (int i = 0; i < c; i++) {
            const int a = p[2 * i + 1] < 0, b = p[2 * i] < 0;
            const int sym = (a << 1) | (a != b);
            const int pd = (sym - prev + 4) % 4;
            static const int remap[4] = { 0, 1, 3, 2 };
            prev = sym;
            bits.push_back((uint8_t)((remap[pd] >> 1) & 1));
            bits.push_back((uint8_t)(remap[pd] & 1));
        }
        dem.out.flush();
        got++;
    }
    feeder.join();
    dem.stop();
    FILE* fs = std::fopen(argv[3], "wb");
    std::fwrite(syms.data(), sizeof(float), syms.size(), fs);
    std::fclose(fs);
    FILE* fb = std::fopen(argv[4], "wb");
    std::fwrite(bits.data(), 1, bits.size(), fb);
    std::fclose(fb);
    std::printf("symbols %zu bits %zu\n", syms.size() / 2, bits.size());
    return 0;
}
