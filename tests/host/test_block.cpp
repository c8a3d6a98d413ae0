// test_block.cpp -- exercises the dsp::block mirror (host/pi4dqpsk_gpu.h) the way the reference plugin drives its
// blocks: a source thread swaps IQ chunks into a stream, PI4DQPSK's worker thread run()s, a sink reads `out`.
// Usage: test_block <iq.f32 (interleaved re,im)> <chunk> <out_symbols.f32> <out_bits.u8>   (chunk must be >= 2 so
// that every chunk yields at least one symbol: like the reference, run() only swaps when symbols were produced)
// Without arguments: only constructs the classes (compile/link check; needs no GPU).
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../sdrpp-tetra-demodulator_amd/host/pi4dqpsk_gpu.h"

int main(int argc, char** argv) {
    tetra_demod_config_t cfg;
    tetra_demod_default_config(&cfg);
    if (argc < 5) {
        dsp::demod::PI4DQPSKBank bank;
        std::printf("abi %d, devices %d, default taps %d, bank channels %d\n", tetra_demod_abi_version(),
                    tetra_demod_device_count(), cfg.rrc_tap_count, bank.channels());
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
