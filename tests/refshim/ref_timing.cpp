// Per-stage CPU timing of the REFERENCE's own src/dsp objects (compiled where they lie against the stand-in core headers of this
// directory; container only) -- the probe of SURVEY.md Appendix B.6 made reproducible: profiles/time_refshim_cpu.py builds and runs
// it.  Chain and call order = PI4DQPSK::process (pi4dqpsk.cpp:132-140) -> DQPSKSymbolExtractor -> BitUnpacker (main.cpp:84-91);
// the sub-blocks are protected members, reached with g++ -fno-access-control like ref_driver.cpp does.  VOLK's dot products are
// the scalar stand-ins of dsp/processor.h: a real SDR++ build runs VOLK's SIMD kernels there (FLL, RRC, interpolator), so the
// FIR-heavy stages are slower here than in the field -- said wherever the number is quoted.
#include <dsp/pi4dqpsk.h>
#include <dsp/dqpsk_sym_extr.h>
#include <dsp/bit_unpacker.h>

#include <chrono>

extern "C" {

// iq: count interleaved samples, processed `reps` times in calls of `chunk` samples (state carried; the stream repeats).
// ns_per_sample[0..6] = AGC, FLL, RRC, COMPLEX_FD, COSTAS, SYM_EXTR, BIT_UNPACK; [7] = the whole chain through
// PI4DQPSK::process + extractor + unpacker on fresh objects (what a user of the plugin pays).  Returns symbols of the last rep.
int ref_time_stages(int count, const float* iq, int chunk, int reps, double symbolrate, double samplerate, int rrc_taps, double rrc_beta,
                    double agc_rate, double costas_bw, double fll_bw, double omega_gain, double mu_gain, double omega_rel_limit,
                    double* ns_per_sample) {
    using clk = std::chrono::steady_clock;
    std::vector<dsp::complex_t> buf((size_t)chunk + 16);
    std::vector<uint8_t> dib((size_t)chunk + 16), bits(2 * (size_t)chunk + 32);
    double acc[7] = { 0 };
    int nsym = 0;
    {
        dsp::demod::PI4DQPSK d;
        dsp::DQPSKSymbolExtractor ex;
        dsp::BitUnpacker un;
        d.init(nullptr, symbolrate, samplerate, rrc_taps, rrc_beta, agc_rate, costas_bw, fll_bw, omega_gain, mu_gain, omega_rel_limit);
        ex.init(nullptr);
        un.init(nullptr);
        for (int r = 0; r < reps; r++) {
            nsym = 0;
            for (int pos = 0; pos < count; pos += chunk) {
                const int c = count - pos < chunk ? count - pos : chunk;
                const dsp::complex_t* in = (const dsp::complex_t*)iq + pos;
                auto t0 = clk::now();
                int ret = d.agc.process(c, (dsp::complex_t*)in, buf.data());
                auto t1 = clk::now();
                ret = d.fll.process(ret, buf.data(), buf.data());
                auto t2 = clk::now();
                ret = d.rrc.process(ret, buf.data(), buf.data());
                auto t3 = clk::now();
                ret = d.recov.process(ret, buf.data(), buf.data());
                auto t4 = clk::now();
                ret = d.costas.process(ret, buf.data(), buf.data());
                auto t5 = clk::now();
                const int nd = ex.process(ret, buf.data(), dib.data());
                auto t6 = clk::now();
                un.process(nd, dib.data(), bits.data());
                auto t7 = clk::now();
                const clk::time_point tp[8] = { t0, t1, t2, t3, t4, t5, t6, t7 };
                for (int s = 0; s < 7; s++) acc[s] += std::chrono::duration<double, std::nano>(tp[s + 1] - tp[s]).count();
                nsym += ret;
            }
        }
    }
    for (int s = 0; s < 7; s++) ns_per_sample[s] = acc[s] / ((double)count * reps);
    {
        dsp::demod::PI4DQPSK d;
        dsp::DQPSKSymbolExtractor ex;
        dsp::BitUnpacker un;
        d.init(nullptr, symbolrate, samplerate, rrc_taps, rrc_beta, agc_rate, costas_bw, fll_bw, omega_gain, mu_gain, omega_rel_limit);
        ex.init(nullptr);
        un.init(nullptr);
        auto t0 = clk::now();
        for (int r = 0; r < reps; r++)
            for (int pos = 0; pos < count; pos += chunk) {
                const int c = count - pos < chunk ? count - pos : chunk;
                const int ns = d.process(c, (const dsp::complex_t*)iq + pos, buf.data());
                const int nd = ex.process(ns, buf.data(), dib.data());
                un.process(nd, dib.data(), bits.data());
            }
        ns_per_sample[7] = std::chrono::duration<double, std::nano>(clk::now() - t0).count() / ((double)count * reps);
    }
    return nsym;
}
}
