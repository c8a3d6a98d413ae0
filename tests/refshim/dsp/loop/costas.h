// Stand-in for SDR++ core's dsp/loop/costas.h + pll.h: only the loop::PLL base the reference derives from (Appendix A).
#pragma once
#include <dsp/loop/phase_control_loop.h>

namespace dsp {
    namespace loop {
        class PLL : public Processor<complex_t, complex_t> {
            using base_type = Processor<complex_t, complex_t>;
        public:
            PLL() {}
            void init(stream<complex_t>* in, double bandwidth, double initPhase = 0.0, double initFreq = 0.0, double minFreq = -FL_M_PI, double maxFreq = FL_M_PI) {
                _initPhase = initPhase;
                _initFreq = initFreq;
                float alpha, beta;
                PhaseControlLoop<float>::criticallyDamped(bandwidth, alpha, beta);
                pcl.init(alpha, beta, initPhase, -FL_M_PI, FL_M_PI, initFreq, minFreq, maxFreq);
                base_type::init(in);
            }
            void setBandwidth(double bandwidth) {
                float alpha, beta;
                PhaseControlLoop<float>::criticallyDamped(bandwidth, alpha, beta);
                pcl.setCoefficients(alpha, beta);
            }
            void reset() { pcl.phase = _initPhase; pcl.freq = _initFreq; }
            virtual int process(int count, complex_t* in, complex_t* out) {
                for (int i = 0; i < count; i++) {
                    out[i] = math::phasor(pcl.phase);
                    pcl.advance((in[i] * complex_t{ out[i].re, -out[i].im }).phase());
                }
                return count;
            }
            int run() { return -1; }
        protected:
            PhaseControlLoop<float> pcl;
            float _initPhase = 0, _initFreq = 0;
        };
    }
}
