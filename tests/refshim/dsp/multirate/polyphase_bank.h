// Stand-in for SDR++ core's dsp/multirate/polyphase_bank.h: phase index reversed, zero padded.
#pragma once
#include <dsp/processor.h>

namespace dsp {
    namespace multirate {
        template <class T> struct PolyphaseBank {
            int phaseCount = 0;
            int tapsPerPhase = 0;
            T** phases = nullptr;
        };
        template <class T> inline PolyphaseBank<T> buildPolyphaseBank(int phaseCount, tap<T>& taps) {
            PolyphaseBank<T> pb;
            pb.phaseCount = phaseCount;
            pb.tapsPerPhase = (taps.size + phaseCount - 1) / phaseCount;
            pb.phases = buffer::alloc<T*>(phaseCount);
            for (int p = 0; p < phaseCount; p++) { pb.phases[p] = buffer::alloc<T>(pb.tapsPerPhase); }
            for (int i = 0; i < taps.size; i++) { pb.phases[(phaseCount - 1) - (i % phaseCount)][i / phaseCount] = taps.taps[i]; }
            return pb;
        }
        template <class T> inline void freePolyphaseBank(PolyphaseBank<T>& bank) {
            if (!bank.phases) { return; }
            for (int p = 0; p < bank.phaseCount; p++) { buffer::free(bank.phases[p]); }
            buffer::free(bank.phases);
            bank.phases = nullptr;
            bank.phaseCount = 0;
        }
    }
}
