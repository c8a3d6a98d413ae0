// constellation_core.hpp -- the plugin's constellation tap, per channel, as k_constellation (tetra_demod.hip) runs it after a launch.
//
// The reference regroups PI4DQPSK's symbol stream with dsp::buffer::Reshaper<complex_t>(keep 1024, skip 0) and copies every block it
// delivers into the GUI's diagram buffer (src/main.cpp:85-89, :376-383): on screen is the last complete block of 1024 consecutive
// symbols, blocks counted from the start of the stream.  Only the LAST block a call completes can be seen afterwards, so a call
// assembles that one (its first `fill` symbols from the carried partial block when the call completes exactly one) and keeps the
// symbols behind the last boundary as the new partial block.
//
// Compiled twice like the other *_core headers: by hipcc into the kernel and by g++ into tests/emul, where the two phases run for
// every thread index in turn (phase 1 for all, then phase 2 for all = the kernel's barrier) against a plain regrouping of the
// concatenated stream.  Index arithmetic and copies only.
#pragma once

#if defined(__HIPCC__) && !defined(TETRA_HOST_EMUL)
#define CD_FN __device__ __forceinline__
#else
#define CD_FN static inline
#endif

namespace tetra_cd {

constexpr int kSyms = 1024;        // Reshaper keep (main.cpp:88); = TETRA_CONSTELLATION_SYMBOLS

struct Plan {
    int nb;        // blocks this call completes
    int r;         // symbols behind the last boundary (the partial block after the call)
    int start;     // index, in this call's symbols, of the first symbol of the last completed block (< 0: its head is carried)
};

CD_FN Plan plan(int fill, int n) {
    const int total = fill + n;
    Plan p;
    p.nb = total / kSyms;
    p.r = total % kSyms;
    p.start = (p.nb - 1) * kSyms - fill;
    return p;
}

// Phase 1, thread tid of nthr: the last completed block.  start < 0 only when nb == 1, and then block index i < fill is partial
// index i.  Nothing to do when the call completes no block.
template <class Z>
CD_FN void assemble_block(const Plan& p, int tid, int nthr, const Z* z, const Z* part, Z* blk) {
    if (p.nb <= 0) return;
    for (int i = tid; i < kSyms; i += nthr) {
        const int j = p.start + i;
        blk[i] = j < 0 ? part[i] : z[j];
    }
}

// Phase 2 (after a barrier: phase 1 reads `part`), thread tid of nthr: the new partial block.
template <class Z>
CD_FN void carry_partial(const Plan& p, int fill, int n, int tid, int nthr, const Z* z, Z* part) {
    if (p.nb > 0) {
        for (int i = tid; i < p.r; i += nthr) part[i] = z[n - p.r + i];
    } else {
        for (int i = tid; i < n; i += nthr) part[fill + i] = z[i];
    }
}

}  // namespace tetra_cd
