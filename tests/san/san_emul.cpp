// san_emul.cpp -- the host builds of the kernels' thread-level code that only moves bits and indices (demux_core.hpp through
// tests/emul/bsync_emul.cpp, constellation_core.hpp) under AddressSanitizer + UBSan: every buffer is a heap
// block of exactly the size the C ABI documents, so a thread of any launch that reads or writes one element too far is a report.
// Results are checked elsewhere (tests/test_burst_sync.py, tests/test_emul.py); this run is about addresses.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int bsync_emul_demux(const uint8_t* frames, int packed, const int32_t* frame_type, int n, int tpsap, int blk_num, uint8_t* rows,
                                int row_stride, int32_t* valid);
extern "C" int bsync_emul_demux_compact(const uint8_t* frames, int packed, const int32_t* frame_type, int n, int tpsap, int blk_num,
                                        uint8_t* rows, int row_stride, int32_t* row_frame, int32_t* n_rows);
#define TETRA_HOST_EMUL 1
#include "../../sdrpp-tetra-demodulator_amd/csrc/constellation_core.hpp"

// k_constellation's two phases for every thread index in turn (as tests/emul/emul.cpp's emul_constellation does)
static void emul_constellation(int n, const float* z, float* blk, float* part, int* fill, int* blocks, int nthr) {
    struct Z { float re, im; };
    const Z* zz = reinterpret_cast<const Z*>(z);
    Z* B = reinterpret_cast<Z*>(blk);
    Z* P = reinterpret_cast<Z*>(part);
    const int f0 = *fill;
    const tetra_cd::Plan p = tetra_cd::plan(f0, n);
    for (int tid = 0; tid < nthr; tid++) tetra_cd::assemble_block(p, tid, nthr, zz, P, B);
    for (int tid = nthr - 1; tid >= 0; tid--) tetra_cd::carry_partial(p, f0, n, tid, nthr, zz, P);
    *fill = p.r;
    *blocks += p.nb;
}

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

int main() {
    uint32_t seed = 12345;
    const int kinds[6][2] = { { 0, 1 }, { 1, 2 }, { 2, 1 }, { 2, 2 }, { 3, 0 }, { 5, 0 } };
    const int longest[6] = { 120, 216, 216, 216, 30, 432 };
    long long launches = 0;
    for (int n : { 1, 2, 63, 64, 65, 257, 1000 }) {
        // exact-size heap blocks: [n][512] bytes / [n][16] words / [n] types
        std::vector<uint8_t> frames((size_t)n * 512);
        std::vector<uint32_t> packed((size_t)n * 16, 0);
        std::vector<int32_t> types((size_t)n);
        for (int r = 0; r < n; r++) {
            for (int i = 0; i < 512; i++) frames[(size_t)r * 512 + i] = i < 510 ? (uint8_t)(lcg(seed) & 1u) : 0;
            for (int i = 0; i < 510; i++) packed[(size_t)r * 16 + (i >> 5)] |= (uint32_t)frames[(size_t)r * 512 + i] << (31 - (i & 31));
            const int pick[7] = { 0, 1, 2, 3, 4, -1, -2 };
            types[r] = pick[lcg(seed) % 7];
        }
        for (int k = 0; k < 6; k++)
            for (int pad : { 0, 2, 4, 8, 40, 100 }) {
                const int stride = ((longest[k] + 3) & ~3) + ((pad + 3) & ~3);
                for (int pk = 0; pk < 2; pk++) {
                    std::vector<uint8_t> rows((size_t)n * stride);
                    std::vector<int32_t> valid((size_t)n), idx((size_t)n);
                    int32_t cnt = -1;
                    const uint8_t* src = pk ? reinterpret_cast<const uint8_t*>(packed.data()) : frames.data();
                    if (bsync_emul_demux(src, pk, types.data(), n, kinds[k][0], kinds[k][1], rows.data(), stride, valid.data()) != 0) return 2;
                    if (bsync_emul_demux_compact(src, pk, types.data(), n, kinds[k][0], kinds[k][1], rows.data(), stride, idx.data(), &cnt) != 0) return 3;
                    if (cnt < 0 || cnt > n) return 4;
                    launches += 2;
                }
            }
    }
    // the constellation tap: exact-size symbol blocks per call, 1024-entry block / partial buffers
    for (int nthr : { 1, 64, 256 }) {
        std::vector<float> blk(2 * 1024), part(2 * 1024);
        int fill = 0, blocks = 0;
        long long total = 0;
        for (int call = 0; call < 200; call++) {
            const int n = (int)(lcg(seed) % 5000u) * (int)(lcg(seed) & 1u);      // half of the calls empty
            std::vector<float> z((size_t)2 * n);
            for (auto& v : z) v = (float)(lcg(seed) & 1023u);
            emul_constellation(n, z.data(), blk.data(), part.data(), &fill, &blocks, nthr);
            total += n;
            if (fill != (int)(total % 1024) || blocks != (int)(total / 1024)) return 5;
        }
    }
    std::printf("san_emul: ok (%lld demultiplexer launches)\n", launches);
    return 0;
}
