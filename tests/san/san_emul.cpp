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
// the lower-MAC decoder's lane code (lmac_core.hpp through tests/emul/lmac_emul.cpp): byte rows (both front ends) and straight from packed frames
extern "C" int lmac_emul_decode_route(int type345, int type2, int type1, int a, const uint8_t* type5, int n_blocks, int in_stride,
                                      const uint32_t* scramb_init, uint8_t* out, int out_stride, int32_t* crc_ok, int route, int32_t* fast_rows);
extern "C" int lmac_emul_decode_frames(int tpsap, int blk_num, const uint32_t* frames, const int32_t* frame_type, const int32_t* row_frame,
                                       int n_rows, const uint32_t* frame_scramb, uint8_t* out, int out_stride, int32_t* crc_ok);
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
    // the decoder: exact-size rows / frames / outputs; plain bits (packed front end), arbitrary bytes (byte route), frames of every type
    long long blocks = 0;
    {
        const int prm[5][4] = { { 120, 80, 60, 11 }, { 216, 144, 124, 101 }, { 168, 112, 92, 13 }, { 432, 288, 268, 103 }, { 216, 144, 124, 101 } };
        for (int k = 0; k < 5; k++)
            for (int route = 0; route < 2; route++)
                for (int mode = 0; mode < 2; mode++) {
                    const int n = 37, stride = prm[k][0];
                    std::vector<uint8_t> rows((size_t)n * stride), out((size_t)n * prm[k][1]);
                    std::vector<uint32_t> code((size_t)n);
                    std::vector<int32_t> ok((size_t)n);
                    for (auto& v : rows) v = mode ? (uint8_t)(lcg(seed) & 0xffu) : (uint8_t)(lcg(seed) & 1u);
                    for (auto& v : code) v = lcg(seed) * 2654435761u;
                    int32_t fast = -1;
                    if (lmac_emul_decode_route(prm[k][0], prm[k][1], prm[k][2], prm[k][3], rows.data(), n, stride, code.data(), out.data(), prm[k][1],
                                               ok.data(), route, &fast) != 0) return 6;
                    blocks += n;
                }
        const int fk[6][3] = { { 0, 1, 80 }, { 1, 2, 144 }, { 2, 1, 144 }, { 2, 2, 144 }, { 5, 0, 288 }, { 3, 0, 32 } };
        const int nfr = 200;
        std::vector<uint32_t> fr((size_t)nfr * 16), code((size_t)nfr);
        std::vector<int32_t> ft((size_t)nfr), list((size_t)nfr);
        for (auto& v : fr) v = lcg(seed) * 2654435761u;
        for (auto& v : code) v = lcg(seed) * 2246822519u;
        for (int i = 0; i < nfr; i++) { ft[i] = (int32_t)(lcg(seed) % 7u) - 2; list[i] = (int32_t)(lcg(seed) % (unsigned)nfr); }
        for (int k = 0; k < 6; k++) {
            std::vector<uint8_t> out((size_t)nfr * fk[k][2]);
            std::vector<int32_t> ok((size_t)nfr);
            if (lmac_emul_decode_frames(fk[k][0], fk[k][1], fr.data(), ft.data(), list.data(), nfr, fk[k][0] == 0 ? nullptr : code.data(), out.data(), fk[k][2],
                                        ok.data()) != 0) return 7;
            blocks += nfr;
        }
    }
    std::printf("san_emul: ok (%lld demultiplexer launches, %lld decoder blocks)\n", launches, blocks);
    return 0;
}
