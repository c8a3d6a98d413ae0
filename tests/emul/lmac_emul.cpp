// Host build of the lower-MAC decoder's lane-level code (sdrpp-tetra-demodulator_amd/csrc/lmac_core.hpp), one block at
// a time with plain arrays behind the accessors the kernel puts on LDS.  Test infrastructure: lets the CPU suite check
// the exact kernel source against the reference-built primitives (oracle/_ref) without a GPU.
#define TETRA_HOST_EMUL 1
#include <cstdint>
#include <cstring>

#include "../../sdrpp-tetra-demodulator_amd/csrc/lmac_core.hpp"

using namespace tetra_lmac;

extern "C" int lmac_emul_decode(int type345, int type2, int type1, int a, const uint8_t* type5, int n_blocks, int in_stride,
                                const uint32_t* scramb_init, uint8_t* out, int out_stride, int32_t* crc_ok) {
    if (type345 > kMaxType345 || type2 > kMaxType2 || (type345 & 3) || (type2 & 15) || (in_stride & 3)) return -1;
    for (int blk = 0; blk < n_blocks; ++blk) {
        const uint8_t* row = type5 + (size_t)blk * in_stride;
        uint32_t cls[(kMaxType345 + 15) / 16];
        uint16_t dec[kMaxType2 + kFlush];
        uint16_t outw[kMaxType2 / 16];
        uint32_t lfsr = scramb_init[blk];
        for (int c0 = 0; c0 < type345 / 4; c0 += 16)      // the kernel's 64-bit staging chunks
            lfsr = descramble_chunk(type345 - 4 * c0, lfsr,
                                    [&](int d) { uint32_t v; std::memcpy(&v, row + 4 * (c0 + d), 4); return v; },
                                    [&](int w, uint32_t word) { cls[c0 / 4 + w] = word; });
        viterbi_forward(type2, type345, a,
                        [&](int idx) { return (int)(cls[idx >> 4] << (30 - 2 * (idx & 15))) >> 30; },
                        [&](int t, uint32_t mask) { dec[t] = (uint16_t)mask; });
        viterbi_traceback(type2, [&](int t) { return (uint32_t)dec[t]; }, [&](int h, uint32_t half) { outw[h] = (uint16_t)half; });
        crc_ok[blk] = crc16_bits(type1 + 16, [&](int h) { return (uint32_t)outw[h]; }) == kCrcOk;
        for (int t4 = 0; t4 < type2 / 4; ++t4) {
            const uint32_t v = spread4((outw[t4 >> 2] >> (4 * (t4 & 3))) & 0xfu);
            std::memcpy(out + (size_t)blk * out_stride + 4 * t4, &v, 4);
        }
    }
    return 0;
}
