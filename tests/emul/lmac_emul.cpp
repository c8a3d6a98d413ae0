// Host build of the lower-MAC decoder's lane-level code (sdrpp-tetra-demodulator_amd/csrc/lmac_core.hpp), one block at
// a time with plain arrays behind the accessors the kernel puts on LDS.  Test infrastructure: lets the CPU suite check
// the exact kernel source against the reference-built primitives (oracle/_ref) without a GPU.
#define TETRA_HOST_EMUL 1
#include <cstdint>
#include <cstring>

#include "../../sdrpp-tetra-demodulator_amd/csrc/lmac_core.hpp"

using namespace tetra_lmac;

namespace {
const uint32_t* seq_table() {
    static uint32_t* tab = nullptr;
    if (!tab) {
        tab = new uint32_t[(size_t)4 * 256 * kSeqStride];
        scramb_sequence_table(tab);
    }
    return tab;
}
}  // namespace

// route: 0 = like the kernel (rows of plain bits take the packed route, any other row the byte route), 1 = byte route always
extern "C" int lmac_emul_decode_route(int type345, int type2, int type1, int a, const uint8_t* type5, int n_blocks, int in_stride,
                                      const uint32_t* scramb_init, uint8_t* out, int out_stride, int32_t* crc_ok, int route, int32_t* fast_rows) {
    if (type345 > kMaxType345 || type2 > kMaxType2 || (type345 & 7) || (type2 & 15) || (in_stride & 3)) return -1;
    static const CrcInvTable crci = make_crc_inv_table();
    (void)type1;                  // n2 = type1 + 16 + 4 for every coded kind: the traceback derives the CRC span from n2
    const uint32_t* tab = seq_table();
    int fast = 0;
    for (int blk = 0; blk < n_blocks; ++blk) {
        const uint8_t* row = type5 + (size_t)blk * in_stride;
        uint32_t cls[(kMaxType345 + 15) / 16 + 1];
        uint32_t dec[(kMaxType2 + kFlush) / 2];
        uint16_t outw[kMaxType2 / 16];
        uint32_t xb[kSeqWords];
        const uint32_t dirty = pack_row_bits(type345, [&](int i) { U2 d; std::memcpy(&d, row + 8 * i, 8); return d; }, xb);
        int pos = a;
        const bool bits_route = route == 0 && !dirty;
        if (bits_route) {
            ++fast;
            descramble_bits(type345, scramb_init[blk], xb, [&](int t, uint32_t byte, int w) { return tab[((size_t)t * 256 + byte) * kSeqStride + w]; },
                            [&](int w, uint32_t word) { cls[w] = word; });
        } else {
            uint32_t lfsr = scramb_init[blk];
            for (int c0 = 0; c0 < type345 / 4; c0 += 16)      // the kernel's 64-bit staging chunks
                lfsr = descramble_chunk(type345 - 4 * c0, lfsr,
                                        [&](int d) { uint32_t v; std::memcpy(&v, row + 4 * (c0 + d), 4); return v; },
                                        [&](int w, uint32_t word) { cls[c0 / 4 + w] = word; });
        }
        auto fetch = [&] {                                    // as decode_core in tetra_lmac.hip
            Raw3 r;
            for (int k = 0; k < 3; ++k) {
                const int p = interleave_next(pos, a, type345);
                r.w[k] = cls[bits_route ? p >> 5 : p >> 4];
                r.at[k] = bits_route ? 31u - (uint32_t)(p & 31) : (uint32_t)(30 - 2 * (p & 15));
            }
            return r;
        };
        auto dec_st = [&](int u, uint32_t word) { dec[u] = word; };
        if (bits_route)
            viterbi_forward(type2, fetch,
                            [&](const Raw3& r) { return bm_from_masks(bfe_mask(r.w[0], r.at[0]), bfe_mask(r.w[1], r.at[1]), bfe_mask(r.w[2], r.at[2])); }, dec_st);
        else
            viterbi_forward(type2, fetch,
                            [&](const Raw3& r) { return bm_from_classes((int)(r.w[0] << r.at[0]) >> 30, (int)(r.w[1] << r.at[1]) >> 30, (int)(r.w[2] << r.at[2]) >> 30); },
                            dec_st);
        crc_ok[blk] = viterbi_traceback(type2, [&](int u) { return dec[u]; }, [&](int h, uint32_t half) { outw[h] = (uint16_t)half; },
                                        [&](uint32_t off) { return crci.t[off >> 2]; });
        for (int t4 = 0; t4 < type2 / 4; ++t4) {
            const uint32_t v = spread4((outw[t4 >> 2] >> (4 * (t4 & 3))) & 0xfu);
            std::memcpy(out + (size_t)blk * out_stride + 4 * t4, &v, 4);
        }
    }
    if (fast_rows) *fast_rows = fast;
    return 0;
}

extern "C" int lmac_emul_decode(int type345, int type2, int type1, int a, const uint8_t* type5, int n_blocks, int in_stride,
                                const uint32_t* scramb_init, uint8_t* out, int out_stride, int32_t* crc_ok) {
    return lmac_emul_decode_route(type345, type2, type1, a, type5, n_blocks, in_stride, scramb_init, out, out_stride, crc_ok, 0, nullptr);
}

// tetra_lmac_decode_frames_device, one job: the lane code of k_lmac_frames for every listed frame.  frames [n_frames][16] packed,
// frame_scramb per frame slot (NULL: SCRAMB_INIT).  out rows: type2 bits (BBK: 30 bits + 2 zero bytes).
extern "C" int lmac_emul_decode_frames(int tpsap, int blk_num, const uint32_t* frames, const int32_t* frame_type, const int32_t* row_frame,
                                       int n_rows, const uint32_t* frame_scramb, uint8_t* out, int out_stride, int32_t* crc_ok) {
    int layout = kLayoutNone, type345 = 0, type2 = 0, type1 = 0, a = 0;
    switch (tpsap) {
        case TETRA_TPSAP_T_SB1: layout = blk_num == 1 ? kLayoutSb1 : kLayoutNone; type345 = 120; type2 = 80; type1 = 60; a = 11; break;
        case TETRA_TPSAP_T_SB2: layout = blk_num == 2 ? kLayoutSb2 : kLayoutNone; type345 = 216; type2 = 144; type1 = 124; a = 101; break;
        case TETRA_TPSAP_T_NDB: layout = blk_num == 1 ? kLayoutNdb1 : blk_num == 2 ? kLayoutNdb2 : kLayoutNone; type345 = 216; type2 = 144; type1 = 124; a = 101; break;
        case TETRA_TPSAP_T_BBK: layout = kLayoutBbk; break;
        case TETRA_TPSAP_T_SCH_F: layout = kLayoutSchF; type345 = 432; type2 = 288; type1 = 268; a = 103; break;
        default: break;
    }
    if (layout == kLayoutNone) return -1;
    static const CrcInvTable crci = make_crc_inv_table();
    const uint32_t* tab = seq_table();
    auto seq = [&](int t, uint32_t byte, int w) { return tab[((size_t)t * 256 + byte) * kSeqStride + w]; };
    for (int blk = 0; blk < n_rows; ++blk) {
        const int f = row_frame[blk];
        const uint32_t code = frame_scramb && tpsap != TETRA_TPSAP_T_SB1 ? frame_scramb[f] : kScrambInitSb1;
        const uint32_t* fw = frames + (size_t)f * kFrameWords;
        uint8_t* row = out + (size_t)blk * out_stride;
        if (layout == kLayoutBbk) {
            const uint32_t x = bbk_bits(fw, frame_type[f]);
            const uint32_t sw = seq(0, code & 0xffu, 0) ^ seq(1, (code >> 8) & 0xffu, 0) ^ seq(2, (code >> 16) & 0xffu, 0) ^ seq(3, code >> 24, 0);
            const uint32_t y = (x ^ sw) & 0xfffffffcu;
            for (int k = 0; k < 8; ++k) { const uint32_t v = bbk_bytes(y, k); std::memcpy(row + 4 * k, &v, 4); }
            crc_ok[blk] = 1;
            continue;
        }
        uint32_t xb[kSeqWords], cls[kSeqWords], dec[(kMaxType2 + kFlush) / 2];
        uint16_t outw[kMaxType2 / 16];
        frame_block(layout, fw, frame_type[f], xb);
        descramble_bits(type345, code, xb, seq, [&](int w, uint32_t word) { cls[w] = word; });
        int pos = a;
        auto fetch = [&] {
            Raw3 r;
            for (int k = 0; k < 3; ++k) {
                const int p = interleave_next(pos, a, type345);
                r.w[k] = cls[p >> 5];
                r.at[k] = 31u - (uint32_t)(p & 31);
            }
            return r;
        };
        viterbi_forward(type2, fetch,
                        [&](const Raw3& r) { return bm_from_masks(bfe_mask(r.w[0], r.at[0]), bfe_mask(r.w[1], r.at[1]), bfe_mask(r.w[2], r.at[2])); },
                        [&](int u, uint32_t word) { dec[u] = word; });
        (void)type1;
        crc_ok[blk] = viterbi_traceback(type2, [&](int u) { return dec[u]; }, [&](int h, uint32_t half) { outw[h] = (uint16_t)half; },
                                        [&](uint32_t off) { return crci.t[off >> 2]; });
        for (int t4 = 0; t4 < type2 / 4; ++t4) {
            const uint32_t v = spread4((outw[t4 >> 2] >> (4 * (t4 & 3))) & 0xfu);
            std::memcpy(row + 4 * t4, &v, 4);
        }
    }
    return 0;
}
