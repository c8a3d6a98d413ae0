// demux_core.hpp -- the burst demultiplexer's thread-level code (include/tetra_burst_sync.h: tetra_burst_demux_*), shared by the
// gfx950 kernels (tetra_burst_sync.hip) and their host build (tests/emul/bsync_emul.cpp, -DTETRA_HOST_EMUL), where every
// (workgroup, thread) index of a launch is run in turn against the restated tetra_burst_rx_cb (src/decoder/src/phy/tetra_burst.c:343-393)
// without a GPU.  Bit moves and index arithmetic only.
#pragma once

#include <stdint.h>
#include <stddef.h>

#include "../../include/tetra_burst_sync.h"

#if defined(__HIPCC__) && !defined(TETRA_HOST_EMUL)
#define DM_FN __device__ __forceinline__
#define DM_HD __host__ __device__ inline
#else
#define DM_FN static inline
#define DM_HD static inline
#endif

namespace demux_core {

constexpr int kThreads = 256;      // threads per workgroup of every demultiplexer kernel
constexpr int kRowIters = 4;       // row groups per wavefront in rows_thread (a wave per 432 bytes is bound by the wave launch rate)

struct alignas(8) U2 { uint32_t x, y; };

// which bits of a burst form the requested block: up to two pieces (offset, length) -- tetra_burst.c:33-49, :343-393
struct Pieces { int off0, len0, off1, len1; };
DM_HD Pieces pieces_for(int train, int tpsap, int blk_num) {
    Pieces p = { 0, 0, 0, 0 };
    if (train == TETRA_TRAIN_SYNC) {
        if (tpsap == TETRA_TPSAP_T_SB1 && blk_num == 1) p = { 94, 120, 0, 0 };
        else if (tpsap == TETRA_TPSAP_T_BBK) p = { 252, 30, 0, 0 };
        else if (tpsap == TETRA_TPSAP_T_SB2 && blk_num == 2) p = { 282, 216, 0, 0 };
    } else if (train == TETRA_TRAIN_NORM_1 || train == TETRA_TRAIN_NORM_2) {
        if (tpsap == TETRA_TPSAP_T_BBK) p = { 230, 14, 266, 16 };
        else if (train == TETRA_TRAIN_NORM_2 && tpsap == TETRA_TPSAP_T_NDB && blk_num == 1) p = { 14, 216, 0, 0 };
        else if (train == TETRA_TRAIN_NORM_2 && tpsap == TETRA_TPSAP_T_NDB && blk_num == 2) p = { 282, 216, 0, 0 };
        else if (train == TETRA_TRAIN_NORM_1 && tpsap == TETRA_TPSAP_T_SCH_F) p = { 14, 216, 282, 216 };
    }
    return p;
}
// the three candidates of a launch (the kind is a launch argument; only the burst type differs per frame)
struct PiecesLut { Pieces sync, norm1, norm2; };
DM_HD PiecesLut lut_for(int tpsap, int blk_num) {
    return PiecesLut{ pieces_for(TETRA_TRAIN_SYNC, tpsap, blk_num), pieces_for(TETRA_TRAIN_NORM_1, tpsap, blk_num),
                      pieces_for(TETRA_TRAIN_NORM_2, tpsap, blk_num) };
}

// four consecutive bits of a block (positions 4 d .. 4 d + 3 of its up to two pieces) as four bytes, first bit in the low byte.
// PACKED: the frame is 16 words, first bit most significant (k_burst_sync<true>); else 512 bytes, one bit per byte.
template <bool PACKED> DM_FN uint32_t demux_dword(const uint8_t* frames, int r, const Pieces& p, int d) {
    uint32_t v = 0;
    if (PACKED) {
        const uint32_t* f = reinterpret_cast<const uint32_t*>(frames) + (size_t)r * TETRA_FRAME_WORDS;
        const int i = 4 * d;
        int x = -1;                                       // all four bits inside one piece: one 4-bit window of the packed row
        if (i + 4 <= p.len0) x = p.off0 + i;
        else if (i >= p.len0 && i + 4 <= p.len0 + p.len1) x = p.off1 + i - p.len0;
        if (x >= 0) {
            const int w = x >> 5;
            const uint64_t two = ((uint64_t)f[w] << 32) | (w + 1 < TETRA_FRAME_WORDS ? f[w + 1] : 0u);
            const uint32_t nib = (uint32_t)(two >> (60 - (x & 31))) & 0xfu;      // first bit = most significant
            return ((nib >> 3) & 1u) | (((nib >> 2) & 1u) << 8) | (((nib >> 1) & 1u) << 16) | ((nib & 1u) << 24);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {                      // a piece boundary or the block's end inside these four bits
            const int ii = i + k;
            const int xx = ii < p.len0 ? p.off0 + ii : (ii < p.len0 + p.len1 ? p.off1 + ii - p.len0 : -1);
            if (xx >= 0) v |= ((f[xx >> 5] >> (31 - (xx & 31))) & 1u) << (8 * k);
        }
    } else {
        const uint8_t* f = frames + (size_t)r * TETRA_FRAME_STRIDE;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 4 * d + k;
            uint32_t byte = 0;
            if (i < p.len0) byte = f[p.off0 + i];
            else if (i < p.len0 + p.len1) byte = f[p.off1 + i - p.len0];
            v |= byte << (8 * k);
        }
    }
    return v;
}

// eight consecutive bits of a block (positions 8 d .. 8 d + 7) as eight bytes.  PACKED frames, all eight inside one piece (every
// pair of the coded blocks: their pieces start and end on multiples of 8; all but two pairs of the BBK): ONE 8-bit window of the
// packed row, spread with three 32-bit operations per half -- the counters showed the slot-layout demultiplexer bound by its vector
// instructions, not by HBM (175 per 8 output bytes, profiles/r05/r05_k_chain_tail_counters.md).
template <bool PACKED> DM_FN U2 demux_pair(const uint8_t* frames, int r, const Pieces& p, int d) {
    if (8 * d >= p.len0 + p.len1) return U2{ 0u, 0u };      // behind the block -- or the frame does not carry the kind (three slots in four)
    if (PACKED) {
        const int i = 8 * d;
        int x = -1;
        if (i + 8 <= p.len0) x = p.off0 + i;
        else if (i >= p.len0 && i + 8 <= p.len0 + p.len1) x = p.off1 + i - p.len0;
        if (x >= 0) {
            const uint32_t* f = reinterpret_cast<const uint32_t*>(frames) + (size_t)r * TETRA_FRAME_WORDS;
            const int w = x >> 5;
            const uint64_t two = ((uint64_t)f[w] << 32) | (w + 1 < TETRA_FRAME_WORDS ? f[w + 1] : 0u);
            const uint32_t rep = ((uint32_t)(two >> (56 - (x & 31))) & 0xffu) * 0x01010101u;      // the byte in every byte; first bit = bit 7
            // byte k of the result = bit 7 - k: keep that one bit per byte, then "non-zero byte -> 1" (+ 0x7f carries into bit 7 only)
            const uint32_t lo = (((rep & 0x10204080u) + 0x7f7f7f7fu) >> 7) & 0x01010101u;
            const uint32_t hi = (((rep & 0x01020408u) + 0x7f7f7f7fu) >> 7) & 0x01010101u;
            return U2{ lo, hi };
        }
    }
    return U2{ demux_dword<PACKED>(frames, r, p, 2 * d), demux_dword<PACKED>(frames, r, p, 2 * d + 1) };
}

// (workgroup, thread) -> (row, unit within the row) with a 32-bit division wherever the launch's index space allows it (a 64-bit
// division by a run-time divisor costs ~100 vector instructions per thread)
DM_FN bool index_of(uint32_t block, uint32_t thread, long long units_total, int row_u, int& r, int& d) {
    if (units_total <= 0xffffff00ll) {
        const uint32_t g = block * (uint32_t)kThreads + thread;
        if (g >= (uint32_t)units_total) return false;
        const uint32_t q = g / (uint32_t)row_u;
        r = (int)q; d = (int)(g - q * (uint32_t)row_u);
        return true;
    }
    const long long gid = (long long)block * kThreads + thread;
    if (gid >= units_total) return false;
    r = (int)(gid / row_u); d = (int)(gid % row_u);
    return true;
}

// ---- one thread of each kernel --------------------------------------------------------------------------------------------------
// k_burst_demux: one thread per output dword -- or, WIDE (rows a multiple of 8 bytes, 8-byte aligned), per pair of dwords
template <bool PACKED, bool WIDE> DM_FN void demux_thread(uint32_t block, uint32_t thread, const uint8_t* frames, const int* frame_type, int n,
                                                           int tpsap, int blk_num, uint8_t* rows, int row_stride, int* valid) {
    const int row_u = row_stride >> (WIDE ? 3 : 2);
    int r, d;
    if (!index_of(block, thread, (long long)n * row_u, row_u, r, d)) return;
    const Pieces p = pieces_for(frame_type[r], tpsap, blk_num);
    if (WIDE) reinterpret_cast<U2*>(rows + (size_t)r * row_stride)[d] = demux_pair<PACKED>(frames, r, p, d);
    else reinterpret_cast<uint32_t*>(rows + (size_t)r * row_stride)[d] = demux_dword<PACKED>(frames, r, p, d);
    if (d == 0) valid[r] = p.len0 > 0;
}

// k_demux_gather: the compacted rows (row j <- frame row_frame[j]); rows past n_rows do not exist
template <bool PACKED, bool WIDE> DM_FN void gather_thread(uint32_t block, uint32_t thread, const uint8_t* frames, const int* frame_type,
                                                            const int* row_frame, int n_rows, int n, int tpsap, int blk_num, uint8_t* rows,
                                                            int row_stride) {
    const int row_u = row_stride >> (WIDE ? 3 : 2);
    int j, d;
    if (!index_of(block, thread, (long long)n * row_u, row_u, j, d)) return;
    if (j >= n_rows) return;
    const int r = row_frame[j];
    const Pieces p = pieces_for(frame_type[r], tpsap, blk_num);
    if (WIDE) reinterpret_cast<U2*>(rows + (size_t)j * row_stride)[d] = demux_pair<PACKED>(frames, r, p, d);
    else reinterpret_cast<uint32_t*>(rows + (size_t)j * row_stride)[d] = demux_dword<PACKED>(frames, r, p, d);
}

// k_demux_rows: PACKED frames, 8-byte row units, rows of at most 512 bytes (every block kind: 120 .. 432): a wavefront takes
// 64 / row_u whole rows at a time -- lane -> (row, unit) by one multiply (inv_row_u = ceil(65536 / row_u): exact for lane < 64), the
// block's pieces picked from the launch's three candidates field by field (selecting whole structs sends them through scratch
// memory), one 8-bit window per lane, and the wave's stores cover rows_per_wave consecutive rows = one contiguous run.
// GATHER: the compacted rows; `have` = rows that exist (n, or *n_rows).
template <bool GATHER> DM_FN void rows_thread(uint32_t block, uint32_t thread, const uint8_t* frames, const int* frame_type, const int* row_frame,
                                               long long have, const PiecesLut& lut, int row_u, int rows_per_wave, uint32_t inv_row_u,
                                               uint8_t* rows, int* valid) {
    const int lane = (int)(thread & 63u);
    const int lr = (int)(((uint32_t)lane * inv_row_u) >> 16);
    const int d = lane - lr * row_u;
    if (lr >= rows_per_wave) return;
#pragma unroll 1
    for (int it = 0; it < kRowIters; ++it) {
        const long long wave = ((long long)block * (kThreads / 64) + (thread >> 6)) * kRowIters + it;
        const long long j = wave * rows_per_wave + lr;
        if (j >= have) continue;
        const int r = GATHER ? row_frame[j] : (int)j;
        const int t = frame_type[r];
        const bool is_s = t == TETRA_TRAIN_SYNC, is_1 = t == TETRA_TRAIN_NORM_1, is_2 = t == TETRA_TRAIN_NORM_2;
        Pieces p;
        p.off0 = is_s ? lut.sync.off0 : is_1 ? lut.norm1.off0 : is_2 ? lut.norm2.off0 : 0;
        p.len0 = is_s ? lut.sync.len0 : is_1 ? lut.norm1.len0 : is_2 ? lut.norm2.len0 : 0;
        p.off1 = is_s ? lut.sync.off1 : is_1 ? lut.norm1.off1 : is_2 ? lut.norm2.off1 : 0;
        p.len1 = is_s ? lut.sync.len1 : is_1 ? lut.norm1.len1 : is_2 ? lut.norm2.len1 : 0;
        reinterpret_cast<U2*>(rows + (size_t)j * ((size_t)row_u * 8))[d] = demux_pair<true>(frames, r, p, d);
        if (!GATHER && d == 0) valid[j] = p.len0 > 0;
    }
}

// what the launchers decide, restated for both builds: whole rows per wavefront when the frames are packed, the rows 8-byte units and
// at most 512 bytes long
DM_HD bool use_rows_kernel(bool packed, bool wide, int row_stride) { return packed && wide && row_stride <= 512; }
DM_HD long long rows_grid(int n, int row_stride) {
    const int row_u = row_stride >> 3, rpw = 64 / row_u;
    const long long waves = ((long long)n + rpw - 1) / rpw;
    return (waves + (kThreads / 64) * kRowIters - 1) / ((kThreads / 64) * kRowIters);
}
DM_HD long long units_grid(int n, int row_stride, bool wide) {
    const long long total = (long long)n * (row_stride >> (wide ? 3 : 2));
    return (total + kThreads - 1) / kThreads;
}

}  // namespace demux_core
