// tetra_burst_sync.hip -- batched burst synchroniser + burst demultiplexer (include/tetra_burst_sync.h).
//
// k_burst_sync: one workgroup of four wavefronts per channel (the byte work of steps 1, 2 and 4 on all 256 lanes; step 3 on one wave).
//   1. the channel's stream of this call -- carried buffer (<= 4096 bits, one byte per bit in HBM) followed by the new
//      bits -- is packed 32 bits per word into LDS; the new bits start word-aligned, 32 bytes -> one word per lane-step
//      with 16-byte loads and one multiply per 8 bytes (the scan kernel's trick);
//   2. three match bitmaps (sync / normal 1 / normal 2 training sequence present at x) are built 32 positions per lane-step;
//   3. the wave walks the event-driven state machine of bsync_core.hpp (about 70 events per second of signal) in lock
//      step; its "first set bit in [a, b)" searches over the bitmaps are wave-cooperative (one word per lane + ballot);
//      the consumed frames are listed in LDS;
//   4. all lanes expand the listed frames to one byte per bit with coalesced dword stores, and write the new carried
//      buffer and the state.
// Byte work: every input byte is read from HBM once, every output byte written once.
// k_burst_demux: one thread per output dword; a gather with the offsets of tetra_burst_rx_cb().
#include <hip/hip_runtime.h>

#include "../../include/tetra_burst_sync.h"
#include "bsync_core.hpp"
#include "demux_core.hpp"

namespace {

using namespace bsync_core;

constexpr int kLanes = 64;
constexpr int kThreadsBS = 256;       // k_burst_sync: four wavefronts per channel for the byte work, one of them walks the state machine
constexpr int kMaxBitsLimit = 262144;

struct FrameRec { int bx; int type; uint32_t bitnum; };

// PACKED: the consumed frames leave as 16 words of 32 bits (first bit = most significant; word 15 holds bits 480 .. 509 in its top 30
// bits) instead of 512 bytes: an eighth of the bytes for the demultiplexer behind (tetra_bsync_process_packed_device).
template <bool PACKED> __global__ __launch_bounds__(kThreadsBS) void k_burst_sync(const uint8_t* __restrict__ bits, int bits_stride, const int* __restrict__ n_bits,
                                                       int max_bits, int max_frames, State* __restrict__ states,
                                                       uint8_t* __restrict__ carry, uint8_t* __restrict__ frames,
                                                       int* __restrict__ frame_type, uint32_t* __restrict__ frame_bitnum,
                                                       int* __restrict__ n_frames) {
    extern __shared__ uint32_t lds[];
    const int words = stream_words(max_bits);
    uint32_t* s = lds;
    uint32_t* m_sync = s + words;
    uint32_t* m_n1 = m_sync + words;
    uint32_t* m_n2 = m_n1 + words;
    uint32_t* m_any = m_n2 + words;
    FrameRec* rec = reinterpret_cast<FrameRec*>(m_any + words);
    __shared__ int sh_nframes, sh_carry_x;
    __shared__ State sh_state;

    const int ch = blockIdx.x, tid = threadIdx.x, lane = tid & (kLanes - 1);
    const uint8_t* in = bits + (size_t)ch * bits_stride;
    uint8_t* cbuf = carry + (size_t)ch * kBuf;
    State st = states[ch];
    int n_new = n_bits[ch];
    n_new = n_new < 0 ? 0 : (n_new > max_bits ? max_bits : n_new);
    n_new = n_new > bits_stride ? bits_stride : n_new;         // a poisoned count never reads into the next channel's row
    const int x0 = kOff - (int)st.bits_in_buf, xe = kOff + n_new;

    // 1. pack the stream
    for (int w = tid; w < words; w += kThreadsBS) {
        uint32_t v = 0;
        const int xb = 32 * w;
        if (xb + 32 > x0 && xb < kOff) {                         // carried bits: byte by byte (at most 128 words)
            for (int b = 0; b < 32; ++b) {
                const int x = xb + b;
                if (x >= x0 && x < kOff) v |= (uint32_t)(cbuf[x - x0] & 1u) << (31 - b);
            }
        } else if (xb >= kOff && xb < xe) {
            const int j = xb - kOff;
            if (j + 32 <= n_new && (((uintptr_t)(in + j)) & 15) == 0) {
                const uint4 lo = *reinterpret_cast<const uint4*>(in + j);
                const uint4 hi = *reinterpret_cast<const uint4*>(in + j + 16);
                const unsigned long long q[4] = { ((unsigned long long)lo.y << 32) | lo.x, ((unsigned long long)lo.w << 32) | lo.z,
                                                  ((unsigned long long)hi.y << 32) | hi.x, ((unsigned long long)hi.w << 32) | hi.z };
#pragma unroll
                for (int z = 0; z < 4; ++z)      // 8 bytes -> 8 bits, first byte = MSB: byte i of x * 0x8040201008040201 reaches bit 63 - i
                    v |= (uint32_t)(((q[z] & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56) << (24 - 8 * z);
            } else {
                for (int b = 0; b < 32; ++b)
                    if (j + b < n_new) v |= (uint32_t)(in[j + b] & 1u) << (31 - b);
            }
        }
        s[w] = v;
    }
    __syncthreads();

    // 2. match bitmaps
    for (int w = tid; w < words; w += kThreadsBS) {
        uint32_t a = 0, b = 0, c = 0;
        if (w + 2 < words && 32 * w + 32 > x0 && 32 * w < xe) match_word(s, w, x0, xe, a, b, c);
        m_sync[w] = a;
        m_n1[w] = b;
        m_n2[w] = c;
        m_any[w] = a | b | c;
    }
    __syncthreads();

    // 3. state machine: all lanes in lock step (uniform control flow); the bitmap searches are wave-cooperative
    auto first_wave = [&](const uint32_t* m, int a, int b) -> int {
        if (a >= b) return -1;
        const int w0 = a >> 5, w1 = (b - 1) >> 5;
        for (int base = w0; base <= w1; base += kLanes) {
            const int w = base + lane;
            uint32_t v = (w <= w1) ? m[w] : 0u;
            if (w == w0) v &= 0xffffffffu << (a & 31);
            if (w == w1 && (b & 31) != 0) v &= (1u << (b & 31)) - 1u;
            const unsigned long long hit = __ballot(v != 0u);
            if (hit) {
                const int l = __builtin_ctzll(hit);
                return 32 * (base + l) + __builtin_ctz(__shfl(v, l));
            }
        }
        return -1;
    };
    if (tid < kLanes) {      // (one wavefront walks the events; the other three join again for the byte work behind the barrier)
        int carry_x = 0;
        // LOCKED steady state: one frame per lane, 64 per round, up to and including the first frame that unlocks the receiver
        auto batch = [&](int bx, int K, int f0, uint32_t abs_bx, bool& unlocked) -> int {
            int done = 0;
            while (done < K) {
                const int k = done + lane, here = K - done < kLanes ? K - done : kLanes;
                FrameEval e = { -1, false };
                if (lane < here) e = locked_frame_eval(s, m_sync, m_n1, m_n2, m_any, bx + kTs * k);
                const unsigned long long um = __ballot(e.unlocks);
                const int take = um ? __builtin_ctzll(um) + 1 : here;
                if (lane < take && f0 + k < max_frames) rec[f0 + k] = FrameRec{ bx + kTs * k, e.reported, abs_bx + (uint32_t)(kTs * k) };
                done += take;
                if (um) { unlocked = true; break; }
            }
            return done;
        };
        const int nrun = run(st, s, m_sync, m_n1, m_n2, m_any, n_new, carry_x, first_wave, [&](int f, int bx, int type, uint32_t bitnum) {
            if (lane == 0 && f < max_frames) rec[f] = FrameRec{ bx, type, bitnum };
        }, batch);
        if (lane == 0) {
            sh_nframes = nrun < max_frames ? nrun : max_frames;
            sh_carry_x = carry_x;
            sh_state = st;
        }
    }
    __syncthreads();

    // 4. frames, carry, state
    const int nf = sh_nframes;
    if (PACKED) {
        uint32_t* pout = reinterpret_cast<uint32_t*>(frames) + (size_t)ch * max_frames * TETRA_FRAME_WORDS;
        for (int i = tid; i < nf * TETRA_FRAME_WORDS; i += kThreadsBS) {
            const int f = i / TETRA_FRAME_WORDS, w = i % TETRA_FRAME_WORDS;
            uint32_t v = window(s, rec[f].bx + 32 * w, 32);
            if (w == TETRA_FRAME_WORDS - 1) v &= 0xfffffffcu;              // bits 510, 511 of the row are not the frame's
            pout[i] = v;
        }
    }
    uint8_t* fout = frames + (size_t)ch * max_frames * TETRA_FRAME_STRIDE;
    for (int f = 0; !PACKED && f < nf; ++f) {
        const int bx = rec[f].bx;
        uint32_t* dst = reinterpret_cast<uint32_t*>(fout + (size_t)f * TETRA_FRAME_STRIDE);
        for (int d = tid; d < TETRA_FRAME_STRIDE / 4; d += kThreadsBS) {
            uint32_t nib = window(s, bx + 4 * d, 4);                       // first bit = MSB
            if (4 * d + 4 > kTs) nib &= (4 * d >= kTs) ? 0u : (0xfu << (4 * d + 4 - kTs)) & 0xfu;
            dst[d] = ((nib >> 3) & 1u) | (((nib >> 2) & 1u) << 8) | (((nib >> 1) & 1u) << 16) | ((nib & 1u) << 24);
        }
    }
    for (int f = tid; f < max_frames; f += kThreadsBS) {
        frame_type[(size_t)ch * max_frames + f] = f < nf ? rec[f].type : TETRA_FRAME_NONE;
        frame_bitnum[(size_t)ch * max_frames + f] = f < nf ? rec[f].bitnum : 0u;
    }
    const int cx = sh_carry_x, nc = xe - cx;
    for (int i = tid; i < nc; i += kThreadsBS) cbuf[i] = (uint8_t)get_bit(s, cx + i);
    if (tid == 0) {
        states[ch] = sh_state;
        n_frames[ch] = nf;
    }
}

using demux_core::Pieces;
using demux_core::PiecesLut;
using demux_core::pieces_for;
using demux_core::lut_for;

// The demultiplexer kernels: thread-level code in demux_core.hpp (shared with the host emulation).
// one thread per output dword -- or, WIDE (rows a multiple of 8 bytes, 8-byte aligned), per pair of dwords: half the threads, 8-byte stores
template <bool PACKED, bool WIDE> __global__ __launch_bounds__(256) void k_burst_demux(const uint8_t* __restrict__ frames, const int* __restrict__ frame_type, int n,
                                                     int tpsap, int blk_num, uint8_t* __restrict__ rows, int row_stride,
                                                     int* __restrict__ valid) {
    demux_core::demux_thread<PACKED, WIDE>(blockIdx.x, threadIdx.x, frames, frame_type, n, tpsap, blk_num, rows, row_stride, valid);
}

// ---- compacting form of the demultiplexer: only frames that carry the block kind produce a row, in frame order ----
// 1. per block of 256 frames: how many carry it
__global__ __launch_bounds__(256) void k_demux_count(const int* __restrict__ frame_type, int n, int tpsap, int blk_num,
                                                     int* __restrict__ block_count) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    const bool has = r < n && pieces_for(frame_type[r], tpsap, blk_num).len0 > 0;
    const int c = __syncthreads_count(has ? 1 : 0);
    if (threadIdx.x == 0) block_count[blockIdx.x] = c;
}
// 2. exclusive scan of the block counts (one workgroup; nblocks is a few thousand), total -> *n_rows
__global__ __launch_bounds__(1024) void k_demux_scan(int* __restrict__ block_count, int nblocks, int* __restrict__ n_rows) {
    __shared__ int part[1024];
    const int per = (nblocks + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(nblocks, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += block_count[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                       // Hillis-Steele inclusive scan
        const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) { const int c = block_count[i]; block_count[i] = run; run += c; }
    if (threadIdx.x == 1023) *n_rows = part[1023];
}
// 3. row j <- frame index, frame order kept
__global__ __launch_bounds__(256) void k_demux_index(const int* __restrict__ frame_type, int n, int tpsap, int blk_num,
                                                     const int* __restrict__ block_off, int* __restrict__ row_frame) {
    __shared__ int wave_cnt[4];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const bool has = r < n && pieces_for(frame_type[r], tpsap, blk_num).len0 > 0;
    const unsigned long long m = __ballot(has);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[w] = __popcll(m);
    __syncthreads();
    int base = block_off[blockIdx.x];
    for (int i = 0; i < w; ++i) base += wave_cnt[i];
    if (has) row_frame[base + __popcll(m & ((1ull << lane) - 1ull))] = r;
}
// 4. the gather itself, one thread per output dword of the worst case; rows past *n_rows do not exist
template <bool PACKED, bool WIDE> __global__ __launch_bounds__(256) void k_demux_gather(const uint8_t* __restrict__ frames, const int* __restrict__ frame_type,
                                                      const int* __restrict__ row_frame, const int* __restrict__ n_rows, int n,
                                                      int tpsap, int blk_num, uint8_t* __restrict__ rows, int row_stride) {
    demux_core::gather_thread<PACKED, WIDE>(blockIdx.x, threadIdx.x, frames, frame_type, row_frame, *n_rows, n, tpsap, blk_num, rows, row_stride);
}

// PACKED frames, 8-byte row units, rows of at most 512 bytes: whole rows per wavefront (demux_core::rows_thread).  GATHER: the compacted rows.
template <bool GATHER> __global__ __launch_bounds__(256) void k_demux_rows(const uint8_t* __restrict__ frames, const int* __restrict__ frame_type,
                                                                         const int* __restrict__ row_frame, const int* __restrict__ n_rows, int n,
                                                                         PiecesLut lut, int row_u, int rows_per_wave, unsigned inv_row_u,
                                                                         uint8_t* __restrict__ rows, int* __restrict__ valid) {
    const long long have = GATHER ? (long long)*n_rows : (long long)n;
    demux_core::rows_thread<GATHER>(blockIdx.x, threadIdx.x, frames, frame_type, row_frame, have, lut, row_u, rows_per_wave, inv_row_u, rows, valid);
}

// ---- the frame lists of a call (tetra_burst_index_device): SYNC / NORM_1 / NORM_2 / any, frame order ----------------------------
__device__ __forceinline__ unsigned list_mask(int t) {      // bit k set <=> a frame of type t belongs to list k
    return t == TETRA_TRAIN_SYNC ? (1u << TETRA_LIST_SYNC) | (1u << TETRA_LIST_ANY)
         : t == TETRA_TRAIN_NORM_1 ? (1u << TETRA_LIST_NORM_1) | (1u << TETRA_LIST_ANY)
         : t == TETRA_TRAIN_NORM_2 ? (1u << TETRA_LIST_NORM_2) | (1u << TETRA_LIST_ANY) : 0u;
}
// 1. entries per 256 frames and list
__global__ __launch_bounds__(256) void k_index_count(const int* __restrict__ frame_type, int n, int nblocks, int* __restrict__ work) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    const unsigned m = r < n ? list_mask(frame_type[r]) : 0u;
#pragma unroll
    for (int k = 0; k < TETRA_N_LISTS; ++k) {
        const int c = __syncthreads_count((m >> k) & 1u);
        if (threadIdx.x == 0) work[k * nblocks + blockIdx.x] = c;
    }
}
// 2. exclusive scan of each list's block counts, totals -> counts.  One workgroup of FOUR wavefronts (a run of blocks per thread,
//    shuffles within a wavefront, one exchange between the four): a workgroup has to find ONE compute unit with room for all its
//    waves, and beside the demodulator -- whose 199-register waves leave 112 registers on two of a CU's four SIMDs -- the
//    1024-thread form of this kernel (four waves of 32 registers per SIMD) found none until the demodulator's launch was over: the
//    tail of the receive chain then ran BEHIND the demodulator it was meant to overlap (two-stream chain 4.17 instead of 3.98 ms).
constexpr int kScanThreads = 256;
__global__ __launch_bounds__(kScanThreads) void k_index_scan(int* __restrict__ work, int nblocks, int* __restrict__ counts) {
    __shared__ int wave_sum[TETRA_N_LISTS][kScanThreads / 64];
    const int per = (nblocks + kScanThreads - 1) / kScanThreads;
    const int lo = threadIdx.x * per, hi = min(nblocks, lo + per);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int sum[TETRA_N_LISTS], inc[TETRA_N_LISTS];
#pragma unroll
    for (int k = 0; k < TETRA_N_LISTS; ++k) {
        sum[k] = 0;
        for (int i = lo; i < hi; ++i) sum[k] += work[k * nblocks + i];
        inc[k] = sum[k];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(inc[k], d);
            inc[k] += lane >= d ? v : 0;
        }
        if (lane == 63) wave_sum[k][w] = inc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TETRA_N_LISTS; ++k) {
        int run = inc[k] - sum[k];
        for (int i = 0; i < w; ++i) run += wave_sum[k][i];
        for (int i = lo; i < hi; ++i) { const int c = work[k * nblocks + i]; work[k * nblocks + i] = run; run += c; }
        if (threadIdx.x == kScanThreads - 1) counts[k] = run;
    }
}
// 3. the lists themselves, and per channel the position of its first entry
__global__ __launch_bounds__(256) void k_index_write(const int* __restrict__ frame_type, int n, int nblocks, int frames_per_channel,
                                                     const int* __restrict__ work, int* __restrict__ lists, int* __restrict__ chan_first) {
    __shared__ int wave_cnt[TETRA_N_LISTS][4];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const unsigned m = r < n ? list_mask(frame_type[r]) : 0u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long b[TETRA_N_LISTS];
#pragma unroll
    for (int k = 0; k < TETRA_N_LISTS; ++k) {
        b[k] = __ballot((m >> k) & 1u);
        if (lane == 0) wave_cnt[k][w] = __popcll(b[k]);
    }
    __syncthreads();
    const bool first_of_channel = chan_first && r < n && r % frames_per_channel == 0;
    const int chans = first_of_channel ? n / frames_per_channel : 0;
#pragma unroll
    for (int k = 0; k < TETRA_N_LISTS; ++k) {
        int at = work[k * nblocks + blockIdx.x];
        for (int i = 0; i < w; ++i) at += wave_cnt[k][i];
        at += __popcll(b[k] & ((1ull << lane) - 1ull));
        if ((m >> k) & 1u) lists[(size_t)k * n + at] = r;
        if (first_of_channel) chan_first[(size_t)k * chans + r / frames_per_channel] = at;
    }
}

size_t lds_bytes(int max_bits, int max_frames) { return (size_t)stream_words(max_bits) * 5 * sizeof(uint32_t) + (size_t)max_frames * sizeof(FrameRec); }

}  // namespace

struct tetra_bsync {
    int n_channels, max_bits, max_frames, device;
    State* d_state;
    uint8_t* d_carry;
};

extern "C" {

int tetra_bsync_create(int n_channels, int max_bits, int device, tetra_bsync_t** out) {
    if (!out || n_channels < 1 || max_bits < 1 || max_bits > kMaxBitsLimit) return TETRA_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TETRA_ERR_NO_DEVICE;
    if (device >= ndev) return TETRA_ERR_NO_DEVICE;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return TETRA_ERR_HIP;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return TETRA_ERR_HIP;
    tetra_bsync* h = new tetra_bsync{ n_channels, max_bits, (kBuf + max_bits) / kTs + 2, device, nullptr, nullptr };
    if (lds_bytes(max_bits, h->max_frames) > 160 * 1024 - 64) { delete h; return TETRA_ERR_UNSUPPORTED; }
    if (hipMalloc(reinterpret_cast<void**>(&h->d_state), sizeof(State) * n_channels) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&h->d_carry), (size_t)kBuf * n_channels) != hipSuccess) {
        (void)hipFree(h->d_state);
        delete h;
        return TETRA_ERR_NOMEM;
    }
    // the attribute belongs to the kernel, not to the handle: always raise it to the ceiling so that handles of different
    // sizes can coexist
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_burst_sync<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 64) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_burst_sync<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 64) != hipSuccess) {
        tetra_bsync_destroy(h);
        return TETRA_ERR_HIP;
    }
    *out = h;
    return tetra_bsync_reset(h);
}

int tetra_bsync_destroy(tetra_bsync_t* h) {
    if (!h) return TETRA_ERR_ARG;
    (void)hipFree(h->d_state);
    (void)hipFree(h->d_carry);
    delete h;
    return TETRA_OK;
}

int tetra_bsync_reset(tetra_bsync_t* h) {
    if (!h) return TETRA_ERR_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return TETRA_ERR_HIP;
    if (hipMemset(h->d_state, 0, sizeof(State) * h->n_channels) != hipSuccess ||
        hipMemset(h->d_carry, 0, (size_t)kBuf * h->n_channels) != hipSuccess)
        return TETRA_ERR_HIP;
    return TETRA_OK;
}

int tetra_bsync_max_frames(tetra_bsync_t* h) { return h ? h->max_frames : TETRA_ERR_ARG; }

}  // extern "C"

namespace {
template <bool PACKED> int bsync_launch(tetra_bsync_t* h, const uint8_t* d_bits, int bits_stride, const int32_t* d_n_bits, void* d_frames,
                                        int32_t* d_frame_type, uint32_t* d_frame_bitnum, int32_t* d_n_frames, void* hip_stream) {
    if (!h || !d_bits || !d_n_bits || !d_frames || !d_frame_type || !d_frame_bitnum || !d_n_frames) return TETRA_ERR_ARG;
    if (bits_stride < 4) return TETRA_ERR_ARG;
    if (bits_stride < h->max_bits) return TETRA_ERR_SIZE;      // rows must be able to hold the max_bits the handle was sized for
    if ((bits_stride & 3) || ((uintptr_t)d_bits & 3) || ((uintptr_t)d_frames & 3)) return TETRA_ERR_ALIGN;
    hipLaunchKernelGGL(k_burst_sync<PACKED>, dim3(h->n_channels), dim3(kThreadsBS), lds_bytes(h->max_bits, h->max_frames),
                       static_cast<hipStream_t>(hip_stream), d_bits, bits_stride, d_n_bits, h->max_bits, h->max_frames, h->d_state,
                       h->d_carry, static_cast<uint8_t*>(d_frames), d_frame_type, d_frame_bitnum, d_n_frames);
    return hipGetLastError() == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
}
}  // namespace

extern "C" {

int tetra_bsync_process_device(tetra_bsync_t* h, const uint8_t* d_bits, int bits_stride, const int32_t* d_n_bits, uint8_t* d_frames,
                               int32_t* d_frame_type, uint32_t* d_frame_bitnum, int32_t* d_n_frames, void* hip_stream) {
    return bsync_launch<false>(h, d_bits, bits_stride, d_n_bits, d_frames, d_frame_type, d_frame_bitnum, d_n_frames, hip_stream);
}

int tetra_bsync_process_packed_device(tetra_bsync_t* h, const uint8_t* d_bits, int bits_stride, const int32_t* d_n_bits,
                                      uint32_t* d_frames_packed, int32_t* d_frame_type, uint32_t* d_frame_bitnum, int32_t* d_n_frames,
                                      void* hip_stream) {
    return bsync_launch<true>(h, d_bits, bits_stride, d_n_bits, d_frames_packed, d_frame_type, d_frame_bitnum, d_n_frames, hip_stream);
}

int tetra_bsync_process(tetra_bsync_t* h, const uint8_t* bits, int bits_stride, const int32_t* n_bits, uint8_t* frames,
                        int32_t* frame_type, uint32_t* frame_bitnum, int32_t* n_frames) {
    if (!h || !bits || !n_bits || !frames || !frame_type || !frame_bitnum || !n_frames) return TETRA_ERR_ARG;
    if (bits_stride < 4 || (bits_stride & 3)) return TETRA_ERR_ALIGN;
    if (hipSetDevice(h->device) != hipSuccess) return TETRA_ERR_HIP;
    const int C = h->n_channels, F = h->max_frames;
    uint8_t *d_bits = nullptr, *d_frames = nullptr;
    int32_t *d_n = nullptr, *d_ft = nullptr, *d_nf = nullptr;
    uint32_t* d_fb = nullptr;
    int rc = TETRA_ERR_HIP;
    do {
        if (hipMalloc(reinterpret_cast<void**>(&d_bits), (size_t)C * bits_stride) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&d_frames), (size_t)C * F * TETRA_FRAME_STRIDE) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&d_n), sizeof(int32_t) * C) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&d_ft), sizeof(int32_t) * C * F) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&d_fb), sizeof(uint32_t) * C * F) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&d_nf), sizeof(int32_t) * C) != hipSuccess) { rc = TETRA_ERR_NOMEM; break; }
        if (hipMemcpy(d_bits, bits, (size_t)C * bits_stride, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_n, n_bits, sizeof(int32_t) * C, hipMemcpyHostToDevice) != hipSuccess) break;
        const int krc = tetra_bsync_process_device(h, d_bits, bits_stride, d_n, d_frames, d_ft, d_fb, d_nf, nullptr);
        if (krc != TETRA_OK) { rc = krc; break; }
        if (hipDeviceSynchronize() != hipSuccess) break;
        if (hipMemcpy(frames, d_frames, (size_t)C * F * TETRA_FRAME_STRIDE, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(frame_type, d_ft, sizeof(int32_t) * C * F, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(frame_bitnum, d_fb, sizeof(uint32_t) * C * F, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(n_frames, d_nf, sizeof(int32_t) * C, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = TETRA_OK;
    } while (false);
    (void)hipFree(d_bits);
    (void)hipFree(d_frames);
    (void)hipFree(d_n);
    (void)hipFree(d_ft);
    (void)hipFree(d_fb);
    (void)hipFree(d_nf);
    return rc;
}

int tetra_bsync_get_state(tetra_bsync_t* h, int first, int count, tetra_bsync_state_t* out) {
    if (!h || !out || first < 0 || count < 0 || first + count > h->n_channels) return TETRA_ERR_ARG;
    if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return TETRA_ERR_HIP;
    static_assert(sizeof(State) == sizeof(tetra_bsync_state_t), "state layout");
    if (count && hipMemcpy(out, h->d_state + first, sizeof(State) * count, hipMemcpyDeviceToHost) != hipSuccess) return TETRA_ERR_HIP;
    return TETRA_OK;
}

}  // extern "C"

namespace {
template <bool PACKED> int demux_launch(const void* d_frames_v, const int32_t* d_frame_type, int n, int tpsap, int blk_num, uint8_t* d_rows,
                                        int row_stride, int32_t* d_valid, void* hip_stream) {
    const uint8_t* d_frames = static_cast<const uint8_t*>(d_frames_v);
    if (!d_frames || !d_frame_type || !d_rows || !d_valid || n < 0 || tpsap < 0 || tpsap > 5) return TETRA_ERR_ARG;
    if (n == 0) return TETRA_OK;
    // the longest block this kind can have must fit the row
    const Pieces longest = tpsap == TETRA_TPSAP_T_SCH_F ? pieces_for(TETRA_TRAIN_NORM_1, tpsap, blk_num)
                           : tpsap == TETRA_TPSAP_T_NDB ? pieces_for(TETRA_TRAIN_NORM_2, tpsap, blk_num)
                           : tpsap == TETRA_TPSAP_T_BBK ? pieces_for(TETRA_TRAIN_NORM_1, tpsap, blk_num)
                                                        : pieces_for(TETRA_TRAIN_SYNC, tpsap, blk_num);
    if (longest.len0 == 0) return TETRA_ERR_ARG;                      // no burst type carries (tpsap, blk_num)
    if (row_stride < longest.len0 + longest.len1) return TETRA_ERR_SIZE;
    if ((row_stride & 3) || ((uintptr_t)d_rows & 3)) return TETRA_ERR_ALIGN;
    if (PACKED && ((uintptr_t)d_frames & 3)) return TETRA_ERR_ALIGN;
    const bool wide = !(row_stride & 7) && !((uintptr_t)d_rows & 7);
    if (demux_core::use_rows_kernel(PACKED, wide, row_stride)) {           // whole rows per wavefront (k_demux_rows)
        const int row_u = row_stride >> 3, rpw = 64 / row_u;
        hipLaunchKernelGGL((k_demux_rows<false>), dim3((unsigned)demux_core::rows_grid(n, row_stride)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_frames,
                           d_frame_type, nullptr, nullptr, n, lut_for(tpsap, blk_num), row_u, rpw, (unsigned)((65536 + row_u - 1) / row_u), d_rows, d_valid);
        return hipGetLastError() == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
    }
    const dim3 grid((unsigned)demux_core::units_grid(n, row_stride, wide));
    if (wide) hipLaunchKernelGGL((k_burst_demux<PACKED, true>), grid, dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_frames, d_frame_type, n,
                                 tpsap, blk_num, d_rows, row_stride, d_valid);
    else hipLaunchKernelGGL((k_burst_demux<PACKED, false>), grid, dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_frames, d_frame_type, n,
                            tpsap, blk_num, d_rows, row_stride, d_valid);
    return hipGetLastError() == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
}

template <bool PACKED> int demux_compact_launch(const void* d_frames_v, const int32_t* d_frame_type, int n, int tpsap, int blk_num,
                                                uint8_t* d_rows, int row_stride, int32_t* d_row_frame, int32_t* d_n_rows, void* hip_stream) {
    const uint8_t* d_frames = static_cast<const uint8_t*>(d_frames_v);
    if (!d_frames || !d_frame_type || !d_rows || !d_row_frame || !d_n_rows || n < 0) return TETRA_ERR_ARG;
    if (tpsap < 0 || tpsap > 5) return TETRA_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (n == 0) return hipMemsetAsync(d_n_rows, 0, sizeof(int32_t), s) == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
    // same argument rules as tetra_burst_demux_device
    const Pieces any[3] = { pieces_for(TETRA_TRAIN_SYNC, tpsap, blk_num), pieces_for(TETRA_TRAIN_NORM_1, tpsap, blk_num),
                            pieces_for(TETRA_TRAIN_NORM_2, tpsap, blk_num) };
    int longest = 0;
    for (const Pieces& p : any) longest = p.len0 + p.len1 > longest ? p.len0 + p.len1 : longest;
    if (longest == 0) return TETRA_ERR_ARG;                      // no burst type carries this (kind, block number)
    if (row_stride < longest) return TETRA_ERR_SIZE;
    if ((row_stride & 3) || ((uintptr_t)d_rows & 3)) return TETRA_ERR_ALIGN;
    if (PACKED && ((uintptr_t)d_frames & 3)) return TETRA_ERR_ALIGN;          // packed frames are read as 32-bit words (as demux_launch<true> checks)
    const int nblocks = (n + 255) / 256;
    // the per-block counts / offsets of steps 1-3 live in the head of the row buffer itself (4 bytes per 256 frames of a buffer that
    // holds at least 30 bytes per frame; 4-byte aligned): the gather, which overwrites it, runs after their last reader in stream
    // order.  (Until round 5 a stream-ordered allocation per call: every so often the pool gave its memory back in between and one
    // call took milliseconds.)
    int* off = reinterpret_cast<int*>(d_rows);
    hipLaunchKernelGGL(k_demux_count, dim3(nblocks), dim3(256), 0, s, d_frame_type, n, tpsap, blk_num, off);
    hipLaunchKernelGGL(k_demux_scan, dim3(1), dim3(1024), 0, s, off, nblocks, d_n_rows);
    hipLaunchKernelGGL(k_demux_index, dim3(nblocks), dim3(256), 0, s, d_frame_type, n, tpsap, blk_num, off, d_row_frame);
    const bool wide = !(row_stride & 7) && !((uintptr_t)d_rows & 7);
    const dim3 grid((unsigned)demux_core::units_grid(n, row_stride, wide));
    if (demux_core::use_rows_kernel(PACKED, wide, row_stride)) {           // whole rows per wavefront (k_demux_rows), sized for the worst case of n rows
        const int row_u = row_stride >> 3, rpw = 64 / row_u;
        hipLaunchKernelGGL((k_demux_rows<true>), dim3((unsigned)demux_core::rows_grid(n, row_stride)), dim3(256), 0, s, d_frames, d_frame_type, d_row_frame, d_n_rows, n,
                           lut_for(tpsap, blk_num), row_u, rpw, (unsigned)((65536 + row_u - 1) / row_u), d_rows, nullptr);
    } else if (wide) hipLaunchKernelGGL((k_demux_gather<PACKED, true>), grid, dim3(256), 0, s, d_frames, d_frame_type, d_row_frame, d_n_rows, n, tpsap,
                                 blk_num, d_rows, row_stride);
    else hipLaunchKernelGGL((k_demux_gather<PACKED, false>), grid, dim3(256), 0, s, d_frames, d_frame_type, d_row_frame, d_n_rows, n, tpsap,
                            blk_num, d_rows, row_stride);
    return hipGetLastError() == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
}
}  // namespace

extern "C" {

int tetra_burst_demux_device(const uint8_t* d_frames, const int32_t* d_frame_type, int n, int tpsap, int blk_num, uint8_t* d_rows,
                             int row_stride, int32_t* d_valid, void* hip_stream) {
    return demux_launch<false>(d_frames, d_frame_type, n, tpsap, blk_num, d_rows, row_stride, d_valid, hip_stream);
}
int tetra_burst_demux_packed_device(const uint32_t* d_frames_packed, const int32_t* d_frame_type, int n, int tpsap, int blk_num,
                                    uint8_t* d_rows, int row_stride, int32_t* d_valid, void* hip_stream) {
    return demux_launch<true>(d_frames_packed, d_frame_type, n, tpsap, blk_num, d_rows, row_stride, d_valid, hip_stream);
}
int tetra_burst_demux_compact_device(const uint8_t* d_frames, const int32_t* d_frame_type, int n, int tpsap, int blk_num,
                                     uint8_t* d_rows, int row_stride, int32_t* d_row_frame, int32_t* d_n_rows, void* hip_stream) {
    return demux_compact_launch<false>(d_frames, d_frame_type, n, tpsap, blk_num, d_rows, row_stride, d_row_frame, d_n_rows, hip_stream);
}
int tetra_burst_demux_compact_packed_device(const uint32_t* d_frames_packed, const int32_t* d_frame_type, int n, int tpsap, int blk_num,
                                            uint8_t* d_rows, int row_stride, int32_t* d_row_frame, int32_t* d_n_rows, void* hip_stream) {
    return demux_compact_launch<true>(d_frames_packed, d_frame_type, n, tpsap, blk_num, d_rows, row_stride, d_row_frame, d_n_rows, hip_stream);
}

int tetra_burst_index_device(const int32_t* d_frame_type, int n, int frames_per_channel, int32_t* d_lists, int32_t* d_counts,
                             int32_t* d_chan_first, int32_t* d_work, void* hip_stream) {
    if (!d_counts || n < 0) return TETRA_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (n == 0) return hipMemsetAsync(d_counts, 0, sizeof(int32_t) * TETRA_N_LISTS, s) == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;      // (no frames: nothing else is touched)
    if (!d_frame_type || !d_lists || !d_work) return TETRA_ERR_ARG;
    if (d_chan_first && (frames_per_channel < 1 || n % frames_per_channel)) return TETRA_ERR_ARG;
    const int nblocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_index_count, dim3(nblocks), dim3(256), 0, s, d_frame_type, n, nblocks, d_work);
    hipLaunchKernelGGL(k_index_scan, dim3(1), dim3(kScanThreads), 0, s, d_work, nblocks, d_counts);
    hipLaunchKernelGGL(k_index_write, dim3(nblocks), dim3(256), 0, s, d_frame_type, n, nblocks, d_chan_first ? frames_per_channel : 1, d_work,
                       d_lists, d_chan_first);
    return hipGetLastError() == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
}

}  // extern "C"
