// tetra_lmac.hip -- batched lower-MAC channel decoding (include/tetra_lmac.h), bit-exact with the reference's
// tp_sap_udata_ind() decoding chain (src/decoder/src/lower_mac/tetra_lower_mac.c:181-236).
//
// One 64-lane workgroup (one wavefront) decodes 64 blocks, one block per lane (lane-level code: lmac_core.hpp):
//   1. front end -> the descrambled type-4 bits of the lane's block as packed words in LDS, [word][lane]:
//      * from PACKED FRAMES (k_lmac_frames, round 6): the lane reads its frame (four 16-byte loads), cuts the kind's one or two bit
//        ranges out with funnel shifts and XORs whole words of its scrambling sequence (linear in the code: four rows of a 64 KB
//        table indexed by the code's bytes);
//      * from byte rows of plain bits (k_lmac_decode; every byte 0 / 1): the workgroup reads its 64 rows as one contiguous run,
//        8 bytes per lane, packs them to bits through LDS, each lane takes its row's words, then the same;
//      * from byte rows with any other byte value anywhere in the workgroup's 64 rows (erasures), or rows that are not 8-byte
//        aligned: the byte route -- rows staged through LDS in coalesced 64-bit chunks, an LFSR step and a three-way classification
//        (0 / erasure 0xff / 1) per byte, soft classes 2 bits per type-4 bit in LDS;
//   2. forward recursion: 16 path metrics in 8 registers (packed int16: states i and i + 8), the three soft values of a step pair
//      gathered from LDS at the deinterleaved positions (wave-uniform addresses), a butterfly = add, subtract, maximum, difference
//      with the halves picked by operand modifiers (no moves), the 2 x 16 decision bits of a step pair gathered with byte permutes
//      and stored as one dword to a global scratch laid out [workgroup][step pair][lane] (one 256-byte line per pair; written
//      once, read once, normally from L2 / MALL): 99 vector instructions per step pair;
//   3. traceback from the scratch (the addresses do not depend on the surviving state, only the bit picked does, so the loads
//      pipeline) with the CRC16 folded in (affine in the message: one AND + XOR per bit with a wave-uniform constant), decoded bits
//      packed 16 per ushort into LDS [half][lane];
//   4. the 64 decoded rows are written back with coalesced 8-byte stores, 8 bits -> 8 bytes per lane.
// The work is integer add / compare / select: bound by vector issue.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "../../include/tetra_lmac.h"
#include "demux_core.hpp"
#include "lmac_core.hpp"

namespace {

using namespace tetra_lmac;

constexpr int kLanes = 64;
constexpr int kChunkDwords = 16;                       // 64 type-5 bits per row per staging chunk
constexpr int kSteps = kMaxType2 + kFlush;             // 292
constexpr int kClsWords = (kMaxType345 + 15) / 16;     // 27
constexpr int kOutHalves = kMaxType2 / 16;             // 18
constexpr int kOutPad = 2;                             // outw rows of 66 ushorts = 33 banks: the write-back's column reads do not collide
bool g_force_byte_route = false;                       // tests / A-B: tetra_lmac_debug_force_byte_route
__constant__ CrcInvTable kCrcInvDev = make_crc_inv_table();      // the traceback's backward CRC table; every workgroup copies it to LDS

struct BlkParam { int type345, type2, type1, a, crc; };
// tetra_blk_param[], tetra_lower_mac.c:58-105 (values of EN 300 392-2 table 8.x / 8.2.4.1)
const BlkParam kBlk[6] = {
    { 120, 80, 60, 11, 1 },     // SB1
    { 216, 144, 124, 101, 1 },  // SB2
    { 216, 144, 124, 101, 1 },  // NDB
    { 30, 30, 14, 0, 0 },       // BBK
    { 168, 112, 92, 13, 1 },    // SCH/HU
    { 432, 288, 268, 103, 1 },  // SCH/F
};

typedef uint16_t OutW[kOutHalves][kLanes + kOutPad];

// steps 2-3 for a workgroup whose type-4 bits (BITS) or soft classes are in `cls`; dec_st(u, word) / dec_ld(u) = the lane's decision
// word of step pair u (global scratch, or LDS for a launch of short blocks); returns the lane's CRC verdict
template <bool BITS, class DecSt, class DecLd>
__device__ __forceinline__ bool decode_core(const uint32_t (*cls)[kLanes], OutW& outw, const uint32_t* crc_inv, int lane, int type345, int type2,
                                            int a, DecSt dec_st, DecLd dec_ld) {
    int pos = a;                       // (a * i) % K for i = 1
    auto fetch = [&] {
        Raw3 r;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int p = interleave_next(pos, a, type345);
            r.w[k] = cls[BITS ? p >> 5 : p >> 4][lane];
            r.at[k] = BITS ? 31u - (uint32_t)(p & 31) : (uint32_t)(30 - 2 * (p & 15));
        }
        return r;
    };
    if (BITS) {
        viterbi_forward(type2, fetch,
                        [&](const Raw3& r) { return bm_from_masks(bfe_mask(r.w[0], r.at[0]), bfe_mask(r.w[1], r.at[1]), bfe_mask(r.w[2], r.at[2])); },
                        dec_st);
    } else {
        viterbi_forward(type2, fetch,
                        [&](const Raw3& r) { return bm_from_classes((int)(r.w[0] << r.at[0]) >> 30, (int)(r.w[1] << r.at[1]) >> 30, (int)(r.w[2] << r.at[2]) >> 30); },
                        dec_st);
    }
    // traceback + CRC (own lane's data only: program order is enough)
    return viterbi_traceback(type2, dec_ld, [&](int h, uint32_t half) { outw[h][lane] = (uint16_t)half; },
                             [&](uint32_t off) { return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(crc_inv) + off); });
}

// the backward CRC table into LDS (4 entries per lane); the caller's next barrier makes it visible
__device__ __forceinline__ void load_crc_inv(uint32_t* crc_inv, int lane) {
#pragma unroll
    for (int k = 0; k < 256 / kLanes; ++k) crc_inv[k * kLanes + lane] = kCrcInvDev.t[k * kLanes + lane];
}

// step 4: decoded rows -> HBM.  WIDE: rows a multiple of 8 bytes and 8-byte aligned -- 8 bits -> 8 bytes per lane, the (row, unit)
// index space flattened so that every lane stores in every round (q = i / units by a multiply: exact for i < 64 * 36).
__device__ __forceinline__ void write_rows(const OutW& outw, int lane, int rows_here, int type2, uint8_t* __restrict__ out0, int out_stride) {
    if (!(out_stride & 7) && !((uintptr_t)out0 & 7)) {
        const int units = type2 >> 3, total = rows_here * units;
        const uint32_t inv = ((1u << 20) + (uint32_t)units - 1u) / (uint32_t)units;
        for (int i = lane; i < total; i += kLanes) {
            const int q = (int)(((uint32_t)i * inv) >> 20), d = i - q * units;
            const uint32_t byte = ((uint32_t)outw[d >> 1][q] >> (8 * (d & 1))) & 0xffu;
            demux_core::U2 v;
            v.x = spread4(byte & 0xfu);
            v.y = spread4(byte >> 4);
            reinterpret_cast<demux_core::U2*>(out0 + (size_t)q * out_stride)[d] = v;
        }
    } else {
        const int out_dw = type2 >> 2;
        for (int q = 0; q < rows_here; ++q) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(out0 + (size_t)q * out_stride);
            for (int d = lane; d < out_dw; d += kLanes) dst[d] = spread4(((uint32_t)outw[d >> 2][q] >> (4 * (d & 3))) & 0xfu);
        }
    }
}

// the lane's scrambling sequence words XORed onto its packed row, result to LDS (sequence rows as 16-byte loads)
__device__ __forceinline__ void descramble_to_lds(int type345, uint32_t code, const uint32_t xb[kSeqWords], const uint32_t* __restrict__ seq_tab,
                                                  uint32_t (*cls)[kLanes], int lane) {
    const uint4* r0 = reinterpret_cast<const uint4*>(seq_tab + ((size_t)0 * 256 + (code & 0xffu)) * kSeqStride);
    const uint4* r1 = reinterpret_cast<const uint4*>(seq_tab + ((size_t)1 * 256 + ((code >> 8) & 0xffu)) * kSeqStride);
    const uint4* r2 = reinterpret_cast<const uint4*>(seq_tab + ((size_t)2 * 256 + ((code >> 16) & 0xffu)) * kSeqStride);
    const uint4* r3 = reinterpret_cast<const uint4*>(seq_tab + ((size_t)3 * 256 + (code >> 24)) * kSeqStride);
#pragma unroll
    for (int g = 0; g < (kSeqWords + 3) / 4; ++g) {
        if (128 * g < type345) {
            const uint4 s0 = r0[g], s1 = r1[g], s2 = r2[g], s3 = r3[g];
            const uint32_t w[4] = { s0.x ^ s1.x ^ s2.x ^ s3.x, s0.y ^ s1.y ^ s2.y ^ s3.y, s0.z ^ s1.z ^ s2.z ^ s3.z, s0.w ^ s1.w ^ s2.w ^ s3.w };
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * g + k < kSeqWords && 32 * (4 * g + k) < type345) cls[4 * g + k][lane] = xb[4 * g + k] ^ w[k];
        }
    }
}

__global__ __launch_bounds__(kLanes) void k_lmac_decode(const uint8_t* __restrict__ type5, int n_blocks, int in_stride,
                                                        const uint32_t* __restrict__ scramb_init, int fixed_init,
                                                        int type345, int type2, int type1, int a,
                                                        uint8_t* __restrict__ out, int out_stride, int* __restrict__ crc_ok,
                                                        uint32_t* __restrict__ dec_scratch, int dec_pairs,
                                                        const int* __restrict__ n_blocks_dev, const int* __restrict__ init_index,
                                                        const uint32_t* __restrict__ seq_tab) {
    __shared__ uint32_t stage[kLanes][kChunkDwords + 1];     // +1: odd row stride, conflict-free column reads
    __shared__ uint32_t cls[kClsWords + 1][kLanes];
    __shared__ OutW outw;
    __shared__ uint32_t crc_inv[256];
    const int lane = threadIdx.x;
    const int blk0 = blockIdx.x * kLanes;
    const int blk = blk0 + lane;
    if (n_blocks_dev) {           // counted form: the number of rows is a device-side result (compacting demultiplexer)
        const int have = *n_blocks_dev;
        n_blocks = have < n_blocks ? have : n_blocks;
        if (blk0 >= n_blocks) return;
    }
    const int rows_here = min(kLanes, n_blocks - blk0);
    const uint32_t code = (fixed_init || blk >= n_blocks) ? kScrambInitSb1 : scramb_init[init_index ? init_index[blk] : blk];
    uint32_t* dec = dec_scratch + (size_t)blockIdx.x * dec_pairs * kLanes + lane;
    load_crc_inv(crc_inv, lane);

    // 1. front end.  Rows of plain bits: each lane packs its own row (8-byte loads), descrambles whole words and leaves the type-4
    //    bits in LDS (cls rows 0..13 as [word][lane]); the workgroup falls back to the byte route if any of its rows holds another
    //    byte value, or if the rows are not 8-byte aligned.
    bool byte_route = seq_tab == nullptr || (in_stride & 7) || ((uintptr_t)type5 & 7) || in_stride > 512;      // (512: the unit -> row map below)
    if (!byte_route) {
        // The workgroup's rows_here rows are one contiguous run of rows_here * in_stride bytes: read it ONCE, 8 bytes per lane and
        // 512 contiguous bytes per load instruction, pack each unit's 8 bytes to 8 bits and drop them as one byte into the row's
        // packed words in LDS (`stage`, 17 words per row); then every lane picks up its own row's words.  (Until late in round 6 every
        // lane read its own row with strided 8-byte loads: 64 cache lines per instruction, and with a few waves per CU the lines were
        // evicted before their other 120 bytes were used.)
        const int units_per_row = in_stride >> 3, total_units = rows_here * units_per_row;
        const uint32_t inv = ((1u << 20) + (uint32_t)units_per_row - 1u) / (uint32_t)units_per_row;      // exact for u < 64 * 64
        const U2* base = reinterpret_cast<const U2*>(type5 + (size_t)blk0 * in_stride);
        uint8_t* sb = reinterpret_cast<uint8_t*>(&stage[0][0]);
        uint32_t dirty = 0;
        for (int u = lane; u < total_units; u += kLanes) {
            const int r = (int)(((uint32_t)u * inv) >> 20), j = u - r * units_per_row;      // row, byte of its packed bits
            if (8 * j < type345) {
                const U2 d = base[u];
                dirty |= (d.x | d.y) & 0xfefefefeu;
                // type-5 bits 8j .. 8j+7, first bit most significant; byte j of the row's bit string sits in word j / 4 at bits 31 - 8 (j % 4) ..
                sb[(size_t)r * (4 * (kChunkDwords + 1)) + (j & ~3) + (3 - (j & 3))] = (uint8_t)((pack4(d.x) << 4) | pack4(d.y));
            }
        }
        byte_route = __builtin_amdgcn_ballot_w64(dirty != 0) != 0;          // wave-uniform
        __syncthreads();
        if (!byte_route) {
            uint32_t xb[kSeqWords];
#pragma unroll
            for (int w = 0; w < kSeqWords; ++w) xb[w] = 32 * w < type345 ? stage[lane][w] : 0u;
            if (type345 & 31) xb[type345 >> 5] &= ~(0xffffffffu >> (type345 & 31));      // (bytes behind the row's last bit were never written)
            descramble_to_lds(type345, code, xb, seq_tab, cls, lane);
        }
    }
    bool good;
    if (!byte_route) {
        __syncthreads();
        good = decode_core<true>(cls, outw, crc_inv, lane, type345, type2, a, [&](int u, uint32_t w) { dec[u * kLanes] = w; },
                                 [&](int u) { return dec[u * kLanes]; });
    } else {
        // rows -> LDS in chunks of 64 bits per row (coalesced 64-byte segments, 4 rows per load instruction), each lane
        // descrambles its own row chunk by chunk (its LFSR carried in a register) and packs the soft classes
        uint32_t lfsr = code;
        const int row_dw = type345 >> 2;
        for (int c0 = 0; c0 < row_dw; c0 += kChunkDwords) {
#pragma unroll 4
            for (int it = 0; it < kLanes * kChunkDwords / kLanes; ++it) {
                const int q = it * (kLanes / kChunkDwords) + lane / kChunkDwords, d = lane % kChunkDwords;
                uint32_t v = 0;
                if (q < rows_here && c0 + d < row_dw)
                    v = reinterpret_cast<const uint32_t*>(type5 + (size_t)(blk0 + q) * in_stride)[c0 + d];
                stage[q][d] = v;
            }
            __syncthreads();
            lfsr = descramble_chunk(type345 - 4 * c0, lfsr, [&](int d) { return stage[lane][d]; },
                                    [&](int w, uint32_t word) { cls[c0 / 4 + w][lane] = word; });
            __syncthreads();
        }
        good = decode_core<false>(cls, outw, crc_inv, lane, type345, type2, a, [&](int u, uint32_t w) { dec[u * kLanes] = w; },
                                  [&](int u) { return dec[u * kLanes]; });
    }
    if (blk < n_blocks) crc_ok[blk] = good;
    __syncthreads();
    write_rows(outw, lane, rows_here, type2, out + (size_t)blk0 * out_stride, out_stride);
}

// ---- straight from packed frames, several kinds per launch (tetra_lmac_decode_frames_device) ----------------------------------
struct DevJob {
    const int* row_frame;
    const int* n_rows_dev;
    const uint32_t* frame_scramb;
    uint8_t* out;
    int* crc_ok;
    tetra_lmac_label_t* labels;
    long long scratch_base;            // first word of the job's decision scratch
    int n_rows, out_stride, first_group, dec_pairs;
    int layout;                        // kLayout*
    int type345, type2, a;
};
struct DevFrames {
    const uint32_t* frames;
    const int* frame_type;
    const uint32_t* bitnum;
    const uint32_t* time_rx;
    const uint32_t* time;
    int frames_per_channel;
    int n_frames;
};
struct JobTable {
    DevJob job[TETRA_LMAC_MAX_JOBS];
    DevFrames src;
    int n;
};
// LDS per workgroup: 3584 (type-4 bits; the decoded halves reuse the space once the forward recursion is through with them) + 1024
// (backward CRC table) = 4608 B <= 5120: LDS never caps the kernel below 8 waves per SIMD -- which matters beside the demodulator:
// the compiler sizes a kernel's register allocation for the occupancy its LDS allows (6992 B -> 6 waves -> 80 registers where 54 are
// used), and next to k_fused's 199-register waves every 8 registers decide how many of this kernel's waves fit a SIMD.
__global__ __launch_bounds__(kLanes) void k_lmac_frames(const JobTable tab, uint32_t* __restrict__ dec_scratch,
                                                        const uint32_t* __restrict__ seq_tab) {
    __shared__ union {
        uint32_t cls[kSeqWords][kLanes];
        OutW outw;
    } sm;
    static_assert(sizeof(OutW) <= sizeof(uint32_t) * kSeqWords * kLanes, "the decoded halves fit the type-4 words' space");
    uint32_t (&cls)[kSeqWords][kLanes] = sm.cls;
    OutW& outw = sm.outw;
    __shared__ uint32_t crc_inv[256];
    const int lane = threadIdx.x;
    int ji = 0;
    for (int i = 1; i < tab.n; ++i) ji = (int)blockIdx.x >= tab.job[i].first_group ? i : ji;
    const DevJob& J = tab.job[ji];
    const int group = (int)blockIdx.x - J.first_group;
    const int blk0 = group * kLanes, blk = blk0 + lane;
    int n_blocks = J.n_rows;
    if (J.n_rows_dev) {
        const int have = *J.n_rows_dev;
        n_blocks = have < n_blocks ? have : n_blocks;
    }
    if (blk0 >= n_blocks) return;
    const int rows_here = min(kLanes, n_blocks - blk0);
    // (a list entry outside [0, n_frames) -- a caller's slip -- reads the nearest frame instead of memory that is not there)
    const int f = min(max(J.row_frame[blk < n_blocks ? blk : blk0], 0), tab.src.n_frames - 1);
    const uint32_t code = J.frame_scramb ? J.frame_scramb[f] : kScrambInitSb1;
    const int ft = tab.src.frame_type[f];
    uint32_t fw[kFrameWords];
    {
        const uint4* src = reinterpret_cast<const uint4*>(tab.src.frames + (size_t)f * kFrameWords);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint4 v = src[g];
            fw[4 * g] = v.x; fw[4 * g + 1] = v.y; fw[4 * g + 2] = v.z; fw[4 * g + 3] = v.w;
        }
    }
    bool good = true;
    if (J.layout == kLayoutBbk) {
        // TPSAP_T_BBK: the reference only descrambles (tetra_lower_mac.c:231-236): 30 bits -> 30 bytes (+ 2 zero bytes), a row per lane
        const uint32_t x = bbk_bits(fw, ft);
        const uint32_t seq = seq_tab[((size_t)0 * 256 + (code & 0xffu)) * kSeqStride] ^ seq_tab[((size_t)1 * 256 + ((code >> 8) & 0xffu)) * kSeqStride] ^
                             seq_tab[((size_t)2 * 256 + ((code >> 16) & 0xffu)) * kSeqStride] ^ seq_tab[((size_t)3 * 256 + (code >> 24)) * kSeqStride];
        const uint32_t y = (x ^ seq) & 0xfffffffcu;          // 30 bits, first bit most significant
        if (blk < n_blocks) {
            demux_core::U2* dst = reinterpret_cast<demux_core::U2*>(J.out + (size_t)blk * J.out_stride);
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[k] = demux_core::U2{ bbk_bytes(y, 2 * k), bbk_bytes(y, 2 * k + 1) };
        }
    } else {
        uint32_t xb[kSeqWords];
        frame_block(J.layout, fw, ft, xb);
        descramble_to_lds(J.type345, code, xb, seq_tab, cls, lane);
        load_crc_inv(crc_inv, lane);
        __syncthreads();
        uint32_t* dec = dec_scratch + J.scratch_base + (size_t)group * J.dec_pairs * kLanes + lane;
        good = decode_core<true>(cls, outw, crc_inv, lane, J.type345, J.type2, J.a, [&](int u, uint32_t w) { dec[u * kLanes] = w; },
                                 [&](int u) { return dec[u * kLanes]; });
    }
    if (blk < n_blocks) {
        J.crc_ok[blk] = good;
        if (J.labels) {
            tetra_lmac_label_t lb;
            lb.channel = f / tab.src.frames_per_channel;
            lb.frame_slot = f - lb.channel * tab.src.frames_per_channel;
            lb.bitnum = tab.src.bitnum[f];
            lb.tdma_time_rx = tab.src.time_rx[f];
            lb.tdma_time = tab.src.time[f];
            lb.crc_ok = good;
            J.labels[blk] = lb;
        }
    }
    if (J.layout != kLayoutBbk) {
        __syncthreads();
        write_rows(outw, lane, rows_here, J.type2, J.out + (size_t)blk0 * J.out_stride, J.out_stride);
    }
}

// Per-device constant of the decoder's packed route: the scrambling-sequence table (64 KB), built on the host once per device and
// kept for the life of the process.
std::mutex g_tab_mu;
uint32_t* g_seq_tab[64] = {};
// nullptr if the table cannot be set up (out of memory): the caller reports TETRA_ERR_NOMEM
const uint32_t* seq_table() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(g_tab_mu);
    if (!g_seq_tab[dev]) {
        const size_t seq_words = (size_t)4 * 256 * kSeqStride;
        std::vector<uint32_t> host(seq_words);
        scramb_sequence_table(host.data());
        uint32_t* d_seq = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&d_seq), sizeof(uint32_t) * seq_words) != hipSuccess ||
            hipMemcpy(d_seq, host.data(), sizeof(uint32_t) * seq_words, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            if (d_seq) (void)hipFree(d_seq);
            return nullptr;
        }
        g_seq_tab[dev] = d_seq;
    }
    return g_seq_tab[dev];
}

// The decoder's decision scratch (up to 200 MB for a second of 4096 channels' SCH/F slots) comes from a stream-ordered pool of this
// library's own, one per device, that KEEPS what is freed into it (release threshold = everything).  The device's default pool hands
// unused memory back to the driver at synchronisation points; the next call then maps 200 MB again and takes milliseconds instead of
// microseconds -- seen as one call in five at 20 ms in the two-stream chain (profiles/r05/README.md).
std::mutex g_pool_mu;
hipMemPool_t g_pool[64] = {};
hipMemPool_t scratch_pool() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (!g_pool[dev]) {
        hipMemPoolProps props = {};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t p = nullptr;
        if (hipMemPoolCreate(&p, &props) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        uint64_t keep = ~0ull;
        (void)hipMemPoolSetAttribute(p, hipMemPoolAttrReleaseThreshold, &keep);
        g_pool[dev] = p;
    }
    return g_pool[dev];
}

// TPSAP_T_BBK: the reference only descrambles (tetra_lower_mac.c:231-236); 30 bits per block, one lane per block.
__global__ __launch_bounds__(256) void k_lmac_bbk(const uint8_t* __restrict__ type5, int n_blocks, int in_stride,
                                                  const uint32_t* __restrict__ scramb_init, int nbits,
                                                  uint8_t* __restrict__ out, int out_stride, int* __restrict__ crc_ok,
                                                  const int* __restrict__ n_blocks_dev, const int* __restrict__ init_index) {
    const int blk = blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= n_blocks || (n_blocks_dev && blk >= *n_blocks_dev)) return;
    uint32_t lfsr = scramb_init[init_index ? init_index[blk] : blk];
    const uint8_t* src = type5 + (size_t)blk * in_stride;
    uint8_t* dst = out + (size_t)blk * out_stride;
    for (int j = 0; j < nbits; ++j) dst[j] = src[j] ^ (uint8_t)lfsr_next(lfsr);
    crc_ok[blk] = 1;
}

// tetra_lower_mac.c:258-266 per channel: walk the frame slots in time order, a good SB1 replaces the scrambling code
__global__ __launch_bounds__(256) void k_track_scramb(const uint8_t* __restrict__ sb1, int stride, const int* __restrict__ crc_ok,
                                                      const int* __restrict__ valid, int n_channels, int frames,
                                                      uint32_t* __restrict__ chan_scramb, uint32_t* __restrict__ row_scramb) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_channels) return;
    uint32_t cur = chan_scramb[c];
    for (int f = 0; f < frames; ++f) {
        const size_t r = (size_t)c * frames + f;
        if (valid[r] && crc_ok[r]) {
            const uint8_t* t2 = sb1 + r * stride;
            auto field = [&](int first, int len) { uint32_t v = 0; for (int i = 0; i < len; ++i) v = (v << 1) | (t2[first + i] & 1u); return v; };
            const uint32_t cc = field(4, 6), mcc = field(31, 10), mnc = field(41, 14);
            cur = (((cc & 0x3f) | ((mnc & 0x3fff) << 6) | ((mcc & 0x3ff) << 20)) << 2) | kScrambInitSb1;   // tetra_scramb.c:87-99
        }
        row_scramb[r] = cur;
    }
    chan_scramb[c] = cur;
}

// The whole SYNC-PDU read-out of tp_sap_udata_ind's SB1 case (tetra_lower_mac.c:246-275) plus the PHY's TDMA clock
// (tetra_burst_sync.c:113, tetra_tdma.c:28-78), per channel, frame slots in time order:
//   every frame the LOCKED receiver consumes      t_phy_state.time += one timeslot (tetra_tdma_time_add_tn, before the callback)
//   a SYNC burst's SB1 block, good CRC            tcd-> colour code, time (tn = bits + 1, fn, mn), mcc, mnc, scramb_init
//   a SYNC burst's SB1 block, any CRC             t_phy_state.time = tcd->time   (:268-269: copied whatever the CRC said)
// Outputs per frame slot: the scrambling code in force for the slot's other blocks, the TDMA time tetra_burst_rx_cb sees on
// entry (t_display_st->curr_multiframe / curr_frame, tetra_burst.c:349-350) and the time after the slot's SB1 (what every
// later block of the burst and the next slot's increment start from).  Times are packed tn | fn << 8 | mn << 16.
__device__ __forceinline__ void tdma_add_tn(uint32_t& tn, uint32_t& fn, uint32_t& mn) {
    tn += 1;                                                  // tetra_tdma_time_add_tn(tm, 1) -> normalize_tn -> _fn -> _mn
    if (tn > 4) { const uint32_t d = tn / 4; tn = tn % 4; fn += d; }
    if (fn > 18) { const uint32_t d = fn / 18; fn = fn % 18; mn += d; }
    if (mn > 60) mn = mn % 60;
}
__global__ __launch_bounds__(256) void k_track_sync(const uint8_t* __restrict__ sb1, int stride, const int* __restrict__ crc_ok,
                                                    const int* __restrict__ valid, const int* __restrict__ n_frames, int n_channels,
                                                    int frames, tetra_lmac_cell_state_t* __restrict__ cell,
                                                    uint32_t* __restrict__ row_scramb, uint32_t* __restrict__ row_time_rx,
                                                    uint32_t* __restrict__ row_time) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_channels) return;
    tetra_lmac_cell_state_t st = cell[c];
    const int nf = n_frames ? (n_frames[c] < frames ? n_frames[c] : frames) : frames;
    for (int f = 0; f < frames; ++f) {
        const size_t r = (size_t)c * frames + f;
        if (f < nf) {
            tdma_add_tn(st.phy_tn, st.phy_fn, st.phy_mn);
            if (row_time_rx) row_time_rx[r] = st.phy_tn | (st.phy_fn << 8) | (st.phy_mn << 16);
            if (valid[r]) {
                if (crc_ok[r]) {
                    const uint8_t* t2 = sb1 + r * stride;
                    auto field = [&](int first, int len) { uint32_t v = 0; for (int i = 0; i < len; ++i) v = (v << 1) | (t2[first + i] & 1u); return v; };
                    st.colour_code = field(4, 6);
                    st.tcd_tn = field(10, 2) + 1;
                    st.tcd_fn = field(12, 5);
                    st.tcd_mn = field(17, 6);
                    st.mcc = field(31, 10);
                    st.mnc = field(41, 14);
                    st.scramb_init = (((st.colour_code & 0x3f) | ((st.mnc & 0x3fff) << 6) | ((st.mcc & 0x3ff) << 20)) << 2) | kScrambInitSb1;
                }
                st.phy_tn = st.tcd_tn; st.phy_fn = st.tcd_fn; st.phy_mn = st.tcd_mn;
            }
        } else if (row_time_rx) {
            row_time_rx[r] = 0;
        }
        row_scramb[r] = st.scramb_init;
        if (row_time) row_time[r] = f < nf ? (st.phy_tn | (st.phy_fn << 8) | (st.phy_mn << 16)) : 0u;
    }
    cell[c] = st;
}

// tetra_lmac_track_sync_lists_device: k_track_sync's walk with the channel's SB1 rows compact, WITHOUT the walk.  One wavefront per
// channel, 64 frame slots at a time, a slot per lane.  What a slot needs from its past is (1) the last SYNC frame before it -- the
// PHY clock was set to tcd's time there -- and how many slots ago that was, (2) the last SYNC frame with a good CRC up to there /
// up to the slot itself -- that is what tcd holds.  Both are "highest set bit below my lane" of two ballots; the fields travel
// with two lane shuffles; and the clock k slots after a reset is one literal tetra_tdma_time_add_tn step (tetra_tdma.c:28-78: any
// state lands in tn 0..4, fn 0..18, mn 0..60 -- the wrap tests run on every call) followed by k - 1 steps in closed form
// (three counters that run 1..4, 1..18, 1..60, a zero taking one step to become 1).  Nothing is serial but the carry from one
// 64-slot group to the next.  (The first version walked the slots on the scalar unit: 2577 scalar instructions per wave, 29 us.)
struct Tdma { uint32_t tn, fn, mn; };
__device__ __forceinline__ Tdma tdma_advance(Tdma t, uint32_t k) {       // k >= 1 calls of tetra_tdma_time_add_tn(t, 1)
    tdma_add_tn(t.tn, t.fn, t.mn);
    k -= 1;
    if (k) {
        const uint32_t p1 = t.tn + k - 1u;                                // t.tn in 0..4: a zero needs one step to become 1
        const uint32_t c1 = p1 >> 2;
        t.tn = (p1 & 3u) + 1u;
        if (c1) {
            const uint32_t p2 = t.fn + c1 - 1u, c2 = p2 / 18u;
            t.fn = p2 - 18u * c2 + 1u;
            if (c2) t.mn = (t.mn + c2 - 1u) % 60u + 1u;
        }
    }
    return t;
}
__device__ __forceinline__ uint32_t tdma_pack(Tdma t) { return t.tn | (t.fn << 8) | (t.mn << 16); }
// highest set bit of m at or below position `upto` (-1: none; upto = -1: none)
__device__ __forceinline__ int last_set_upto(unsigned long long m, int upto) {
    if (upto < 0) return -1;
    const unsigned long long x = m & (upto >= 63 ? ~0ull : ((2ull << upto) - 1ull));
    return x ? 63 - __clzll((long long)x) : -1;
}
__global__ __launch_bounds__(kLanes) void k_track_sync_lists(const uint8_t* __restrict__ sb1, int stride, const int* __restrict__ crc_ok,
                                                             const int* __restrict__ frame_type, const int* __restrict__ n_frames,
                                                             const int* __restrict__ chan_first, int frames,
                                                             tetra_lmac_cell_state_t* __restrict__ cell, uint32_t* __restrict__ row_scramb,
                                                             uint32_t* __restrict__ row_time_rx, uint32_t* __restrict__ row_time,
                                                             const uint32_t* __restrict__ frame_bitnum, tetra_lmac_label_t* __restrict__ labels) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int nf = n_frames ? min(n_frames[c], frames) : frames;
    int base = chan_first[c];
    tetra_lmac_cell_state_t st = cell[c];                 // wave-uniform; carried from group to group
    for (int f0 = 0; f0 < frames; f0 += kLanes) {
        const int f = f0 + lane;
        const size_t r = (size_t)c * frames + f;
        const bool is_sync = f < frames && frame_type[r] == TETRA_TRAIN_SYNC;
        const unsigned long long m = __ballot(is_sync);
        const int j = base + __popcll(m & ((1ull << lane) - 1ull));
        base += __popcll(m);
        // a: colour << 2, tn << 8, fn << 11, mn << 16;  b: mcc | mnc << 10  (of a SYNC frame with a good CRC)
        const bool valid = is_sync && f < nf;
        bool good = false;
        uint32_t a = 0, b = 0;
        if (valid && crc_ok[j]) {
            good = true;
            const uint32_t* t2 = reinterpret_cast<const uint32_t*>(sb1 + (size_t)j * stride);
            uint64_t v = 0;                               // type-2 bits 0..55, first bit most significant
#pragma unroll
            for (int k = 0; k < 14; ++k) v |= (uint64_t)pack4(t2[k] & 0x01010101u) << (60 - 4 * k);
            auto field = [&](int first, int len) { return (uint32_t)(v >> (64 - first - len)) & ((1u << len) - 1u); };
            a = (field(4, 6) << 2) | ((field(10, 2) + 1u) << 8) | (field(12, 5) << 11) | (field(17, 6) << 16);
            b = field(31, 10) | (field(41, 14) << 10);
        }
        const unsigned long long mv = __ballot(valid), mg = __ballot(good);
        // tcd as a slot sees it: the fields of good frame h (ah, bh = its words, fetched by every lane: a shuffle reads active lanes
        // only), or what the group started with
        auto tcd_of = [&](int h, uint32_t ah, uint32_t bh, uint32_t& colour, uint32_t& mcc, uint32_t& mnc) {
            Tdma t = { st.tcd_tn, st.tcd_fn, st.tcd_mn };
            colour = st.colour_code; mcc = st.mcc; mnc = st.mnc;
            if (h >= 0) {
                t = Tdma{ (ah >> 8) & 7u, (ah >> 11) & 0x1fu, (ah >> 16) & 0x3fu };
                colour = (ah >> 2) & 0x3fu; mcc = bh & 0x3ffu; mnc = bh >> 10;
            }
            return t;
        };
        const int gp = last_set_upto(mv, lane - 1);                          // the last SYNC frame before this slot
        const int hp = gp >= 0 ? last_set_upto(mg, gp) : -1;                 // ... and the good one tcd held there
        const int hs = last_set_upto(mg, lane);                              // the good one tcd holds after this slot
        const uint32_t ap = (uint32_t)__shfl((int)a, hp < 0 ? 0 : hp), bp = (uint32_t)__shfl((int)b, hp < 0 ? 0 : hp);
        const uint32_t as = (uint32_t)__shfl((int)a, hs < 0 ? 0 : hs), bs = (uint32_t)__shfl((int)b, hs < 0 ? 0 : hs);
        uint32_t cc, mcc, mnc;
        // time on entry: k slots after the last SYNC frame before this one (the clock was set to tcd's time there), or after the group's start
        Tdma from = tcd_of(hp, ap, bp, cc, mcc, mnc);
        if (gp < 0) from = Tdma{ st.phy_tn, st.phy_fn, st.phy_mn };
        const Tdma t_rx = tdma_advance(from, (uint32_t)(lane - gp));
        // after the slot's SB1: tcd as of this slot if it is a SYNC frame, else unchanged
        const Tdma tcd_now = tcd_of(hs, as, bs, cc, mcc, mnc);
        const Tdma t_after = valid ? tcd_now : t_rx;
        const uint32_t scramb = hs >= 0 ? ((((cc & 0x3f) | ((mnc & 0x3fff) << 6) | ((mcc & 0x3ff) << 20)) << 2) | kScrambInitSb1) : st.scramb_init;
        const bool live = f < nf;
        // the group's last live slot is the state the next group (and the next call) starts from; slots past the channel's frame
        // count carry the code in force at its end
        const int last_live = min(nf - f0, kLanes) - 1;                       // < 0: no live slot in this group
        const int src = last_live < 0 ? 0 : last_live;
        const uint32_t code_end = last_live < 0 ? st.scramb_init : (uint32_t)__shfl((int)scramb, src);
        const Tdma end_phy = { (uint32_t)__shfl((int)t_after.tn, src), (uint32_t)__shfl((int)t_after.fn, src), (uint32_t)__shfl((int)t_after.mn, src) };
        const Tdma end_tcd = { (uint32_t)__shfl((int)tcd_now.tn, src), (uint32_t)__shfl((int)tcd_now.fn, src), (uint32_t)__shfl((int)tcd_now.mn, src) };
        const uint32_t end_cc = (uint32_t)__shfl((int)cc, src), end_mcc = (uint32_t)__shfl((int)mcc, src), end_mnc = (uint32_t)__shfl((int)mnc, src);
        if (f < frames) {
            const uint32_t o_rx = live ? tdma_pack(t_rx) : 0u, o_t = live ? tdma_pack(t_after) : 0u;
            row_scramb[r] = live ? scramb : code_end;
            if (row_time_rx) row_time_rx[r] = o_rx;
            if (row_time) row_time[r] = o_t;
            if (labels && valid) {
                tetra_lmac_label_t lb;
                lb.channel = c;
                lb.frame_slot = f;
                lb.bitnum = frame_bitnum[r];
                lb.tdma_time_rx = o_rx;
                lb.tdma_time = o_t;
                lb.crc_ok = good;
                labels[j] = lb;
            }
        }
        if (last_live >= 0) {
            st.phy_tn = end_phy.tn; st.phy_fn = end_phy.fn; st.phy_mn = end_phy.mn;
            st.tcd_tn = end_tcd.tn; st.tcd_fn = end_tcd.fn; st.tcd_mn = end_tcd.mn;
            st.colour_code = end_cc; st.mcc = end_mcc; st.mnc = end_mnc;
            st.scramb_init = code_end;
        }
    }
    if (lane == 0) cell[c] = st;
}

int check_args(int type, const void* in, int n_blocks, int in_stride, const void* init, const void* out, int out_stride,
               const void* ok, bool device_ptrs) {
    if (type < 0 || type > 5 || n_blocks < 0) return TETRA_ERR_ARG;
    if (n_blocks == 0) return TETRA_OK;
    if (!in || !out || !ok) return TETRA_ERR_ARG;
    if (type != TETRA_TPSAP_T_SB1 && !init) return TETRA_ERR_ARG;
    const BlkParam& p = kBlk[type];
    if (in_stride < p.type345 || out_stride < p.type2) return TETRA_ERR_ARG;
    if ((in_stride & 3) || (out_stride & 3)) return TETRA_ERR_ALIGN;
    if (device_ptrs && (((uintptr_t)in & 3) || ((uintptr_t)out & 3))) return TETRA_ERR_ALIGN;
    return TETRA_OK;
}

}  // namespace

extern "C" {

int tetra_lmac_blk_param(int type, tetra_lmac_blk_param_t* out) {
    if (type < 0 || type > 5 || !out) return TETRA_ERR_ARG;
    out->type345_bits = kBlk[type].type345;
    out->type2_bits = kBlk[type].type2;
    out->type1_bits = kBlk[type].type1;
    out->interleave_a = kBlk[type].a;
    out->have_crc16 = kBlk[type].crc;
    return TETRA_OK;
}

uint32_t tetra_lmac_scramb_init(uint16_t mcc, uint16_t mnc, uint8_t colour) {
    // tetra_scramb.c:87-99: colour (6 bits) | MNC (14) << 6 | MCC (10) << 20, then two 1 bits shifted in below
    const uint32_t v = (uint32_t)(colour & 0x3f) | ((uint32_t)(mnc & 0x3fff) << 6) | ((uint32_t)(mcc & 0x3ff) << 20);
    return (v << 2) | kScrambInitSb1;
}

int tetra_lmac_decode_batch_device(int type, const uint8_t* d_type5, int n_blocks, int in_stride, const uint32_t* d_scramb_init,
                                   uint8_t* d_type2, int out_stride, int32_t* d_crc_ok, void* hip_stream) {
    return tetra_lmac_decode_counted_device(type, d_type5, n_blocks, nullptr, in_stride, d_scramb_init, nullptr, d_type2, out_stride,
                                            d_crc_ok, hip_stream);
}

int tetra_lmac_decode_counted_device(int type, const uint8_t* d_type5, int n_blocks, const int32_t* d_n_blocks, int in_stride,
                                     const uint32_t* d_scramb_init, const int32_t* d_init_index, uint8_t* d_type2, int out_stride,
                                     int32_t* d_crc_ok, void* hip_stream) {
    const int rc = check_args(type, d_type5, n_blocks, in_stride, d_scramb_init, d_type2, out_stride, d_crc_ok, true);
    if (rc != TETRA_OK || n_blocks == 0) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const BlkParam& p = kBlk[type];
    if (type == TETRA_TPSAP_T_BBK) {
        hipLaunchKernelGGL(k_lmac_bbk, dim3((n_blocks + 255) / 256), dim3(256), 0, s, d_type5, n_blocks, in_stride, d_scramb_init,
                           p.type345, d_type2, out_stride, d_crc_ok, d_n_blocks, d_init_index);
    } else {
        // decision scratch: (type2 + 4) steps x 64 lanes x u16 per workgroup, from the library's keeping pool (scratch_pool():
        // after the first call a free-list hit), released in stream order right behind the kernel
        const int groups = (n_blocks + kLanes - 1) / kLanes;
        const int dec_pairs = (p.type2 + kFlush) / 2;
        const size_t bytes = (size_t)groups * dec_pairs * kLanes * sizeof(uint32_t);
        const uint32_t* seq = seq_table();
        if (!seq) return TETRA_ERR_NOMEM;
        uint32_t* scratch = nullptr;
        hipMemPool_t pool = scratch_pool();
        const hipError_t got = pool ? hipMallocFromPoolAsync(reinterpret_cast<void**>(&scratch), bytes, pool, s)
                                    : hipMallocAsync(reinterpret_cast<void**>(&scratch), bytes, s);
        if (got != hipSuccess) { (void)hipGetLastError(); return TETRA_ERR_NOMEM; }
        hipLaunchKernelGGL(k_lmac_decode, dim3(groups), dim3(kLanes), 0, s, d_type5, n_blocks, in_stride, d_scramb_init,
                           type == TETRA_TPSAP_T_SB1 ? 1 : 0, p.type345, p.type2, p.type1, p.a, d_type2, out_stride, d_crc_ok,
                           scratch, dec_pairs, d_n_blocks, d_init_index, g_force_byte_route ? nullptr : seq);
        const hipError_t launch = hipGetLastError();
        if (hipFreeAsync(scratch, s) != hipSuccess || launch != hipSuccess) return TETRA_ERR_HIP;
        return TETRA_OK;
    }
    return hipGetLastError() == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
}

int tetra_lmac_decode_frames_device(const tetra_lmac_frames_t* src, const tetra_lmac_job_t* jobs, int n_jobs, void* hip_stream) {
    if (!src || !jobs || n_jobs < 0 || n_jobs > TETRA_LMAC_MAX_JOBS) return TETRA_ERR_ARG;
    if (n_jobs == 0) return TETRA_OK;
    if (!src->d_frames || !src->d_frame_type || src->n_frames < 0) return TETRA_ERR_ARG;
    if ((uintptr_t)src->d_frames & 15) return TETRA_ERR_ALIGN;
    JobTable tab = {};
    if (src->n_frames == 0) return TETRA_OK;              // no frames: no row can exist
    tab.src = DevFrames{ src->d_frames, src->d_frame_type, src->d_frame_bitnum, src->d_time_rx, src->d_time, src->frames_per_channel, src->n_frames };
    long long groups_total = 0, scratch_words = 0;
    int n = 0, max_pairs = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const tetra_lmac_job_t& j = jobs[i];
        if (j.type < 0 || j.type > 5 || j.max_rows < 0) return TETRA_ERR_ARG;
        if (j.max_rows == 0) continue;
        if (!j.d_row_frame || !j.d_type2 || !j.d_crc_ok) return TETRA_ERR_ARG;
        if (j.type != TETRA_TPSAP_T_SB1 && !j.d_frame_scramb) return TETRA_ERR_ARG;
        if (j.d_labels && (!src->d_frame_bitnum || !src->d_time_rx || !src->d_time || src->frames_per_channel < 1)) return TETRA_ERR_ARG;
        int layout = kLayoutNone;
        switch (j.type) {
            case TETRA_TPSAP_T_SB1: layout = j.blk_num == 1 ? kLayoutSb1 : kLayoutNone; break;
            case TETRA_TPSAP_T_SB2: layout = j.blk_num == 2 ? kLayoutSb2 : kLayoutNone; break;
            case TETRA_TPSAP_T_NDB: layout = j.blk_num == 1 ? kLayoutNdb1 : j.blk_num == 2 ? kLayoutNdb2 : kLayoutNone; break;
            case TETRA_TPSAP_T_BBK: layout = kLayoutBbk; break;
            case TETRA_TPSAP_T_SCH_F: layout = kLayoutSchF; break;
            default: break;                                    // SCH/HU: an uplink block, no downlink burst carries it
        }
        if (layout == kLayoutNone) return TETRA_ERR_ARG;       // no burst type carries this (kind, block number)
        const BlkParam& p = kBlk[j.type];
        if (j.out_stride < (layout == kLayoutBbk ? 32 : p.type2)) return TETRA_ERR_SIZE;
        if ((j.out_stride & 7) || ((uintptr_t)j.d_type2 & 7)) return TETRA_ERR_ALIGN;
        DevJob& d = tab.job[n++];
        d.row_frame = j.d_row_frame;
        d.n_rows_dev = j.d_n_rows;
        d.frame_scramb = j.type == TETRA_TPSAP_T_SB1 ? nullptr : j.d_frame_scramb;
        d.out = j.d_type2;
        d.crc_ok = j.d_crc_ok;
        d.labels = j.d_labels;
        d.n_rows = j.max_rows;
        d.out_stride = j.out_stride;
        d.layout = layout;
        d.type345 = p.type345; d.type2 = p.type2; d.a = p.a;
        d.dec_pairs = layout == kLayoutBbk ? 0 : (p.type2 + kFlush) / 2;
        max_pairs = d.dec_pairs > max_pairs ? d.dec_pairs : max_pairs;
        const long long groups = ((long long)j.max_rows + kLanes - 1) / kLanes;
        d.first_group = (int)groups_total;
        d.scratch_base = scratch_words;
        groups_total += groups;
        scratch_words += groups * d.dec_pairs * kLanes;
        if (groups_total > 0x7fffffffLL) return TETRA_ERR_SIZE;
    }
    tab.n = n;
    if (n == 0) return TETRA_OK;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const uint32_t* seq = seq_table();
    if (!seq) return TETRA_ERR_NOMEM;
    const size_t bytes = (size_t)scratch_words * sizeof(uint32_t);
    uint32_t* scratch = nullptr;
    bool pooled = false;
    if (scratch_words) {
        if (src->d_workspace) {
            if (src->workspace_bytes < bytes) return TETRA_ERR_SIZE;
            if ((uintptr_t)src->d_workspace & 3) return TETRA_ERR_ALIGN;
            scratch = static_cast<uint32_t*>(src->d_workspace);
        } else {
            hipMemPool_t pool = scratch_pool();
            const hipError_t got = pool ? hipMallocFromPoolAsync(reinterpret_cast<void**>(&scratch), bytes, pool, s)
                                        : hipMallocAsync(reinterpret_cast<void**>(&scratch), bytes, s);
            if (got != hipSuccess) { (void)hipGetLastError(); return TETRA_ERR_NOMEM; }
            pooled = true;
        }
    }
    hipLaunchKernelGGL(k_lmac_frames, dim3((unsigned)groups_total), dim3(kLanes), 0, s, tab, scratch, seq);
    const hipError_t launch = hipGetLastError();
    if ((pooled && hipFreeAsync(scratch, s) != hipSuccess) || launch != hipSuccess) return TETRA_ERR_HIP;
    return TETRA_OK;
}

size_t tetra_lmac_decode_frames_workspace_bytes(const tetra_lmac_job_t* jobs, int n_jobs) {
    if (!jobs || n_jobs < 0) return 0;
    size_t words = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const tetra_lmac_job_t& j = jobs[i];
        if (j.type < 0 || j.type > 5 || j.type == TETRA_TPSAP_T_BBK || j.max_rows <= 0) continue;
        words += (size_t)(((long long)j.max_rows + kLanes - 1) / kLanes) * ((kBlk[j.type].type2 + kFlush) / 2) * kLanes;
    }
    return words * sizeof(uint32_t);
}

int tetra_lmac_debug_force_byte_route(int on) {
    const int was = g_force_byte_route ? 1 : 0;
    g_force_byte_route = on != 0;
    return was;
}

int tetra_lmac_track_scramb_device(const uint8_t* d_sb1_type2, int type2_stride, const int32_t* d_crc_ok, const int32_t* d_valid,
                                   int n_channels, int frames_per_channel, uint32_t* d_chan_scramb, uint32_t* d_row_scramb,
                                   void* hip_stream) {
    if (!d_sb1_type2 || !d_crc_ok || !d_valid || !d_chan_scramb || !d_row_scramb) return TETRA_ERR_ARG;
    if (n_channels < 1 || frames_per_channel < 0 || type2_stride < 60) return TETRA_ERR_ARG;
    hipLaunchKernelGGL(k_track_scramb, dim3((n_channels + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_sb1_type2,
                       type2_stride, d_crc_ok, d_valid, n_channels, frames_per_channel, d_chan_scramb, d_row_scramb);
    return hipGetLastError() == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
}

int tetra_lmac_track_sync_device(const uint8_t* d_sb1_type2, int type2_stride, const int32_t* d_crc_ok, const int32_t* d_valid,
                                 const int32_t* d_n_frames, int n_channels, int frames_per_channel, tetra_lmac_cell_state_t* d_cell,
                                 uint32_t* d_row_scramb, uint32_t* d_row_time_rx, uint32_t* d_row_time, void* hip_stream) {
    if (!d_sb1_type2 || !d_crc_ok || !d_valid || !d_cell || !d_row_scramb) return TETRA_ERR_ARG;
    if (n_channels < 1 || frames_per_channel < 0 || type2_stride < 60) return TETRA_ERR_ARG;
    hipLaunchKernelGGL(k_track_sync, dim3((n_channels + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_sb1_type2,
                       type2_stride, d_crc_ok, d_valid, d_n_frames, n_channels, frames_per_channel, d_cell, d_row_scramb, d_row_time_rx,
                       d_row_time);
    return hipGetLastError() == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
}

int tetra_lmac_track_sync_lists_device(const uint8_t* d_sb1_type2, int type2_stride, const int32_t* d_crc_ok, const int32_t* d_frame_type,
                                       const int32_t* d_n_frames, const int32_t* d_chan_first_sync, int n_channels, int frames_per_channel,
                                       tetra_lmac_cell_state_t* d_cell, uint32_t* d_row_scramb, uint32_t* d_row_time_rx, uint32_t* d_row_time,
                                       const uint32_t* d_frame_bitnum, tetra_lmac_label_t* d_sb1_labels, void* hip_stream) {
    if (!d_sb1_type2 || !d_crc_ok || !d_frame_type || !d_chan_first_sync || !d_cell || !d_row_scramb) return TETRA_ERR_ARG;
    if (n_channels < 1 || frames_per_channel < 0 || type2_stride < 60 || (d_sb1_labels && !d_frame_bitnum)) return TETRA_ERR_ARG;
    if ((type2_stride & 3) || ((uintptr_t)d_sb1_type2 & 3)) return TETRA_ERR_ALIGN;
    if (frames_per_channel == 0) return TETRA_OK;
    hipLaunchKernelGGL(k_track_sync_lists, dim3(n_channels), dim3(kLanes), 0, static_cast<hipStream_t>(hip_stream), d_sb1_type2, type2_stride, d_crc_ok, d_frame_type, d_n_frames, d_chan_first_sync,
                       frames_per_channel, d_cell, d_row_scramb, d_row_time_rx, d_row_time, d_frame_bitnum, d_sb1_labels);
    return hipGetLastError() == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
}

int tetra_lmac_decode_batch(int type, const uint8_t* type5, int n_blocks, int in_stride, const uint32_t* scramb_init,
                            uint8_t* type2, int out_stride, int32_t* crc_ok, int device) {
    int rc = check_args(type, type5, n_blocks, in_stride, scramb_init, type2, out_stride, crc_ok, false);
    if (rc != TETRA_OK || n_blocks == 0) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TETRA_ERR_NO_DEVICE;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return TETRA_ERR_HIP;
    uint8_t *d_in = nullptr, *d_out = nullptr;
    uint32_t* d_init = nullptr;
    int32_t* d_ok = nullptr;
    const size_t in_bytes = (size_t)n_blocks * in_stride, out_bytes = (size_t)n_blocks * out_stride;
    rc = TETRA_ERR_HIP;
    do {
        if (hipMalloc(&d_in, in_bytes) != hipSuccess || hipMalloc(&d_out, out_bytes) != hipSuccess ||
            hipMalloc(&d_ok, sizeof(int32_t) * n_blocks) != hipSuccess) { rc = TETRA_ERR_NOMEM; break; }
        if (scramb_init) {
            if (hipMalloc(&d_init, sizeof(uint32_t) * n_blocks) != hipSuccess) { rc = TETRA_ERR_NOMEM; break; }
            if (hipMemcpy(d_init, scramb_init, sizeof(uint32_t) * n_blocks, hipMemcpyHostToDevice) != hipSuccess) break;
        }
        if (hipMemcpy(d_in, type5, in_bytes, hipMemcpyHostToDevice) != hipSuccess) break;
        const int krc = tetra_lmac_decode_batch_device(type, d_in, n_blocks, in_stride, d_init, d_out, out_stride, d_ok, nullptr);
        if (krc != TETRA_OK) { rc = krc; break; }
        if (hipDeviceSynchronize() != hipSuccess) break;
        // only the type2_bits columns: the caller's row padding is left alone
        // (rows without padding: one contiguous copy -- a strided device-to-host copy of many narrow rows crawls)
        if (out_stride == kBlk[type].type2 ? hipMemcpy(type2, d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess
                                           : hipMemcpy2D(type2, out_stride, d_out, out_stride, kBlk[type].type2, n_blocks, hipMemcpyDeviceToHost) != hipSuccess) break;
        if (hipMemcpy(crc_ok, d_ok, sizeof(int32_t) * n_blocks, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = TETRA_OK;
    } while (false);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    (void)hipFree(d_init);
    (void)hipFree(d_ok);
    return rc;
}

}  // extern "C"
