// tetra_burst_scan.hip -- batched training-sequence search (include/tetra_burst_scan.h), bit-exact with the reference's
// tetra_find_train_seq() (src/decoder/src/phy/tetra_burst.c:271-341).
//
// One 256-thread workgroup per channel.  The row is walked in tiles of 8192 positions: the tile's bytes (one bit each)
// are read once with coalesced dword loads and packed MSB-first into 32-bit words in LDS; every position then pulls its
// 22-bit look-ahead window out of two adjacent words with a funnel shift and compares it with the five sequence heads.
// Candidates are verified against the full sequence (bytes, rare) and reduced with an LDS atomicMin on
// (position << 3 | check order), which is exactly "first position, then the reference's if-chain order".  The first 21
// positions reproduce the reference's misaligned pre-filter (see the header).  Integer/byte work: HBM-bound, every input
// byte is read from HBM once.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <vector>

#include "../../include/tetra_burst_scan.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 32768;                 // positions per tile
constexpr int kTileWords = kTile / 32 + 2;   // + look-ahead

// ETSI EN 300 392-2 9.4.4.3.2-4 (the reference holds the same bits at tetra_burst.c:61-72)
__constant__ uint8_t c_seq[5][38] = {
    /* check order of the reference's if-chain: y (sync), n, p, q (normal 1-3), x (extended) */
    { 1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1 },
    { 1,1, 0,1, 0,0, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,1, 0,0 },
    { 0,1, 1,1, 1,0, 1,0, 0,1, 0,0, 0,0, 1,1, 0,1, 1,1, 1,0 },
    { 1,0, 1,1, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 1,0, 1,1, 0,1 },
    { 1,0, 0,1, 1,1, 0,1, 0,0, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,1, 0,0, 0,0, 1,1 },
};
__constant__ int c_len[5] = { 38, 22, 22, 22, 30 };
__constant__ int c_type[5] = { TETRA_TRAIN_SYNC, TETRA_TRAIN_NORM_1, TETRA_TRAIN_NORM_2, TETRA_TRAIN_NORM_3, TETRA_TRAIN_EXT };

__device__ __forceinline__ unsigned head22(int s) {
    unsigned v = 0;
    for (int i = 0; i < 22; i++) v = (v << 1) | c_seq[s][i];
    return v;
}

// full check of the reference's if-chain at position cur; returns the check-order index 0..4 or 5 for none
__device__ int verify(const uint8_t* in, int cur, int end_of_in, unsigned mask) {
    const int remain = end_of_in - cur;
    for (int s = 0; s < 5; s++) {
        if (!(mask & (1u << c_type[s])) || remain < c_len[s]) continue;
        bool eq = true;
        for (int i = 0; i < c_len[s]; i++)
            if (in[cur + i] != c_seq[s][i]) { eq = false; break; }
        if (eq) return s;
    }
    return 5;
}

__global__ __launch_bounds__(kThreads) void k_find_train_seq(const uint8_t* bits, int bits_stride, const int* end_of_in,
                                                             unsigned mask, int* type_out, int* off_out) {
    __shared__ unsigned packed[kTileWords];
    __shared__ unsigned best;              // (cur << 3) | check-order index
    __shared__ unsigned heads[5];
    const int ch = blockIdx.x;
    const uint8_t* in = bits + (long long)ch * bits_stride;
    // the search reads in[cur + 21] for every cur < end (tetra_burst.c:296): a count that would take that look-ahead out of
    // the row is cut back to what the row holds (documented in tetra_burst_scan.h: rows extend 21 bytes past end_of_in)
    int end = end_of_in[ch];
    end = end > bits_stride - 21 ? bits_stride - 21 : end;
    if (threadIdx.x == 0) best = 0xffffffffu;
    if (threadIdx.x < 5) heads[threadIdx.x] = head22(threadIdx.x);
    __syncthreads();

    // positions 0..20: the reference's pre-filter is seeded with in[0..19] and then receives in[cur+21] (in[20] is skipped)
    if (threadIdx.x == 0 && end > 0) {
        unsigned filter = 0;
        for (int i = 0; i < 20; i++) filter = (filter << 1) | in[i];
        const int lim = end < 21 ? end : 21;
        for (int cur = 0; cur < lim; cur++) {
            filter = ((filter << 1) | in[cur + 21]) & 0x3fffffu;
            bool m = false;
            for (int s = 0; s < 5; s++) m |= (filter == heads[s]);
            if (m) {
                const int s = verify(in, cur, end, mask);
                if (s < 5) { atomicMin(&best, ((unsigned)cur << 3) | (unsigned)s); break; }
            }
        }
    }

    for (int base = 0; base < end; base += kTile) {
        __syncthreads();
        if (best != 0xffffffffu && (int)(best >> 3) < base) break;      // an earlier match ends the scan (uniform)
        // pack bytes [base, base + kTile + 64) to bits, MSB first; bytes past the row are taken as 0 (never reached by
        // a position < end whose 22-bit window lies inside end + 21 <= bits_stride)
        for (int w = threadIdx.x; w < kTileWords; w += kThreads) {
            const int b0 = base + 32 * w;
            unsigned v = 0;
            if (b0 + 32 <= bits_stride) {
                // 32 bytes as two 16-byte loads (rows and tiles are 16-byte aligned when bits_stride % 16 == 0; otherwise the
                // dword path below is used); 8 bytes -> 8 bits, first byte = MSB, with one multiply: the partial products
                // of x * 0x8040201008040201 land on distinct bit positions, byte i reaching bit 63 - i.
                unsigned long long q[4];
                if (((bits_stride | b0) & 15) == 0) {
                    const uint4 lo = *reinterpret_cast<const uint4*>(in + b0);
                    const uint4 hi = *reinterpret_cast<const uint4*>(in + b0 + 16);
                    q[0] = ((unsigned long long)lo.y << 32) | lo.x; q[1] = ((unsigned long long)lo.w << 32) | lo.z;
                    q[2] = ((unsigned long long)hi.y << 32) | hi.x; q[3] = ((unsigned long long)hi.w << 32) | hi.z;
                } else {
                    const unsigned* pd = reinterpret_cast<const unsigned*>(in + b0);     // rows are 4-byte aligned
                    for (int z = 0; z < 4; z++) q[z] = ((unsigned long long)pd[2 * z + 1] << 32) | pd[2 * z];
                }
                for (int z = 0; z < 4; z++)
                    v |= (unsigned)(((q[z] & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56) << (24 - 8 * z);
            } else {
                for (int z = 0; z < 32; z++)
                    if (b0 + z < bits_stride) v |= (unsigned)(in[b0 + z] & 1u) << (31 - z);
            }
            packed[w] = v;
        }
        __syncthreads();
        const int lim = (end - base < kTile) ? (end - base) : kTile;
        for (int r = threadIdx.x; r < lim; r += kThreads) {
            const int cur = base + r;
            if (cur < 21) continue;                                      // handled above
            const unsigned long long two = ((unsigned long long)packed[r >> 5] << 32) | packed[(r >> 5) + 1];
            const unsigned f = (unsigned)(two >> (64 - 22 - (r & 31))) & 0x3fffffu;
            if (f == heads[0] || f == heads[1] || f == heads[2] || f == heads[3] || f == heads[4]) {
                const int s = verify(in, cur, end, mask);
                if (s < 5) atomicMin(&best, ((unsigned)cur << 3) | (unsigned)s);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (best == 0xffffffffu) { type_out[ch] = -1; off_out[ch] = -1; }
        else { type_out[ch] = c_type[best & 7u]; off_out[ch] = (int)(best >> 3); }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The plugin's indicator (src/main.cpp:385-414): see tetra_burst_scan.h.  One 256-thread workgroup per channel; the stream
// v = [carried 44 bits | this call's bits] is packed MSB-first into LDS tile by tile like above, every position pulls its
// 45-bit window out of three adjacent words and compares its head with the eight sequences; the last hit is an LDS atomicMax.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kIndWin = 45, kIndTail = kIndWin - 1, kIndArm = 2048;
constexpr int kIndTile = 8192, kIndTileWords = kIndTile / 32 + 3;

// main.cpp:457-468, in the order of the if-chain at :395-402 (the order does not matter: any hit arms the counter)
__constant__ uint8_t c_ind_seq[8][45] = {
    { 1,1, 0,1, 0,0, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,1, 0,0 },
    { 0,1, 1,1, 1,0, 1,0, 0,1, 0,0, 0,0, 1,1, 0,1, 1,1, 1,0 },
    { 1,0, 1,1, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 1,0, 1,1, 0,1 },
    { 1,1,1, 0,0,1, 1,0,1, 1,1,1, 0,0,0, 1,1,1, 1,0,0, 0,1,1, 1,1,0, 0,0,0, 0,0,0 },
    { 1,0,1, 0,1,1, 1,1,1, 1,0,1, 0,1,0, 1,0,1, 1,1,0, 0,0,1, 1,0,0, 0,1,0, 0,1,0 },
    { 1,0, 0,1, 1,1, 0,1, 0,0, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,1, 0,0, 0,0, 1,1 },
    { 0,1,1,1,0,0,1,1,0,1,0,0,0,0,1,0,0,0,1,1,1,0,1,1,0,1,0,1,0,1,1,1,1,1,0,1,0,0,0,0,0,1,1,1,0 },
    { 1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1 },
};
__constant__ int c_ind_len[8] = { 22, 22, 22, 33, 33, 30, 45, 38 };

__global__ __launch_bounds__(kThreads) void k_ts_indicator(const uint8_t* bits, int bits_stride, const int* n_bits, uint8_t* tail,
                                                           int* expire, uint8_t* found_out, int* expire_out) {
    __shared__ unsigned packed[kIndTileWords];
    __shared__ unsigned long long heads[8];       // sequence s as the top c_ind_len[s] bits of a 45-bit window
    __shared__ uint8_t new_tail[kIndTail];
    __shared__ int last_hit;
    const int ch = blockIdx.x;
    const uint8_t* in = bits + (long long)ch * bits_stride;
    uint8_t* tl = tail + (long long)ch * kIndTail;
    int n = n_bits[ch];
    n = n < 0 ? 0 : (n > bits_stride ? bits_stride : n);
    if (threadIdx.x == 0) last_hit = -1;
    if (threadIdx.x < 8) {
        unsigned long long v = 0;
        for (int i = 0; i < c_ind_len[threadIdx.x]; i++) v = (v << 1) | c_ind_seq[threadIdx.x][i];
        heads[threadIdx.x] = v;
    }
    // v[i] = i < 44 ? carried bit i : bits[i - 44]; position q (the window after bit q of the call) covers v[q .. q + 44]
    auto vbit = [&](int i) -> unsigned { return i < kIndTail ? (tl[i] & 1u) : (i - kIndTail < n ? (in[i - kIndTail] & 1u) : 0u); };
    for (int base = 0; base < n; base += kIndTile) {
        __syncthreads();
        for (int w = threadIdx.x; w < kIndTileWords; w += kThreads) {
            const int b0 = base + 32 * w;
            unsigned v = 0;
            if (b0 >= kIndTail + 4 && b0 - kIndTail + 32 <= n) {
                const unsigned* pd = reinterpret_cast<const unsigned*>(in + (b0 - kIndTail));      // 44 % 4 == 0: dword aligned
                for (int z = 0; z < 4; z++) {
                    const unsigned long long q = ((unsigned long long)pd[2 * z + 1] << 32) | pd[2 * z];
                    v |= (unsigned)(((q & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56) << (24 - 8 * z);
                }
            } else {
                for (int z = 0; z < 32; z++) v |= vbit(b0 + z) << (31 - z);
            }
            packed[w] = v;
        }
        __syncthreads();
        const int lim = (n - base < kIndTile) ? (n - base) : kIndTile;
        int mine = -1;
        for (int r = threadIdx.x; r < lim; r += kThreads) {
            const int sh = r & 31;
            const unsigned long long hi = ((unsigned long long)packed[r >> 5] << 32) | packed[(r >> 5) + 1];
            const unsigned long long top = sh ? ((hi << sh) | ((unsigned long long)packed[(r >> 5) + 2] >> (32 - sh))) : hi;
            const unsigned long long win = top >> (64 - kIndWin);
            bool hit = false;
            for (int s = 0; s < 8; s++) hit |= (win >> (kIndWin - c_ind_len[s])) == heads[s];
            if (hit) mine = base + r;       // r ascends: the thread's last hit
        }
        if (mine >= 0) atomicMax(&last_hit, mine);
    }
    __syncthreads();
    if (n > 0) {
        if (threadIdx.x < kIndTail) new_tail[threadIdx.x] = (uint8_t)vbit(n + threadIdx.x);
        __syncthreads();
        if (threadIdx.x < kIndTail) tl[threadIdx.x] = new_tail[threadIdx.x];
    }
    if (threadIdx.x == 0) {
        int e = expire[ch];
        if (n > 0) {
            // a hit at bit p arms 2048 and the same bit counts it down to 2047; every later bit takes one more
            if (last_hit >= 0) e = kIndArm - 1 - (n - 1 - last_hit);
            else e = e - n;
            e = e < 0 ? 0 : e;
            expire[ch] = e;
        }
        found_out[ch] = e > 0 ? 1 : 0;
        if (expire_out) expire_out[ch] = e;
    }
}

}  // namespace

extern "C" {

int tetra_find_train_seq_batch_device(const uint8_t* d_bits, int n_channels, int bits_stride, const int32_t* d_end_of_in,
                                      uint32_t mask, int32_t* d_type, int32_t* d_offset, void* hip_stream) {
    if (!d_bits || !d_end_of_in || !d_type || !d_offset || n_channels < 1 || bits_stride < 4) return TETRA_ERR_ARG;
    if ((bits_stride & 3) || (reinterpret_cast<uintptr_t>(d_bits) & 3)) return TETRA_ERR_ALIGN;
    hipLaunchKernelGGL(k_find_train_seq, dim3(n_channels), dim3(kThreads), 0, (hipStream_t)hip_stream, d_bits, bits_stride,
                       d_end_of_in, mask, d_type, d_offset);
    return hipGetLastError() == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
}

int tetra_find_train_seq_batch(const uint8_t* bits, int n_channels, int bits_stride, const int32_t* end_of_in, uint32_t mask,
                               int32_t* type, int32_t* offset, int device) {
    if (!bits || !end_of_in || !type || !offset || n_channels < 1 || bits_stride < 4) return TETRA_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TETRA_ERR_NO_DEVICE;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (device >= 0) {
        if (device >= ndev || hipSetDevice(device) != hipSuccess) return TETRA_ERR_NO_DEVICE;
    }
    uint8_t* d_bits = nullptr;
    int *d_end = nullptr, *d_t = nullptr, *d_o = nullptr;
    const size_t nb = (size_t)n_channels * (size_t)bits_stride;
    int rc = TETRA_OK;
    if (hipMalloc((void**)&d_bits, nb) != hipSuccess || hipMalloc((void**)&d_end, sizeof(int) * n_channels) != hipSuccess ||
        hipMalloc((void**)&d_t, sizeof(int) * n_channels) != hipSuccess || hipMalloc((void**)&d_o, sizeof(int) * n_channels) != hipSuccess)
        rc = TETRA_ERR_NOMEM;
    if (rc == TETRA_OK && (hipMemcpy(d_bits, bits, nb, hipMemcpyHostToDevice) != hipSuccess ||
                           hipMemcpy(d_end, end_of_in, sizeof(int) * n_channels, hipMemcpyHostToDevice) != hipSuccess))
        rc = TETRA_ERR_HIP;
    if (rc == TETRA_OK) rc = tetra_find_train_seq_batch_device(d_bits, n_channels, bits_stride, d_end, mask, d_t, d_o, nullptr);
    if (rc == TETRA_OK && (hipStreamSynchronize(0) != hipSuccess ||
                           hipMemcpy(type, d_t, sizeof(int) * n_channels, hipMemcpyDeviceToHost) != hipSuccess ||
                           hipMemcpy(offset, d_o, sizeof(int) * n_channels, hipMemcpyDeviceToHost) != hipSuccess))
        rc = TETRA_ERR_HIP;
    if (d_bits) (void)hipFree(d_bits);
    if (d_end) (void)hipFree(d_end);
    if (d_t) (void)hipFree(d_t);
    if (d_o) (void)hipFree(d_o);
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

struct tetra_ts_indicator {
    int C = 0, device = 0;
    uint8_t* tail = nullptr;      // [C][44]
    int* expire = nullptr;        // [C]
};

namespace {
struct IndDeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit IndDeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (dev != prev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~IndDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

int tetra_ts_indicator_create(int n_channels, int device, tetra_ts_indicator_t** out) {
    if (!out) return TETRA_ERR_ARG;
    *out = nullptr;
    if (n_channels < 1) return TETRA_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TETRA_ERR_NO_DEVICE;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return TETRA_ERR_NO_DEVICE;
    if (device >= ndev) return TETRA_ERR_NO_DEVICE;
    IndDeviceGuard g(device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    tetra_ts_indicator* h = new (std::nothrow) tetra_ts_indicator;
    if (!h) return TETRA_ERR_NOMEM;
    h->C = n_channels;
    h->device = device;
    if (hipMalloc((void**)&h->tail, (size_t)n_channels * kIndTail) != hipSuccess ||
        hipMalloc((void**)&h->expire, sizeof(int) * (size_t)n_channels) != hipSuccess) {
        tetra_ts_indicator_destroy(h);
        return TETRA_ERR_NOMEM;
    }
    const int rc = tetra_ts_indicator_reset(h, -1);
    if (rc != TETRA_OK) { tetra_ts_indicator_destroy(h); return rc; }
    *out = h;
    return TETRA_OK;
}

void tetra_ts_indicator_destroy(tetra_ts_indicator_t* h) {
    if (!h) return;
    IndDeviceGuard g(h->device);
    if (h->tail) (void)hipFree(h->tail);
    if (h->expire) (void)hipFree(h->expire);
    delete h;
}

int tetra_ts_indicator_reset(tetra_ts_indicator_t* h, int channel) {
    if (!h || channel < -1 || channel >= h->C) return TETRA_ERR_ARG;
    IndDeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    const size_t first = channel < 0 ? 0 : (size_t)channel, count = channel < 0 ? (size_t)h->C : 1;
    if (hipMemset(h->tail + first * kIndTail, 0, count * kIndTail) != hipSuccess ||
        hipMemset(h->expire + first, 0, sizeof(int) * count) != hipSuccess)
        return TETRA_ERR_HIP;
    return TETRA_OK;
}

int tetra_ts_indicator_process_device(tetra_ts_indicator_t* h, const uint8_t* d_bits, int bits_stride, const int32_t* d_n_bits,
                                      uint8_t* d_found, int32_t* d_expire, void* hip_stream) {
    if (!h || !d_bits || !d_n_bits || !d_found || bits_stride < 4) return TETRA_ERR_ARG;
    if ((bits_stride & 3) || (reinterpret_cast<uintptr_t>(d_bits) & 3)) return TETRA_ERR_ALIGN;
    IndDeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    hipLaunchKernelGGL(k_ts_indicator, dim3(h->C), dim3(kThreads), 0, (hipStream_t)hip_stream, d_bits, bits_stride, d_n_bits,
                       h->tail, h->expire, d_found, d_expire);
    return hipGetLastError() == hipSuccess ? TETRA_OK : TETRA_ERR_HIP;
}

int tetra_ts_indicator_process(tetra_ts_indicator_t* h, const uint8_t* bits, int bits_stride, const int32_t* n_bits,
                               uint8_t* found, int32_t* expire) {
    if (!h || !bits || !n_bits || !found || bits_stride < 4) return TETRA_ERR_ARG;
    if (bits_stride & 3) return TETRA_ERR_ALIGN;
    IndDeviceGuard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    uint8_t *d_bits = nullptr, *d_found = nullptr;
    int *d_n = nullptr, *d_e = nullptr;
    const size_t nb = (size_t)h->C * (size_t)bits_stride;
    int rc = TETRA_OK;
    if (hipMalloc((void**)&d_bits, nb) != hipSuccess || hipMalloc((void**)&d_n, sizeof(int) * h->C) != hipSuccess ||
        hipMalloc((void**)&d_found, (size_t)h->C) != hipSuccess || hipMalloc((void**)&d_e, sizeof(int) * h->C) != hipSuccess)
        rc = TETRA_ERR_NOMEM;
    if (rc == TETRA_OK && (hipMemcpy(d_bits, bits, nb, hipMemcpyHostToDevice) != hipSuccess ||
                           hipMemcpy(d_n, n_bits, sizeof(int) * h->C, hipMemcpyHostToDevice) != hipSuccess))
        rc = TETRA_ERR_HIP;
    if (rc == TETRA_OK) rc = tetra_ts_indicator_process_device(h, d_bits, bits_stride, d_n, d_found, d_e, nullptr);
    if (rc == TETRA_OK && (hipStreamSynchronize(0) != hipSuccess ||
                           hipMemcpy(found, d_found, (size_t)h->C, hipMemcpyDeviceToHost) != hipSuccess ||
                           (expire && hipMemcpy(expire, d_e, sizeof(int) * h->C, hipMemcpyDeviceToHost) != hipSuccess)))
        rc = TETRA_ERR_HIP;
    if (d_bits) (void)hipFree(d_bits);
    if (d_n) (void)hipFree(d_n);
    if (d_found) (void)hipFree(d_found);
    if (d_e) (void)hipFree(d_e);
    return rc;
}

}  // extern "C"
