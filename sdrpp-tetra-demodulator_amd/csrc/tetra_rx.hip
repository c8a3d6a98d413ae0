// tetra_rx.hip -- the device-resident receive chain behind one handle (include/tetra_rx.h): the ordering, buffers, streams and
// events around this library's own stage entry points.
//
// Reference chain per receiver: tetra_burst_sync_in (phy/tetra_burst_sync.c:54-155) -> tetra_burst_rx_cb (phy/tetra_burst.c:343-393)
// -> tp_sap_udata_ind (lower_mac/tetra_lower_mac.c:148-237) with the cell state fed back at :246-275.  Here, per process call k:
//
//   caller's stream  [wait: tail k-2 has read bit rows k & 1]  demodulator -> bits[k & 1]                          (event D_k)
//   tail stream      [wait D_k]  synchroniser (packed frames, types, bit numbers, counts)
//                    frame lists (SYNC / NORM_1 / NORM_2 / any) in one pass
//                    SB1:   decode straight from the SYNC frames -> tracker: cell state, per-slot scrambling code, TDMA time
//                           before / after the slot's SB1, the SB1 rows' labels
//                    every other configured kind: ONE launch that decodes them all straight from the frames with the TRACKER's
//                           per-slot codes and labels every row (channel, slot, bit number, times, crc)            (event T_k)
//
// so the demodulator of call k+1 runs beside the tail of call k; results and bit rows are double buffered by call parity.
// (Until round 6: per kind a compacting demultiplexer into byte rows (4 launches), a counted decode and a label kernel -- 40 launches
// and 0.36 GB of byte rows per second of 4096 channels; now 7 launches and no byte rows.)
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "../../include/tetra_rx.h"

namespace {

struct KindInfo {
    int tpsap, blk, list, out_stride, type1_bits;
};
// rows as the decoder writes them (tetra_lower_mac.c:58-105: type2 / type1 bits) and the frame list a kind's rows come from
constexpr KindInfo kKinds[TETRA_RX_N_KINDS] = {
    { TETRA_TPSAP_T_SB1, 1, TETRA_LIST_SYNC, 80, 60 },       // SB1
    { TETRA_TPSAP_T_BBK, 0, TETRA_LIST_ANY, 32, 30 },        // BBK
    { TETRA_TPSAP_T_SB2, 2, TETRA_LIST_SYNC, 144, 124 },     // SB2
    { TETRA_TPSAP_T_NDB, 1, TETRA_LIST_NORM_2, 144, 124 },   // NDB blk 1
    { TETRA_TPSAP_T_NDB, 2, TETRA_LIST_NORM_2, 144, 124 },   // NDB blk 2
    { TETRA_TPSAP_T_SCH_F, 0, TETRA_LIST_NORM_1, 288, 268 }, // SCH/F
};
// the second decode launch's job order: long blocks first, so that the short ones fill the machine while the long ones finish
constexpr int kJobOrder[] = { TETRA_RX_KIND_SCH_F, TETRA_RX_KIND_SB2, TETRA_RX_KIND_NDB1, TETRA_RX_KIND_NDB2, TETRA_RX_KIND_BBK };

static_assert(sizeof(tetra_rx_block_t) == sizeof(tetra_lmac_label_t), "tetra_rx_block_t is the decoder's row label");

// the type-1 bits of the first n rows, packed: out[j][0 .. nb) = t2[j][0 .. nb) (two bytes per thread: every kind's count is even)
__global__ __launch_bounds__(256) void k_rx_pack_type1(const uint8_t* __restrict__ t2, int in_stride, int nb, int n, uint8_t* __restrict__ out) {
    const int half = nb >> 1;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n * half) return;
    const int j = (int)(i / half), u = (int)(i - (long long)j * half);
    reinterpret_cast<uint16_t*>(out)[i] = reinterpret_cast<const uint16_t*>(t2 + (size_t)j * in_stride)[u];
}

struct Guard {
    int prev = -1;
    bool ok;
    explicit Guard(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; ok = hipSetDevice(d) == hipSuccess; }
    ~Guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct KindBufs {                 // one parity's results of one kind
    uint8_t* t2 = nullptr;        // [rows][out_stride]
    int32_t* ok = nullptr;        // [rows]
    tetra_rx_block_t* blocks = nullptr;   // [rows]
    // into the parity's frame lists (not owned): the kind's rows are the frames row_frame[0 .. *n_rows)
    const int32_t* row_frame = nullptr;
    const int32_t* n_rows = nullptr;
};

}  // namespace

struct tetra_rx {
    tetra_rx_config_t cfg;
    int device = 0, last_hip = 0;
    int C = 0, F = 0, rows = 0, stride = 0, kinds = 0;
    bool one_stream = false;
    tetra_demod_t* dem = nullptr;
    tetra_bsync_t* bs = nullptr;
    hipStream_t tail = nullptr;
    hipStream_t fetch_s = nullptr;        // tetra_rx_fetch's pack + copy (never behind a queued tail)
    // per call parity
    uint8_t* bits[2] = { nullptr, nullptr };
    int32_t* nbits[2] = { nullptr, nullptr };
    KindBufs res[2][TETRA_RX_N_KINDS];
    int32_t* lists[2] = { nullptr, nullptr };         // [TETRA_N_LISTS][rows] frame lists
    int32_t* counts[2] = { nullptr, nullptr };        // [TETRA_N_LISTS]
    hipEvent_t ev_demod[2] = { nullptr, nullptr }, ev_tail[2] = { nullptr, nullptr };
    // the tail's working set (one: tails run one after the other on one stream)
    uint32_t* frames = nullptr;           // [rows][16] packed frames
    int32_t* ft = nullptr;                // [rows] frame types
    uint32_t* fb = nullptr;               // [rows] frame bit numbers
    int32_t* nf = nullptr;                // [C]
    int32_t* chan_first = nullptr;        // [TETRA_N_LISTS][C] position in each list of a channel's first entry
    int32_t* index_work = nullptr;        // tetra_burst_index_device's scratch
    void* lmac_ws = nullptr;              // the decoder's decision scratch for the launch of every other kind
    size_t lmac_ws_bytes = 0;
    uint8_t* fetch_stage = nullptr;       // tetra_rx_fetch: a kind's type-1 bits packed row after row (allocated on first use)
    uint32_t *row_scramb = nullptr, *row_time_rx = nullptr, *row_time = nullptr;
    tetra_lmac_cell_state_t* cell = nullptr;   // [C]
    float* st_iq = nullptr;               // host-path staging
    hipEvent_t ev_stage[4] = { nullptr, nullptr, nullptr, nullptr };
    long long calls = 0;
    bool stage_valid = false;
};

#define RX_TRY(h, expr)                                   \
    do {                                                  \
        hipError_t e__ = (expr);                          \
        if (e__ != hipSuccess) {                          \
            (h)->last_hip = (int)e__;                     \
            return TETRA_ERR_HIP;                         \
        }                                                 \
    } while (0)
#define RX_OK(expr)                                       \
    do {                                                  \
        const int rc__ = (expr);                          \
        if (rc__ != TETRA_OK) return rc__;                \
    } while (0)

namespace {

void free_all(tetra_rx* h) {
    if (h->dem) (void)tetra_demod_destroy(h->dem);
    if (h->bs) (void)tetra_bsync_destroy(h->bs);
    void* ptrs[] = { h->bits[0], h->bits[1], h->nbits[0], h->nbits[1], h->lists[0], h->lists[1], h->counts[0], h->counts[1], h->frames, h->ft,
                     h->fb, h->nf, h->chan_first, h->index_work, h->lmac_ws, h->fetch_stage, h->row_scramb, h->row_time_rx, h->row_time, h->cell, h->st_iq };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& par : h->res)
        for (auto& k : par) {
            void* q[] = { k.t2, k.ok, k.blocks };
            for (void* p : q) if (p) (void)hipFree(p);
        }
    for (auto* evs : { h->ev_demod, h->ev_tail })
        for (int i = 0; i < 2; i++) if (evs[i]) (void)hipEventDestroy(evs[i]);
    for (auto& e : h->ev_stage) if (e) (void)hipEventDestroy(e);
    if (h->tail) (void)hipStreamDestroy(h->tail);
    if (h->fetch_s) (void)hipStreamDestroy(h->fetch_s);
}

template <typename T> bool dalloc(T*& p, size_t count) { return hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * (count ? count : 1)) == hipSuccess; }

int zero_results(tetra_rx* h) {
    for (int b = 0; b < 2; b++) RX_TRY(h, hipMemset(h->counts[b], 0, sizeof(int32_t) * TETRA_N_LISTS));
    RX_TRY(h, hipMemset(h->cell, 0, sizeof(tetra_lmac_cell_state_t) * (size_t)h->C));
    for (int b = 0; b < 2; b++) RX_TRY(h, hipMemset(h->nbits[b], 0, sizeof(int32_t) * (size_t)h->C));
    return TETRA_OK;
}

// the tail of one call on stream s (see the header of this file)
int enqueue_tail(tetra_rx* h, int b, hipStream_t s) {
    const int n = h->rows;
    RX_TRY(h, hipEventRecord(h->ev_stage[0], s));
    RX_OK(tetra_bsync_process_packed_device(h->bs, h->bits[b], h->stride, h->nbits[b], h->frames, h->ft, h->fb, h->nf, s));
    RX_TRY(h, hipEventRecord(h->ev_stage[1], s));
    RX_OK(tetra_burst_index_device(h->ft, n, h->F, h->lists[b], h->counts[b], h->chan_first, h->index_work, s));
    tetra_lmac_frames_t src = {};
    src.d_frames = h->frames;
    src.d_frame_type = h->ft;
    src.n_frames = n;
    src.frames_per_channel = h->F;
    src.d_frame_bitnum = h->fb;
    src.d_time_rx = h->row_time_rx;
    src.d_time = h->row_time;
    src.d_workspace = h->lmac_ws;
    src.workspace_bytes = h->lmac_ws_bytes;
    auto job_of = [&](int k, bool labels) {
        const KindInfo& ki = kKinds[k];
        const KindBufs& r = h->res[b][k];
        tetra_lmac_job_t j = {};
        j.type = ki.tpsap;
        j.blk_num = ki.blk;
        j.d_row_frame = r.row_frame;
        j.d_n_rows = r.n_rows;
        j.max_rows = n;
        j.out_stride = ki.out_stride;
        j.d_frame_scramb = h->row_scramb;
        j.d_type2 = r.t2;
        j.d_crc_ok = r.ok;
        j.d_labels = labels ? reinterpret_cast<tetra_lmac_label_t*>(r.blocks) : nullptr;
        return j;
    };
    {   // SB1 first: its SYNC PDUs set the code and the clock for everything else in the same burst (tetra_lower_mac.c:246-275)
        const KindBufs& r = h->res[b][TETRA_RX_KIND_SB1];
        const tetra_lmac_job_t j = job_of(TETRA_RX_KIND_SB1, false);
        RX_OK(tetra_lmac_decode_frames_device(&src, &j, 1, s));
        RX_OK(tetra_lmac_track_sync_lists_device(r.t2, kKinds[TETRA_RX_KIND_SB1].out_stride, r.ok, h->ft, h->nf,
                                                 h->chan_first + (size_t)TETRA_LIST_SYNC * h->C, h->C, h->F, h->cell, h->row_scramb, h->row_time_rx,
                                                 h->row_time, h->fb, reinterpret_cast<tetra_lmac_label_t*>(r.blocks), s));
    }
    RX_TRY(h, hipEventRecord(h->ev_stage[2], s));
    tetra_lmac_job_t jobs[TETRA_RX_N_KINDS];
    int nj = 0;
    for (int k : kJobOrder)
        if (h->kinds & (1 << k)) jobs[nj++] = job_of(k, true);
    RX_OK(tetra_lmac_decode_frames_device(&src, jobs, nj, s));
    RX_TRY(h, hipEventRecord(h->ev_stage[3], s));
    return TETRA_OK;
}

// parity of the call `which` calls back (0 = latest); -1 if there is no such call yet
int parity_of(const tetra_rx* h, int which) {
    if (which < 0 || which > 1 || h->calls <= which) return -1;
    return (int)((h->calls - 1 - which) & 1);
}

}  // namespace

extern "C" {

int tetra_rx_default_config(tetra_rx_config_t* cfg) {
    if (!cfg) return TETRA_ERR_ARG;
    std::memset(cfg, 0, sizeof(*cfg));
    return tetra_demod_default_config(&cfg->demod);
}

int tetra_rx_type1_bits(int kind) {
    return kind < 0 || kind >= TETRA_RX_N_KINDS ? TETRA_ERR_ARG : kKinds[kind].type1_bits;
}

int tetra_rx_create(const tetra_rx_config_t* cfg, tetra_rx_t** out) {
    if (!cfg || !out) return TETRA_ERR_ARG;
    *out = nullptr;
    if ((cfg->kinds & ~((1 << TETRA_RX_N_KINDS) - 1)) || (cfg->flags & ~TETRA_RX_FLAG_ONE_STREAM)) return TETRA_ERR_ARG;
    tetra_rx* h = new (std::nothrow) tetra_rx();
    if (!h) return TETRA_ERR_NOMEM;
    h->cfg = *cfg;
    h->cfg.demod.rrc_taps = h->cfg.demod.bandedge_taps = h->cfg.demod.interp_bank = nullptr;
    h->kinds = (cfg->kinds ? cfg->kinds : (1 << TETRA_RX_N_KINDS) - 1) | (1 << TETRA_RX_KIND_SB1);
    h->one_stream = (cfg->flags & TETRA_RX_FLAG_ONE_STREAM) != 0;
    int rc = tetra_demod_create(&cfg->demod, &h->dem);
    if (rc != TETRA_OK) { delete h; return rc; }
    h->C = cfg->demod.n_channels;
    int dev = cfg->demod.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { free_all(h); delete h; return TETRA_ERR_NO_DEVICE; }
    h->device = dev;
    Guard g(dev);
    if (!g.ok) { free_all(h); delete h; return TETRA_ERR_NO_DEVICE; }
    h->stride = tetra_demod_bits_stride_for(h->dem, cfg->demod.max_samples);
    if (h->stride < 0) { rc = h->stride; free_all(h); delete h; return rc; }
    rc = tetra_bsync_create(h->C, h->stride, dev, &h->bs);
    if (rc != TETRA_OK) { free_all(h); delete h; return rc; }
    h->F = tetra_bsync_max_frames(h->bs);
    const long long rows = (long long)h->C * h->F;
    if (rows > 0x7fffffffLL / 512) { free_all(h); delete h; return TETRA_ERR_SIZE; }      // 32-bit row / byte indices downstream
    h->rows = (int)rows;
    const size_t n = (size_t)rows;
    bool ok = hipStreamCreateWithFlags(&h->tail, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->fetch_s, hipStreamNonBlocking) == hipSuccess;
    for (int b = 0; b < 2 && ok; b++) {
        ok = dalloc(h->bits[b], (size_t)h->C * h->stride) && dalloc(h->nbits[b], (size_t)h->C) && dalloc(h->lists[b], (size_t)TETRA_N_LISTS * n) &&
             dalloc(h->counts[b], (size_t)TETRA_N_LISTS) && hipEventCreateWithFlags(&h->ev_demod[b], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&h->ev_tail[b], hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < TETRA_RX_N_KINDS && ok; k++) {
            if (!(h->kinds & (1 << k))) continue;
            KindBufs& r = h->res[b][k];
            ok = dalloc(r.t2, n * kKinds[k].out_stride) && dalloc(r.ok, n) && dalloc(r.blocks, n);
            r.row_frame = h->lists[b] + (size_t)kKinds[k].list * n;
            r.n_rows = h->counts[b] + kKinds[k].list;
        }
    }
    ok = ok && dalloc(h->frames, n * TETRA_FRAME_WORDS) && dalloc(h->ft, n) && dalloc(h->fb, n) && dalloc(h->nf, (size_t)h->C) &&
         dalloc(h->chan_first, (size_t)TETRA_N_LISTS * h->C) && dalloc(h->index_work, (size_t)TETRA_N_LISTS * ((n + 255) / 256)) &&
         dalloc(h->row_scramb, n) && dalloc(h->row_time_rx, n) && dalloc(h->row_time, n) && dalloc(h->cell, (size_t)h->C);
    for (auto& e : h->ev_stage) ok = ok && hipEventCreate(&e) == hipSuccess;
    if (ok) {      // the decision scratch of the two decode launches (they run one after the other), sized for the worst case (every frame slot a row of every kind)
        tetra_lmac_job_t jobs[TETRA_RX_N_KINDS] = {};
        int nj = 0;
        for (int k : kJobOrder)
            if (h->kinds & (1 << k)) { jobs[nj].type = kKinds[k].tpsap; jobs[nj].blk_num = kKinds[k].blk; jobs[nj].max_rows = h->rows; nj++; }
        h->lmac_ws_bytes = tetra_lmac_decode_frames_workspace_bytes(jobs, nj);
        tetra_lmac_job_t sb1 = {};
        sb1.type = TETRA_TPSAP_T_SB1; sb1.blk_num = 1; sb1.max_rows = h->rows;
        const size_t sb1_bytes = tetra_lmac_decode_frames_workspace_bytes(&sb1, 1);
        h->lmac_ws_bytes = sb1_bytes > h->lmac_ws_bytes ? sb1_bytes : h->lmac_ws_bytes;
        ok = h->lmac_ws_bytes == 0 || hipMalloc(&h->lmac_ws, h->lmac_ws_bytes) == hipSuccess;
    }
    rc = ok ? zero_results(h) : TETRA_ERR_NOMEM;
    if (rc != TETRA_OK) { free_all(h); delete h; return rc; }
    *out = h;
    return TETRA_OK;
}

int tetra_rx_destroy(tetra_rx_t* h) {
    if (!h) return TETRA_ERR_ARG;
    Guard g(h->device);
    (void)hipDeviceSynchronize();
    free_all(h);
    delete h;
    return TETRA_OK;
}

int tetra_rx_reset(tetra_rx_t* h) {
    if (!h) return TETRA_ERR_ARG;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    RX_TRY(h, hipDeviceSynchronize());
    RX_OK(tetra_demod_reset(h->dem, -1));
    RX_OK(tetra_bsync_reset(h->bs));
    RX_OK(zero_results(h));
    h->calls = 0;
    h->stage_valid = false;
    return TETRA_OK;
}

int tetra_rx_process_device(tetra_rx_t* h, const float* d_iq, int n_samples, void* hip_stream) {
    if (!h || (!d_iq && n_samples > 0)) return TETRA_ERR_ARG;
    if (n_samples < 0 || n_samples > h->cfg.demod.max_samples) return TETRA_ERR_SIZE;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    hipStream_t sa = static_cast<hipStream_t>(hip_stream);
    hipStream_t sb = h->one_stream ? sa : h->tail;
    const int b = (int)(h->calls & 1);
    // the bit rows of this parity were last read by the tail of call k - 2
    if (h->calls >= 2 && !h->one_stream) RX_TRY(h, hipStreamWaitEvent(sa, h->ev_tail[b], 0));
    RX_OK(tetra_demod_process_device(h->dem, d_iq, n_samples, h->bits[b], h->stride, h->nbits[b], nullptr, sa));
    RX_TRY(h, hipEventRecord(h->ev_demod[b], sa));
    if (!h->one_stream) RX_TRY(h, hipStreamWaitEvent(sb, h->ev_demod[b], 0));
    h->calls++;                     // the call exists from here on: a failing tail leaves its rows undefined, not the bookkeeping
    h->stage_valid = false;
    RX_OK(enqueue_tail(h, b, sb));
    RX_TRY(h, hipEventRecord(h->ev_tail[b], sb));
    h->stage_valid = true;
    return TETRA_OK;
}

int tetra_rx_process(tetra_rx_t* h, const float* iq, int n_samples) {
    if (!h || (!iq && n_samples > 0)) return TETRA_ERR_ARG;
    if (n_samples < 0 || n_samples > h->cfg.demod.max_samples) return TETRA_ERR_SIZE;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    const size_t bytes = sizeof(float) * 2 * (size_t)h->C * (size_t)n_samples;
    if (!h->st_iq) {
        if (hipMalloc(reinterpret_cast<void**>(&h->st_iq), sizeof(float) * 2 * (size_t)h->C * (size_t)h->cfg.demod.max_samples) != hipSuccess)
            return TETRA_ERR_NOMEM;
    }
    // the staging buffer is read by the demodulator launch of the previous call: wait for it before overwriting
    RX_TRY(h, hipStreamSynchronize(nullptr));
    if (bytes) RX_TRY(h, hipMemcpy(h->st_iq, iq, bytes, hipMemcpyHostToDevice));
    return tetra_rx_process_device(h, h->st_iq, n_samples, nullptr);
}

int tetra_rx_wait(tetra_rx_t* h) {
    if (!h) return TETRA_ERR_ARG;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    for (int b = 0; b < 2; b++)
        if (h->calls > b) {
            RX_TRY(h, hipEventSynchronize(h->ev_demod[b]));
            RX_TRY(h, hipEventSynchronize(h->ev_tail[b]));
        }
    long long over = 0;
    RX_OK(tetra_demod_get_overruns(h->dem, &over));
    return over > 0 ? TETRA_ERR_OVERRUN : TETRA_OK;
}

int tetra_rx_max_rows(tetra_rx_t* h) { return h ? h->rows : TETRA_ERR_ARG; }

int tetra_rx_fetch(tetra_rx_t* h, int which, int kind, tetra_rx_block_t* blocks, uint8_t* type1, int type1_stride, int capacity, int* n_rows) {
    if (!h || !n_rows || kind < 0 || kind >= TETRA_RX_N_KINDS || which < 0 || which > 1 || capacity < 0) return TETRA_ERR_ARG;
    if (!(h->kinds & (1 << kind))) return TETRA_ERR_UNSUPPORTED;
    if (type1 && type1_stride < kKinds[kind].type1_bits) return TETRA_ERR_SIZE;
    *n_rows = 0;
    const int b = parity_of(h, which);
    if (b < 0) return TETRA_OK;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    RX_TRY(h, hipEventSynchronize(h->ev_tail[b]));
    const KindBufs& r = h->res[b][kind];
    int32_t n = 0;
    RX_TRY(h, hipMemcpy(&n, r.n_rows, sizeof(n), hipMemcpyDeviceToHost));
    if (n < 0 || n > h->rows) return TETRA_ERR_HIP;          // (cannot happen: the demultiplexer counts at most `rows` frames)
    *n_rows = n;
    if (n > capacity) return (blocks || type1) ? TETRA_ERR_SIZE : TETRA_OK;
    if (n == 0) return TETRA_OK;
    if (blocks) RX_TRY(h, hipMemcpy(blocks, r.blocks, sizeof(tetra_rx_block_t) * (size_t)n, hipMemcpyDeviceToHost));
    if (type1) {
        const int nb = kKinds[kind].type1_bits;
        if (type1_stride == nb) {
            // contiguous rows at the caller's: pack on the device, ONE copy (a strided device-to-host copy of 10^5 narrow rows moves
            // ~50 MB/s: 1.9 s for a second of 4096 channels' blocks, measured; this way the link's rate)
            if (!h->fetch_stage && hipMalloc(reinterpret_cast<void**>(&h->fetch_stage), (size_t)h->rows * 268) != hipSuccess) {
                (void)hipGetLastError();
                return TETRA_ERR_NOMEM;
            }
            const long long units = (long long)n * (nb >> 1);
            hipLaunchKernelGGL(k_rx_pack_type1, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, h->fetch_s, r.t2, kKinds[kind].out_stride, nb, n,
                               h->fetch_stage);
            RX_TRY(h, hipGetLastError());
            RX_TRY(h, hipMemcpyAsync(type1, h->fetch_stage, (size_t)n * nb, hipMemcpyDeviceToHost, h->fetch_s));
            RX_TRY(h, hipStreamSynchronize(h->fetch_s));
        } else {
            RX_TRY(h, hipMemcpy2D(type1, (size_t)type1_stride, r.t2, (size_t)kKinds[kind].out_stride, (size_t)nb, (size_t)n, hipMemcpyDeviceToHost));
        }
    }
    return TETRA_OK;
}

int tetra_rx_rows_device(tetra_rx_t* h, int which, int kind, const uint8_t** d_type2, int* type2_stride, const tetra_rx_block_t** d_blocks,
                         const int32_t** d_n_rows, void* hip_stream) {
    if (!h || kind < 0 || kind >= TETRA_RX_N_KINDS || which < 0 || which > 1) return TETRA_ERR_ARG;
    if (!(h->kinds & (1 << kind))) return TETRA_ERR_UNSUPPORTED;
    const int b = parity_of(h, which);
    if (b < 0) return TETRA_ERR_ARG;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    RX_TRY(h, hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), h->ev_tail[b], 0));
    const KindBufs& r = h->res[b][kind];
    if (d_type2) *d_type2 = r.t2;
    if (type2_stride) *type2_stride = kKinds[kind].out_stride;
    if (d_blocks) *d_blocks = r.blocks;
    if (d_n_rows) *d_n_rows = r.n_rows;
    return TETRA_OK;
}

int tetra_rx_get_cell(tetra_rx_t* h, int first, int count, tetra_lmac_cell_state_t* out) {
    if (!h || !out || first < 0 || count < 0 || first + (long long)count > h->C) return TETRA_ERR_ARG;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    if (h->calls > 0) RX_TRY(h, hipEventSynchronize(h->ev_tail[(h->calls - 1) & 1]));
    if (count) RX_TRY(h, hipMemcpy(out, h->cell + first, sizeof(tetra_lmac_cell_state_t) * (size_t)count, hipMemcpyDeviceToHost));
    return TETRA_OK;
}

int tetra_rx_get_sync_state(tetra_rx_t* h, int first, int count, tetra_bsync_state_t* out) {
    if (!h || !out || first < 0 || count < 0 || first + (long long)count > h->C) return TETRA_ERR_ARG;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    if (h->calls > 0) RX_TRY(h, hipEventSynchronize(h->ev_tail[(h->calls - 1) & 1]));
    return count ? tetra_bsync_get_state(h->bs, first, count, out) : TETRA_OK;
}

int tetra_rx_bits_device(tetra_rx_t* h, int which, const uint8_t** d_bits, int* bits_stride, const int32_t** d_n_bits, void* hip_stream) {
    if (!h || which < 0 || which > 1) return TETRA_ERR_ARG;
    const int b = parity_of(h, which);
    if (b < 0) return TETRA_ERR_ARG;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    RX_TRY(h, hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), h->ev_demod[b], 0));
    if (d_bits) *d_bits = h->bits[b];
    if (bits_stride) *bits_stride = h->stride;
    if (d_n_bits) *d_n_bits = h->nbits[b];
    return TETRA_OK;
}

tetra_demod_t* tetra_rx_demod(tetra_rx_t* h) { return h ? h->dem : nullptr; }

int tetra_rx_stage_ms(tetra_rx_t* h, float ms[4]) {
    if (!h || !ms || !h->stage_valid) return TETRA_ERR_ARG;
    Guard g(h->device);
    if (!g.ok) return TETRA_ERR_NO_DEVICE;
    RX_TRY(h, hipEventSynchronize(h->ev_stage[3]));
    RX_OK(tetra_demod_last_kernel_ms(h->dem, &ms[0]));
    for (int i = 0; i < 3; i++) RX_TRY(h, hipEventElapsedTime(&ms[1 + i], h->ev_stage[i], h->ev_stage[i + 1]));
    return TETRA_OK;
}

}  // extern "C"
