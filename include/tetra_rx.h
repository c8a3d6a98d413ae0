/*
 * tetra_rx.h -- C ABI of the whole device-resident receive chain behind ONE handle (round 6).
 *
 * The reference's receive side is one entry point per receiver: the plugin's decoder block feeds the demodulator's bits to
 *     tetra_burst_sync_in()    src/decoder/src/phy/tetra_burst_sync.c:54-155   (find the training sequences, lock, cut frames)
 *  -> tetra_burst_rx_cb()      src/decoder/src/phy/tetra_burst.c:343-393        (split a frame into its logical-channel blocks)
 *  -> tp_sap_udata_ind()       src/decoder/src/lower_mac/tetra_lower_mac.c:148-237  (descramble, deinterleave, depuncture, Viterbi, CRC)
 * with the cell state fed back inside the same call chain (:246-275: a SYNC PDU with a good CRC sets the scrambling code every
 * later block -- starting with the BBK and SB2 of the same burst -- is descrambled with, and the TDMA time).  This library has those
 * stages as separate C ABIs (tetra_demod.h, tetra_burst_sync.h, tetra_lmac.h) for callers that want one of them; a caller that
 * wants the chain had to order ~12 device calls per block of signal, size their buffers and hand the tracker's output to the next
 * decode itself.  This header is that chain as the reference has it -- IQ in, decoded type-1 blocks with CRC verdict, TDMA time and
 * cell state out -- for C channels at once:
 *
 *     tetra_rx_process_device(h, d_iq, n)      enqueue:  demodulator (block k) on the caller's stream  ||  on the handle's own
 *                                              stream, for block k: synchroniser -> frame lists -> SB1 decoded straight from the
 *                                              packed frames -> SYNC-PDU tracker (cell, per-slot scrambling code + TDMA time) ->
 *                                              every other block kind decoded straight from the frames with the tracker's codes,
 *                                              labelled, in ONE launch (7 launches per call; no byte rows between the stages)
 *     tetra_rx_fetch(h, which, kind, ...)      the decoded blocks of one kind of the latest (which = 0) or the previous (1) call
 *     tetra_rx_get_cell(h, first, count, ..)   tcd / t_phy_state of channels (tetra_lmac_cell_state_t)
 *
 * The demodulator of call k+1 overlaps the tail of call k (two streams, events in between; bit rows and results are double
 * buffered), which is where the chain's steady-state rate comes from.  Everything the stages guarantee on their own
 * carries over unchanged: bits = the demodulator contract (tetra_demod.h), frames = the reference's one-bit-per-call behaviour
 * (tetra_burst_sync.h), blocks bit-exact with the reference's lower-MAC primitives (tetra_lmac.h); the tests compare this handle's
 * output with the reference's own encoder / burst builders / decoder (oracle/_ref) exactly as they do for the separate stages.
 * Same conventions: extern "C", int status (TETRA_OK / TETRA_ERR_*), no exceptions, one thread per handle, GPU only; every
 * mis-sized buffer is a status.
 */
#ifndef TETRA_RX_H
#define TETRA_RX_H

#include <stdint.h>

#include "tetra_burst_sync.h"
#include "tetra_demod.h"
#include "tetra_lmac.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The logical-channel blocks tetra_burst_rx_cb hands to tp_sap_udata_ind (tetra_burst.c:355-391), in the order a SYNC burst's come
 * (SB1, BBK, SB2); a NORM_1 burst carries SCH/F + BBK, a NORM_2 burst NDB blk 1 + blk 2 + BBK. */
enum {
    TETRA_RX_KIND_SB1 = 0,     /* TPSAP_T_SB1, BLK_1: BSCH, the SYNC PDU (60 type-1 bits) */
    TETRA_RX_KIND_BBK = 1,     /* TPSAP_T_BBK: AACH, 30 descrambled bits (pass-through like the reference, tetra_lower_mac.c:231-236) */
    TETRA_RX_KIND_SB2 = 2,     /* TPSAP_T_SB2, BLK_2 (124 type-1 bits) */
    TETRA_RX_KIND_NDB1 = 3,    /* TPSAP_T_NDB, BLK_1 (124) */
    TETRA_RX_KIND_NDB2 = 4,    /* TPSAP_T_NDB, BLK_2 (124) */
    TETRA_RX_KIND_SCH_F = 5,   /* TPSAP_T_SCH_F (268) */
    TETRA_RX_N_KINDS = 6
};

enum {
    TETRA_RX_FLAG_ONE_STREAM = 1   /* run the tail on the caller's stream behind the demodulator (no overlap between calls); A/B, tests */
};

typedef struct tetra_rx_config {
    tetra_demod_config_t demod;    /* channel count, max_samples, layout, device, the ten PI4DQPSK::init arguments, TETRA_FLAG_* */
    int32_t kinds;                 /* bit mask (1 << TETRA_RX_KIND_*) of the block kinds to decode; 0 = all.  SB1 is always decoded:
                                      the chain's scrambling codes and clock come from it */
    int32_t flags;                 /* TETRA_RX_FLAG_* */
} tetra_rx_config_t;

/* One decoded block = one tp_sap_udata_ind hand-over to the upper MAC. */
typedef struct tetra_rx_block {
    int32_t channel;
    int32_t frame_slot;            /* index of the block's frame among the frames its channel consumed in this call */
    uint32_t bitnum;               /* absolute bit number of the frame's first bit in the channel's bit stream (bitbuf_start_bitnum) */
    uint32_t tdma_time_rx;         /* t_phy_state.time when tetra_burst_rx_cb is entered for the frame: tn | fn << 8 | mn << 16 */
    uint32_t tdma_time;            /* ... after the frame's SB1 block, if it has one: the time every later block of the burst is handled under */
    int32_t crc_ok;                /* tup->crc_ok */
} tetra_rx_block_t;

typedef struct tetra_rx tetra_rx_t;

int tetra_rx_default_config(tetra_rx_config_t* cfg);      /* tetra_demod_default_config (the plugin's parameters, C = 1), all kinds */
int tetra_rx_create(const tetra_rx_config_t* cfg, tetra_rx_t** out);
int tetra_rx_destroy(tetra_rx_t* h);
/* Fresh receivers: demodulator loops (tetra_demod_reset), synchroniser UNLOCKED and empty, cell state zero.  Synchronises. */
int tetra_rx_reset(tetra_rx_t* h);

/* d_iq: n_channels x n_samples complex64 in cfg.demod.layout (device pointer; read by the demodulator launch enqueued on
 * hip_stream -- the caller may reuse it once that stream has passed the call).  n_samples <= cfg.demod.max_samples
 * (TETRA_ERR_SIZE).  Returns without synchronising; at most two calls are in flight (the third waits for the first one's tail on
 * the device, not on the host). */
int tetra_rx_process_device(tetra_rx_t* h, const float* d_iq, int n_samples, void* hip_stream);
/* Host-pointer variant: copies the samples in (synchronously), then the same. */
int tetra_rx_process(tetra_rx_t* h, const float* iq, int n_samples);
/* Blocks until everything enqueued has run.  TETRA_ERR_OVERRUN if the demodulator cut a channel off (tetra_demod.h). */
int tetra_rx_wait(tetra_rx_t* h);

/* Upper bound of the rows a fetch of one kind can return: n_channels x frames per call ((4096 + bits per call) / 510 + 2).  Device
 * memory of a handle: about 5 KB per such row with every kind enabled (results and labels of two calls, the decoder's decision scratch,
 * frames, the two bit-row buffers): 1.4 GB for 4096 channels x 36000 samples per call. */
int tetra_rx_max_rows(tetra_rx_t* h);
/* type-1 bits per block of a kind (60 / 30 / 124 / 124 / 124 / 268); < 0: TETRA_ERR_ARG. */
int tetra_rx_type1_bits(int kind);
/*
 * The decoded blocks of one kind from the latest process call (which = 0) or the one before it (which = 1), in (channel, frame)
 * order.  Waits for that call's tail (only).
 *   blocks   [capacity] host, may be NULL
 *   type1    [capacity][type1_stride] uint8 host, one bit per byte, tetra_rx_type1_bits(kind) per row, may be NULL;
 *            type1_stride >= that count or TETRA_ERR_SIZE.  Contiguous rows (type1_stride == the count) are packed on the device and
 *            come over in ONE copy; any other stride is a strided device-to-host copy, orders of magnitude slower for 10^5 rows
 *   *n_rows  rows available; more than capacity: TETRA_ERR_SIZE and nothing is copied (call again with room for *n_rows)
 * A kind that the configuration does not decode: TETRA_ERR_UNSUPPORTED.  Before the first call / which = 1 before the second: 0 rows.
 */
int tetra_rx_fetch(tetra_rx_t* h, int which, int kind, tetra_rx_block_t* blocks, uint8_t* type1, int type1_stride, int capacity,
                   int* n_rows);
/* The same rows where they are, for consumers on the device: d_type2 [*][*type2_stride] (the first tetra_rx_type1_bits(kind) of a
 * row are the type-1 bits), d_blocks [*], d_n_rows [1].  Valid until the next-but-one process call; ordered behind the tail of the
 * call they belong to: `hip_stream` is made to wait for it (no host synchronisation).  Any out pointer may be NULL. */
int tetra_rx_rows_device(tetra_rx_t* h, int which, int kind, const uint8_t** d_type2, int* type2_stride,
                         const tetra_rx_block_t** d_blocks, const int32_t** d_n_rows, void* hip_stream);

/* tcd / t_phy_state of channels [first, first + count) after the latest call (waits for its tail). */
int tetra_rx_get_cell(tetra_rx_t* h, int first, int count, tetra_lmac_cell_state_t* out);
/* The synchronisers' states (tetra_rx_state: UNLOCKED / KNOW_FSTART / LOCKED, buffer positions) after the latest call. */
int tetra_rx_get_sync_state(tetra_rx_t* h, int first, int count, tetra_bsync_state_t* out);
/* The demodulator bits of the latest (which = 0) / previous (1) call where they are: d_bits [C][*bits_stride], d_n_bits [C] -- the
 * NETSYMS payload (src/main.cpp:387-389).  Ordered like tetra_rx_rows_device. */
int tetra_rx_bits_device(tetra_rx_t* h, int which, const uint8_t** d_bits, int* bits_stride, const int32_t** d_n_bits, void* hip_stream);
/* The demodulator inside, for the PI4DQPSK setters, the quality / constellation taps and checkpoints (tetra_demod_set_param,
 * _get_quality, _get_state ...).  Call tetra_rx_wait first; never destroy it. */
tetra_demod_t* tetra_rx_demod(tetra_rx_t* h);
/* GPU time (ms) of the latest call's stages from HIP events on their streams: ms[0] demodulator launch, ms[1] synchroniser,
 * ms[2] frame lists + SB1 decode + tracker, ms[3] the other kinds' decode + labels (waits for the tail). */
int tetra_rx_stage_ms(tetra_rx_t* h, float ms[4]);

#ifdef __cplusplus
}
#endif
#endif
