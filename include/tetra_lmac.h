/*
 * tetra_lmac.h -- C ABI of the batched lower-MAC channel decoding.
 *
 * Replaces, for n_blocks logical-channel blocks at once, the channel-decoding half of the reference's
 *     void tp_sap_udata_ind(enum tp_sap_data_type type, int blk_num, const uint8_t *bits, unsigned int len, void *priv)
 * (src/decoder/src/lower_mac/tetra_lower_mac.c:148-240), i.e. its lines :181-236:
 *     type5 --tetra_scramb_bits (lower_mac/tetra_scramb.c:34-51, :77-85)--> type4
 *           --block_deinterleave (lower_mac/tetra_interleave.c:36-39, :51-59)--> type3
 *           --tetra_rcpc_depunct(TETRA_RCPC_PUNCT_2_3) (lower_mac/tetra_conv_enc.c:229-251)--> mother code, 0xff = punctured
 *           --viterbi_dec_sb1_wrapper (lower_mac/viterbi.c:6-25 -> viterbi_cch.c:59-67 -> osmo_conv.c osmo_conv_decode_acc,
 *             K=5 rate-1/4 code of EN 300 392-2 8.2.3.1.1, CONV_TERM_FLUSH)--> type2
 *           --crc16_ccitt_bits(type2, type1_bits+16) == TETRA_CRC_OK (lower_mac/crc_simple.c:103, tetra_common.h:330)--> crc_ok
 * with the block parameters of tetra_blk_param[] (tetra_lower_mac.c:58-105).  TPSAP_T_BBK follows the reference's
 * pass-through (:231-236: descramble only, crc_ok = 1; its Reed-Muller decode is a FIXME there).
 *
 * Bit-exact with the reference for ANY input bytes, including its treatment of non-0/1 bytes (after descrambling a byte
 * 0 is a strong 0, 0xff is an erasure, anything else a strong 1 -- viterbi.c:12-23), the decoder's tie-breaks
 * (osmo_conv.c:63-95, even predecessor wins) and the four zero-metric flush steps it runs past the block
 * (osmo_conv.c:678-679 and :727-738, reading the zero-initialised tail of viterbi.c:8).  Parity for this entry point is PINNED: tests compare it with the reference
 * functions themselves, built from the reference's own source files into oracle/_ref (oracle/build_ref.sh), and with
 * the reference's encoder primitives as known-answer generator.
 *
 * What stays on the host: the SYNC-PDU field extraction (:246-275), TDMA time keeping and everything above the lower MAC.
 * Same conventions as tetra_demod.h: extern "C", int status (TETRA_OK / TETRA_ERR_*), no exceptions, GPU only.
 */
#ifndef TETRA_LMAC_H
#define TETRA_LMAC_H

#include <stddef.h>
#include <stdint.h>

#include "tetra_demod.h"

#ifdef __cplusplus
extern "C" {
#endif

/* enum tp_sap_data_type of src/decoder/src/phy/tetra_burst.h:9-16 */
enum {
    TETRA_TPSAP_T_SB1 = 0,
    TETRA_TPSAP_T_SB2 = 1,
    TETRA_TPSAP_T_NDB = 2,
    TETRA_TPSAP_T_BBK = 3,
    TETRA_TPSAP_T_SCH_HU = 4,
    TETRA_TPSAP_T_SCH_F = 5
};

/* struct tetra_blk_param of tetra_lower_mac.c:48-55 (without the name) */
typedef struct tetra_lmac_blk_param {
    int32_t type345_bits;
    int32_t type2_bits;
    int32_t type1_bits;
    int32_t interleave_a;
    int32_t have_crc16;
} tetra_lmac_blk_param_t;

int tetra_lmac_blk_param(int type, tetra_lmac_blk_param_t* out);
/* tetra_scramb_get_init (lower_mac/tetra_scramb.c:87-99): scrambling code of a cell from MCC, MNC and colour code. */
uint32_t tetra_lmac_scramb_init(uint16_t mcc, uint16_t mnc, uint8_t colour);

/*
 * type          TETRA_TPSAP_T_x, one block kind per call
 * d_type5       [n_blocks][in_stride] uint8, one bit per byte: the `bits` argument of tp_sap_udata_ind, type345_bits
 *               bytes used per row (device pointer, 4-byte aligned, in_stride a multiple of 4 and >= type345_bits)
 * d_scramb_init [n_blocks] uint32: the cell's scrambling code per block (tcd->scramb_init).  Ignored (may be NULL) for
 *               TETRA_TPSAP_T_SB1, which the reference always descrambles with SCRAMB_INIT = 3 (tetra_lower_mac.c:185-187)
 * d_type2       [n_blocks][out_stride] uint8 out: type2_bits decoded bits per row, one per byte; the first type1_bits
 *               are the type-1 payload the reference hands to the upper MAC (4-byte aligned, out_stride a multiple of
 *               4 and >= type2_bits)
 * d_crc_ok      [n_blocks] int32 out: the reference's tup->crc_ok (1/0)
 * Enqueued on hip_stream of the current device, no synchronisation.  n_blocks == 0 is a no-op.
 * Memory: the Viterbi decisions of a launch (n_blocks x (type2_bits + 4) x 2 bytes) live in a scratch taken from, and returned in
 * stream order to, a stream-ordered pool this library creates per device on first use and that keeps what is freed into it for
 * the life of the process (at most the largest launch's scratch; the device's default pool would hand the memory back to the driver
 * at every synchronisation point and map it again on the next call -- milliseconds).
 */
int tetra_lmac_decode_batch_device(int type, const uint8_t* d_type5, int n_blocks, int in_stride,
                                   const uint32_t* d_scramb_init, uint8_t* d_type2, int out_stride, int32_t* d_crc_ok,
                                   void* hip_stream);
/*
 * Counted form for rows that come out of tetra_burst_demux_compact_device: n_blocks is the CAPACITY of the row arrays, the
 * number of rows actually present is read on the device from *d_n_blocks (NULL: all n_blocks), and the scrambling code of
 * row j is d_scramb_init[d_init_index[j]] (d_init_index = the demultiplexer's d_row_frame, so that the per-frame-slot code
 * array of tetra_lmac_track_scramb_device is used as it is; NULL: d_scramb_init[j]).  Rows past the count are not touched.
 */
int tetra_lmac_decode_counted_device(int type, const uint8_t* d_type5, int n_blocks, const int32_t* d_n_blocks, int in_stride,
                                     const uint32_t* d_scramb_init, const int32_t* d_init_index, uint8_t* d_type2, int out_stride,
                                     int32_t* d_crc_ok, void* hip_stream);
/* Rows of plain bits (every byte 0 or 1 -- what this library's demultiplexers write) take a packed route inside the decoder (bytes ->
 * bits, whole words of the scrambling sequence, ~6 x fewer instructions in the front end); a workgroup of 64 rows with any other byte
 * value, or rows that are not 8-byte aligned (d_type5, in_stride) or further than 512 bytes apart, takes the byte route; the results
 * are the same bit for bit.
 * tetra_lmac_debug_force_byte_route(1) sends every row through the byte route (process-wide; tests and A/B); returns the old setting. */
int tetra_lmac_debug_force_byte_route(int on);
/*
 * Decoding straight from the burst synchroniser's PACKED frames, several block kinds in ONE launch (round 6).  The reference's
 * tetra_burst_rx_cb (src/decoder/src/phy/tetra_burst.c:343-393) cuts a frame into its blocks and hands each to tp_sap_udata_ind;
 * tetra_burst_demux_*_device + tetra_lmac_decode_*_device do the same through a byte row per block in HBM and one launch pair per
 * kind.  Here the decoder's front end reads the frame itself (64 bytes per lane instead of up to 432), cuts the kind's bits out of
 * the packed words with funnel shifts -- they are already the packed bits its trellis wants -- and one launch carries every job, so
 * that the short kinds fill the machine while the long ones run.  Results are bit for bit those of the two-step route.
 *
 * src     the frames of a call: d_frames [n_frames][TETRA_FRAME_WORDS = 16] uint32 (first bit most significant) and d_frame_type
 *         [n_frames] as tetra_bsync_process_packed_device writes them; for labels (optional, see d_labels): frames_per_channel and
 *         the per-frame-slot arrays d_frame_bitnum (synchroniser), d_time_rx / d_time (tetra_lmac_track_sync_device)
 * jobs    one per (type, blk_num): the rows are the frames listed in d_row_frame[0 .. *d_n_rows) (NULL count: max_rows), e.g. the
 *         lists of tetra_burst_index_device; a listed frame whose burst type does not carry the kind decodes as an all-zero block.
 *         d_frame_scramb [n_frames]: the scrambling code per FRAME SLOT (tetra_lmac_track_sync_device's d_row_scramb; NULL for
 *         TETRA_TPSAP_T_SB1, which always uses SCRAMB_INIT).  d_type2 [max_rows][out_stride] (8-byte aligned, out_stride a multiple
 *         of 8 and >= type2_bits; TETRA_TPSAP_T_BBK: 30 descrambled bits + 2 zero bytes, out_stride >= 32), d_crc_ok [max_rows].
 *         d_labels [max_rows], may be NULL: (channel, frame slot, bit number, TDMA times, crc_ok) per row -- needs src's label arrays.
 * Jobs run in the order given; put the long kinds first (SCH/F before SB2 / NDB before BBK).  At most TETRA_LMAC_MAX_JOBS.
 */
#define TETRA_LMAC_MAX_JOBS 8
typedef struct tetra_lmac_label {
    int32_t channel;
    int32_t frame_slot;            /* frame index within its channel: frame / frames_per_channel, frame % frames_per_channel */
    uint32_t bitnum;               /* d_frame_bitnum[frame] */
    uint32_t tdma_time_rx;         /* d_time_rx[frame] */
    uint32_t tdma_time;            /* d_time[frame] */
    int32_t crc_ok;
} tetra_lmac_label_t;
typedef struct tetra_lmac_frames {
    const uint32_t* d_frames;
    const int32_t* d_frame_type;
    int32_t n_frames;
    int32_t frames_per_channel;            /* labels only */
    const uint32_t* d_frame_bitnum;        /* labels only */
    const uint32_t* d_time_rx;             /* labels only */
    const uint32_t* d_time;                /* labels only */
    void* d_workspace;                     /* optional: the launch's decision scratch (tetra_lmac_decode_frames_workspace_bytes of the jobs;
                                              4-byte aligned); NULL: taken from / returned to the library's stream-ordered pool around the launch */
    size_t workspace_bytes;
} tetra_lmac_frames_t;
typedef struct tetra_lmac_job {
    int32_t type;                          /* TETRA_TPSAP_T_x */
    int32_t blk_num;                       /* 1 / 2 as tetra_burst_rx_cb numbers them (SB1: 1, SB2: 2, NDB: 1 or 2; BBK, SCH/F: ignored) */
    const int32_t* d_row_frame;
    const int32_t* d_n_rows;
    int32_t max_rows;
    int32_t out_stride;
    const uint32_t* d_frame_scramb;
    uint8_t* d_type2;
    int32_t* d_crc_ok;
    tetra_lmac_label_t* d_labels;
} tetra_lmac_job_t;
int tetra_lmac_decode_frames_device(const tetra_lmac_frames_t* src, const tetra_lmac_job_t* jobs, int n_jobs, void* hip_stream);
/* Bytes of decision scratch a launch of these jobs needs (2 bytes per trellis step and row of max_rows; 0 for AACH jobs).  A caller that hands the scratch in (d_workspace) saves the two stream-ordered pool
 * operations around the launch -- the receive chain does. */
size_t tetra_lmac_decode_frames_workspace_bytes(const tetra_lmac_job_t* jobs, int n_jobs);

/* Host-pointer variant (copies in/out, synchronises; device = HIP ordinal or -1 for the current one). */
int tetra_lmac_decode_batch(int type, const uint8_t* type5, int n_blocks, int in_stride, const uint32_t* scramb_init,
                            uint8_t* type2, int out_stride, int32_t* crc_ok, int device);

/*
 * The one piece of the SYNC-PDU read-out (tetra_lower_mac.c:246-275) that feeds back into the decoding chain: every SB1
 * block with a good CRC sets the cell's scrambling code, tcd->scramb_init = tetra_scramb_get_init(mcc, mnc, colour code)
 * with colour code = type2[4..9], mcc = type2[31..40], mnc = type2[41..54] (bits_to_uint, MSB first; :258-266), and every
 * later block of that receiver -- starting with the BBK and SB2 of the same burst, which tetra_burst_rx_cb hands over
 * after the SB1 -- is descrambled with it.  The reference keeps one process-global tcd (:116); here every channel has one.
 *
 * d_sb1_type2   [n_channels * frames_per_channel][type2_stride] decoded SB1 rows, frame slots in time order per channel
 *               (the layout tetra_bsync_process_device + tetra_burst_demux_device + tetra_lmac_decode_batch_device produce)
 * d_crc_ok, d_valid  per row: the decoder's crc_ok, the demultiplexer's valid (row is an SB1 at all)
 * d_chan_scramb [n_channels] uint32 in/out: the code in force when the call starts / after its last frame (0 for a fresh
 *               receiver, as the reference's zero-initialised tcd)
 * d_row_scramb  [n_channels * frames_per_channel] uint32 out: the code in force for the non-SB1 blocks of each frame slot
 *               -- feed it to tetra_lmac_decode_batch_device as d_scramb_init for SB2 / NDB / SCH-F / BBK rows
 */
int tetra_lmac_track_scramb_device(const uint8_t* d_sb1_type2, int type2_stride, const int32_t* d_crc_ok, const int32_t* d_valid,
                                   int n_channels, int frames_per_channel, uint32_t* d_chan_scramb, uint32_t* d_row_scramb,
                                   void* hip_stream);

/*
 * The whole SYNC-PDU read-out and the PHY's TDMA clock for every channel at once -- what labels a frame with its TDMA time.
 * Reference: tp_sap_udata_ind's SB1 case (src/decoder/src/lower_mac/tetra_lower_mac.c:246-275: colour code, TN/FN/MN, MCC,
 * MNC of a SYNC PDU with a good CRC go to tcd; tcd->time is then copied to t_phy_state.time WHATEVER the CRC was, :268-269),
 * the per-frame increment of the LOCKED receiver (src/decoder/src/phy/tetra_burst_sync.c:113: tetra_tdma_time_add_tn before
 * the callback) and its normalisation (src/decoder/src/tetra_tdma.c:28-78, reproduced to the letter including the wrap
 * thresholds tn > 4, fn > 18, mn > 60).  The reference keeps ONE process-global tcd / t_phy_state (tetra_lower_mac.c:116,
 * tetra_burst_sync.c:34); here every channel has its own, device-resident.
 */
typedef struct tetra_lmac_cell_state {
    uint32_t scramb_init;              /* tcd->scramb_init (0 for a fresh receiver) */
    uint32_t colour_code, mcc, mnc;    /* tcd->colour_code / mcc / mnc of the last SYNC PDU with a good CRC */
    uint32_t tcd_tn, tcd_fn, tcd_mn;   /* tcd->time */
    uint32_t phy_tn, phy_fn, phy_mn;   /* t_phy_state.time */
} tetra_lmac_cell_state_t;
/*
 * d_sb1_type2, d_crc_ok, d_valid   as tetra_lmac_track_scramb_device: decoded SB1 rows per frame slot, the decoder's crc_ok,
 *               the demultiplexer's valid (the slot's frame is a SYNC burst)
 * d_n_frames    [n_channels] int32: frame slots of each channel that hold a consumed frame in this call
 *               (tetra_bsync_process_device's d_n_frames; NULL: all frames_per_channel) -- only those advance the clock
 * d_cell        [n_channels] in/out (all zero for fresh receivers, like the reference's zero-initialised globals)
 * d_row_scramb  [n_channels * frames_per_channel] uint32 out: the code in force for the slot's non-SB1 blocks
 * d_row_time_rx [..] uint32 out, may be NULL: t_phy_state.time when tetra_burst_rx_cb is entered for the slot's frame
 *               (t_display_st->curr_multiframe / curr_frame, tetra_burst.c:349-350), packed tn | fn << 8 | mn << 16
 * d_row_time    [..] uint32 out, may be NULL: t_phy_state.time after the slot's SB1 block (if it has one), i.e. the time
 *               every later block of the burst is handled under; unused slots get 0 in both.
 */
int tetra_lmac_track_sync_device(const uint8_t* d_sb1_type2, int type2_stride, const int32_t* d_crc_ok, const int32_t* d_valid,
                                 const int32_t* d_n_frames, int n_channels, int frames_per_channel, tetra_lmac_cell_state_t* d_cell,
                                 uint32_t* d_row_scramb, uint32_t* d_row_time_rx, uint32_t* d_row_time, void* hip_stream);

/*
 * The same read-out for SB1 rows that are COMPACT (round 6): d_sb1_type2 / d_crc_ok hold one row per entry of the SYNC list of
 * tetra_burst_index_device (decoded by tetra_lmac_decode_frames_device), d_chan_first_sync[c] is the position in that list of
 * channel c's first entry, and a frame slot is a SYNC burst where d_frame_type says so.  One wavefront per channel, a frame slot
 * per lane, no walk over the slots: the last SYNC frame before a slot and the last one with a good CRC are found with two ballots,
 * their fields fetched by lane shuffle, and the clock k slots after it was set is one literal tetra_tdma_time_add_tn step plus
 * k - 1 in closed form (the slot-layout form above walks HBM: a dependent load per slot).  Outputs as above; with d_sb1_labels != NULL (needs d_frame_bitnum) also the label of every SB1 row, which the decode
 * launch cannot write because the times come from here.  Any frames_per_channel (64 slots at a time, the state carried in between).
 */
int tetra_lmac_track_sync_lists_device(const uint8_t* d_sb1_type2, int type2_stride, const int32_t* d_crc_ok, const int32_t* d_frame_type,
                                       const int32_t* d_n_frames, const int32_t* d_chan_first_sync, int n_channels, int frames_per_channel,
                                       tetra_lmac_cell_state_t* d_cell, uint32_t* d_row_scramb, uint32_t* d_row_time_rx, uint32_t* d_row_time,
                                       const uint32_t* d_frame_bitnum, tetra_lmac_label_t* d_sb1_labels, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
