/*
 * tetra_burst_sync.h -- C ABI of the batched burst synchroniser and burst demultiplexer (second
 * half: "lets the device emit aligned 510-bit slots + lock state instead of raw bits").
 *
 * Replaces, for C channels at once, the reference's per-receiver
 *     int tetra_burst_sync_in(struct tetra_rx_state *trs, uint8_t *bits, unsigned int len)
 * (src/decoder/src/phy/tetra_burst_sync.c:54-155; state struct tetra_rx_state, tetra_burst_sync.h:12-20; fed by the
 * plugin's decoder block, src/dsp/osmotetra_dec.h:182-184) and the block split of
 *     void tetra_burst_rx_cb(const uint8_t *burst, unsigned int len, enum tetra_train_seq type, void *priv)
 * (src/decoder/src/phy/tetra_burst.c:343-393).  The reference keeps this state in one process-global receiver
 * (t_phy_state, tetra_burst_sync.c:34); here every channel has its own.
 *
 * Semantics.  The reference consumes at most one 510-bit frame per call and its 4096-byte bit buffer drops the oldest bits
 * when a call overfills it (make_bitbuf_space, :38-51), so what it outputs depends on how its caller chunks the stream.
 * This entry point is defined as -- and tested bit-exact against -- the reference's behaviour when it is handed the same
 * bits ONE BIT PER CALL, the chunking-independent limit of the small stream buffers the plugin feeds it: for every
 * channel, after a call the state (state, bits_in_buf, bitbuf_start_bitnum, next_frame_start_bitnum and the buffered
 * bits) equals the reference's, and one frame record is emitted per frame its LOCKED state consumes, in order:
 *     frame_type >= 0   the reference calls tetra_burst_rx_cb(burst, 510, frame_type) (TETRA_TRAIN_SYNC at offset 214,
 *                       TETRA_TRAIN_NORM_1 / _2 at offset 244)
 *     frame_type == -1  the reference consumes the frame without a callback (training sequence missing or misplaced;
 *                       it also falls back to UNLOCKED unless a normal sequence was merely misplaced)
 * including the reference search's misaligned look-ahead filter in the first 21 buffer positions (tetra_burst_scan.h).
 * Input bits must be 0/1 bytes (the demodulator's output).
 * Parity: the training-sequence search underneath is pinned against the reference's own tetra_find_train_seq
 * (oracle/_ref); the burst layouts are pinned by round trip through the reference's own burst builders; the state
 * machine is checked against a literal restatement (oracle/burst_sync_oracle.c) -- tetra_burst_sync_in itself cannot be
 * run from oracle/_ref because its callback chain ends in tetra_lower_mac.c, which needs the ETSI codec sources the
 * reference repository does not carry.
 */
#ifndef TETRA_BURST_SYNC_H
#define TETRA_BURST_SYNC_H

#include <stdint.h>

#include "tetra_burst_scan.h"
#include "tetra_lmac.h"

#ifdef __cplusplus
extern "C" {
#endif

/* enum rx_state, src/decoder/src/phy/tetra_burst_sync.h:6-10 */
enum { TETRA_RX_S_UNLOCKED = 0, TETRA_RX_S_KNOW_FSTART = 1, TETRA_RX_S_LOCKED = 2 };

#define TETRA_BITS_PER_TS 510      /* tetra_common.h:237-238 */
#define TETRA_FRAME_STRIDE 512     /* bytes per emitted frame row (510 bits + 2 zero bytes) */
#define TETRA_FRAME_WORDS 16       /* 32-bit words per PACKED frame row (first bit = most significant; word 15 carries bits 480 .. 509) */
#define TETRA_FRAME_NONE (-2)      /* frame_type of an unused output slot */

/* struct tetra_rx_state (tetra_burst_sync.h:12-20) without bitbuf / burst_cb_priv */
typedef struct tetra_bsync_state {
    int32_t state;
    uint32_t bits_in_buf;
    uint32_t bitbuf_start_bitnum;
    uint32_t next_frame_start_bitnum;
} tetra_bsync_state_t;

typedef struct tetra_bsync tetra_bsync_t;

/* n_channels receivers, all UNLOCKED and empty; max_bits = largest n_bits of one process call (1 .. 262144). */
int tetra_bsync_create(int n_channels, int max_bits, int device, tetra_bsync_t** out);
int tetra_bsync_destroy(tetra_bsync_t* h);
int tetra_bsync_reset(tetra_bsync_t* h);
/* Frame slots per channel a process call needs: (4096 + max_bits) / 510 + 2. */
int tetra_bsync_max_frames(tetra_bsync_t* h);
/*
 * d_bits        [C][bits_stride] uint8, one bit per byte (device pointer, 4-byte aligned, bits_stride % 4 == 0,
 *               bits_stride >= max_bits or TETRA_ERR_SIZE)
 * d_n_bits      [C] int32: new bits per channel (e.g. the demodulator's n_bits); values above max_bits (or the row) are clamped
 * d_frames      [C][max_frames][512] uint8 out: the consumed frames, one bit per byte
 * d_frame_type  [C][max_frames] int32 out: see above; unused slots = TETRA_FRAME_NONE
 * d_frame_bitnum[C][max_frames] uint32 out: bitbuf_start_bitnum of the frame (absolute bit number of its first bit)
 * d_n_frames    [C] int32 out
 * Enqueued on hip_stream, no synchronisation.
 */
int tetra_bsync_process_device(tetra_bsync_t* h, const uint8_t* d_bits, int bits_stride, const int32_t* d_n_bits,
                               uint8_t* d_frames, int32_t* d_frame_type, uint32_t* d_frame_bitnum, int32_t* d_n_frames,
                               void* hip_stream);
/*
 * The same call with the frames PACKED, for consumers on the device (round 5): d_frames_packed [C][max_frames][TETRA_FRAME_WORDS]
 * uint32, 32 bits per word, first bit of the frame = most significant bit of word 0, the two spare bits of word 15 zero.  The
 * synchroniser has the stream packed in LDS anyway; handing the frames on like that writes 64 bytes per frame instead of 512 and
 * lets the demultiplexer behind (tetra_burst_demux_packed_device) read an eighth of the bytes.  Everything else -- state, frame
 * types, bit numbers, counts -- is identical to tetra_bsync_process_device (tests compare the two bit for bit); the byte-per-bit
 * form stays what the host decoder's tetra_burst_rx_cb() takes.
 */
int tetra_bsync_process_packed_device(tetra_bsync_t* h, const uint8_t* d_bits, int bits_stride, const int32_t* d_n_bits,
                                      uint32_t* d_frames_packed, int32_t* d_frame_type, uint32_t* d_frame_bitnum, int32_t* d_n_frames,
                                      void* hip_stream);
/* Host-pointer variant (copies in/out, synchronises). */
int tetra_bsync_process(tetra_bsync_t* h, const uint8_t* bits, int bits_stride, const int32_t* n_bits, uint8_t* frames,
                        int32_t* frame_type, uint32_t* frame_bitnum, int32_t* n_frames);
/* States of channels [first, first + count) (synchronises). */
int tetra_bsync_get_state(tetra_bsync_t* h, int first, int count, tetra_bsync_state_t* out);

/*
 * tetra_burst_rx_cb's block split for n frames at once: extracts block kind `tpsap` (TETRA_TPSAP_T_x), block number
 * blk_num (1 / 2 as the reference passes BLK_1 / BLK_2; ignored for BBK and SCH/F) from every frame whose type carries it:
 *     TETRA_TRAIN_SYNC   -> SB1 (blk 1, 120 bits at 94), BBK (30 bits at 252), SB2 (blk 2, 216 bits at 282)
 *     TETRA_TRAIN_NORM_2 -> BBK (14 bits at 230 + 16 bits at 266), NDB blk 1 (216 at 14), NDB blk 2 (216 at 282)
 *     TETRA_TRAIN_NORM_1 -> BBK (same), SCH/F (216 at 14 + 216 at 282)
 * (offsets tetra_burst.c:33-49).  d_rows [n][row_stride] receives the type-5 bits in the layout
 * tetra_lmac_decode_batch_device reads; d_valid[n] = 1 where the frame carries the block, else 0 and the row is zeroed.
 * d_frames [n][512], d_frame_type [n] as written by tetra_bsync_process_device (n = C * max_frames covers everything).
 */
int tetra_burst_demux_device(const uint8_t* d_frames, const int32_t* d_frame_type, int n, int tpsap, int blk_num,
                             uint8_t* d_rows, int row_stride, int32_t* d_valid, void* hip_stream);

/*
 * Compacting form: only the frames that carry the block kind produce a row.  d_rows [n][row_stride] is filled densely from
 * row 0 in frame order, d_row_frame[j] = index (into d_frames) of the frame row j came from, *d_n_rows = number of rows
 * (device memory; at most n).  With a real downlink three quarters of the frame slots do not carry a given kind, so the
 * decoder behind (tetra_lmac_decode_counted_device reads the count on the device) has a quarter of the rows to do.
 * Rows at and beyond *d_n_rows are unspecified (the head of d_rows serves as the call's scratch before the rows are written: no
 * allocation, nothing but kernel launches on hip_stream).  Nothing is read back to the host; enqueued on hip_stream.
 */
int tetra_burst_demux_compact_device(const uint8_t* d_frames, const int32_t* d_frame_type, int n, int tpsap, int blk_num,
                                     uint8_t* d_rows, int row_stride, int32_t* d_row_frame, int32_t* d_n_rows, void* hip_stream);

/* The two demultiplexers reading PACKED frames (d_frames_packed [n][TETRA_FRAME_WORDS] as tetra_bsync_process_packed_device writes
 * them); rows, validity, row order and counts are those of the byte-per-bit forms above. */
int tetra_burst_demux_packed_device(const uint32_t* d_frames_packed, const int32_t* d_frame_type, int n, int tpsap, int blk_num,
                                    uint8_t* d_rows, int row_stride, int32_t* d_valid, void* hip_stream);
int tetra_burst_demux_compact_packed_device(const uint32_t* d_frames_packed, const int32_t* d_frame_type, int n, int tpsap, int blk_num,
                                            uint8_t* d_rows, int row_stride, int32_t* d_row_frame, int32_t* d_n_rows, void* hip_stream);

/*
 * The frame lists of a call in one pass (round 6): which frame slots hold a SYNC / NORM_1 / NORM_2 burst, and which any of the
 * three, each in frame order -- the d_row_frame of every block kind tetra_burst_rx_cb hands on (SYNC: SB1, SB2; NORM_1: SCH/F;
 * NORM_2: NDB blk 1 + 2; any: BBK), for tetra_lmac_decode_frames_device.  Replaces the count / scan / index passes the compacting
 * demultiplexer runs per kind (18 launches for the six kinds of a downlink) by three launches.
 *   d_lists       [TETRA_N_LISTS][n] int32 out: list k holds indices into d_frame_type, ascending; entries past its count unspecified
 *   d_counts      [TETRA_N_LISTS] int32 out
 *   d_chan_first  [TETRA_N_LISTS][n / frames_per_channel] int32 out, may be NULL: position in list k of channel c's first entry
 *                 (= entries of list k that belong to channels < c); frames_per_channel must divide n
 *   d_work        [TETRA_N_LISTS * ((n + 255) / 256)] int32 scratch
 * Enqueued on hip_stream; nothing is read back.
 */
enum { TETRA_LIST_SYNC = 0, TETRA_LIST_NORM_1 = 1, TETRA_LIST_NORM_2 = 2, TETRA_LIST_ANY = 3, TETRA_N_LISTS = 4 };
int tetra_burst_index_device(const int32_t* d_frame_type, int n, int frames_per_channel, int32_t* d_lists, int32_t* d_counts,
                             int32_t* d_chan_first, int32_t* d_work, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
