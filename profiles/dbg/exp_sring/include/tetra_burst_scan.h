/*
 * tetra_burst_scan.h -- C ABI of the batched training-sequence search.
 *
 * Replaces, for C channels at once, the reference's
 *     int tetra_find_train_seq(const uint8_t *in, unsigned int end_of_in, uint32_t mask_of_train_seq, unsigned int *offset)
 * (src/decoder/src/phy/tetra_burst.c:271-341, called by the burst synchroniser tetra_burst_sync_in(),
 * src/decoder/src/phy/tetra_burst_sync.c:54-155, and mirrored by the plugin's NETSYMS indicator, src/main.cpp:385-414):
 * scan positions cur = 0 .. end_of_in-1 of a one-bit-per-byte stream and return the FIRST position holding one of the five
 * ETSI EN 300 392-2 9.4.4.3 training sequences enabled in the mask, checked in the reference's order (sync, normal 1, 2, 3,
 * extended), a sequence counting only if it lies entirely inside [0, end_of_in).  Bit-exact with the reference including
 * its look-ahead quirk: the 22-bit pre-filter is seeded from in[0..19] and then fed in[cur+21], so in[20] is never
 * shifted in and the first 21 positions see a misaligned filter (a sequence starting there can be missed) -- reproduced.
 * Like the reference, the scan reads up to in[end_of_in + 20]; every row must have those bytes readable
 * (end_of_in[c] + 21 <= bits_stride; a larger end_of_in is cut back to bits_stride - 21 inside the kernel, so a bad count
 * can never read into the next channel's row).
 * Parity for this entry point is PINNED: tests compare it with the reference function itself, built from the reference's
 * own source file into oracle/_ref (oracle/build_ref.sh).
 */
#ifndef TETRA_BURST_SCAN_H
#define TETRA_BURST_SCAN_H

#include <stdint.h>

#include "tetra_demod.h"

#ifdef __cplusplus
extern "C" {
#endif

/* enum tetra_train_seq of src/decoder/src/phy/tetra_burst.h:26-32 */
enum {
    TETRA_TRAIN_NORM_1 = 0,
    TETRA_TRAIN_NORM_2 = 1,
    TETRA_TRAIN_NORM_3 = 2,
    TETRA_TRAIN_SYNC = 3,
    TETRA_TRAIN_EXT = 4
};

/*
 * d_bits      [n_channels][bits_stride] uint8, one bit per byte (the demodulator's output layout), device pointer
 * d_end_of_in [n_channels] int32: the reference's end_of_in per channel (e.g. the demodulator's n_bits)
 * mask        bit (1 << TETRA_TRAIN_x) enables sequence x (the reference's mask_of_train_seq)
 * d_type      [n_channels] int32 out: the reference's return value (TETRA_TRAIN_x, or -1 if nothing found)
 * d_offset    [n_channels] int32 out: the reference's *offset (-1 if nothing found)
 * Enqueued on hip_stream of the current device, no synchronisation.  Returns TETRA_OK or TETRA_ERR_*.
 */
int tetra_find_train_seq_batch_device(const uint8_t* d_bits, int n_channels, int bits_stride, const int32_t* d_end_of_in,
                                      uint32_t mask, int32_t* d_type, int32_t* d_offset, void* hip_stream);
/* Host-pointer variant (copies in/out, synchronises; device = HIP ordinal or -1 for the current one). */
int tetra_find_train_seq_batch(const uint8_t* bits, int n_channels, int bits_stride, const int32_t* end_of_in, uint32_t mask,
                               int32_t* type, int32_t* offset, int device);

/*
 * The plugin's own training-sequence indicator (src/main.cpp:385-414 with the sequences at :457-468 and the state at
 * :470-472; the GUI draws it as a box indicator in the NETSYMS mode, main.cpp:331), for C channels at once.  Per received bit the reference
 * shifts the bit into a 45-entry window and compares the window's HEAD with eight sequences (normal n/p/q 22 bits, N/P 33,
 * extended x 30 / X 45, synchronisation y 38); a hit sets `tsfound` and arms `symsbeforeexpire = 2048`, which every bit
 * then counts down (the arming bit included), clearing `tsfound` when it reaches zero.  Only the value after the last bit
 * of a call is observable, so the device finds the LAST hit of the call and derives both from it; the window (its newest
 * 44 bits) and the counter are carried per channel.  State starts as the plugin's members would if zero-initialised
 * (window all zero, counter 0, tsfound false).  Checker: oracle/burst_sync_oracle.c restates the handler literally
 * (parity unpinned: main.cpp needs SDR++ and cannot be built here).
 */
typedef struct tetra_ts_indicator tetra_ts_indicator_t;

/* device = HIP ordinal or -1 for the current one */
int tetra_ts_indicator_create(int n_channels, int device, tetra_ts_indicator_t** out);
void tetra_ts_indicator_destroy(tetra_ts_indicator_t* h);
/* channel = -1: every channel back to the initial state */
int tetra_ts_indicator_reset(tetra_ts_indicator_t* h, int channel);
/*
 * d_bits    [n_channels][bits_stride] uint8, one bit per byte (values 0 / 1), device pointer; bits_stride % 4 == 0
 * d_n_bits  [n_channels] int32: bits of this call per channel (the demodulator's n_bits; cut back to bits_stride)
 * d_found   [n_channels] uint8 out: the reference's tsfound after the call's last bit
 * d_expire  [n_channels] int32 out or NULL: the reference's symsbeforeexpire after the call's last bit
 * Enqueued on hip_stream of the handle's device, no synchronisation.
 */
int tetra_ts_indicator_process_device(tetra_ts_indicator_t* h, const uint8_t* d_bits, int bits_stride, const int32_t* d_n_bits,
                                      uint8_t* d_found, int32_t* d_expire, void* hip_stream);
/* Host-pointer variant (copies in/out, synchronises). */
int tetra_ts_indicator_process(tetra_ts_indicator_t* h, const uint8_t* bits, int bits_stride, const int32_t* n_bits,
                               uint8_t* found, int32_t* expire);

#ifdef __cplusplus
}
#endif
#endif
