/*
 * burst_sync_oracle.c -- CPU restatement of the reference's burst synchroniser and burst demultiplexer.
 *
 * TEST INFRASTRUCTURE ONLY (see tetra_oracle.h): the product never includes, links or calls this file.
 *
 * Restates, line by line, src/decoder/src/phy/:
 *   tetra_find_train_seq()   tetra_burst.c:271-341   (the search, including its 22-bit look-ahead pre-filter that is
 *                                                     seeded from in[0..19] and then fed in[cur+21], :289-296)
 *   make_bitbuf_space()      tetra_burst_sync.c:38-51
 *   tetra_burst_sync_in()    tetra_burst_sync.c:54-155 (UNLOCKED -> KNOW_FSTART -> LOCKED; KNOW_FSTART falls through)
 *   tetra_burst_rx_cb()      tetra_burst.c:343-393    (which blocks of a burst go to the lower MAC, and from where)
 * Pinning: tetra_find_train_seq is checked against the reference's own function built into oracle/_ref
 * (tests/test_burst_sync.py); the block offsets are checked by round trip through the reference's own burst builders
 * (build_sync_c_d_burst / build_norm_c_d_burst, tetra_burst.c:171-269).  The state machine itself calls into the lower MAC
 * (tp_sap_udata_ind, in a file that needs the ETSI codec sources the repository does not carry), so it cannot be run
 * from oracle/_ref: for it this restatement is the checker, PARITY UNPINNED beyond the two anchors above.
 *
 * The reference hands each tetra_burst_rx_cb() burst on; here every frame the LOCKED state consumes is reported through
 * a callback: type >= 0 where the reference calls tetra_burst_rx_cb(burst, 510, type), -1 where it does not.
 */
#include <stdint.h>
#include <string.h>

#define BITS_PER_TS 510           /* TETRA_BITS_PER_TS, tetra_common.h:237-238 */
#define BITBUF_SIZE 4096          /* sizeof(trs->bitbuf), tetra_burst_sync.h:15 */

enum { RX_S_UNLOCKED = 0, RX_S_KNOW_FSTART = 1, RX_S_LOCKED = 2 };                  /* tetra_burst_sync.h:6-10 */
enum { TRAIN_NORM_1 = 0, TRAIN_NORM_2 = 1, TRAIN_NORM_3 = 2, TRAIN_SYNC = 3, TRAIN_EXT = 4 }; /* tetra_burst.h:26-32 */

/* EN 300 392-2 9.4.4.3.2-4 (tetra_burst.c:61-72) */
static const uint8_t n_bits[22] = { 1,1, 0,1, 0,0, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,1, 0,0 };
static const uint8_t p_bits[22] = { 0,1, 1,1, 1,0, 1,0, 0,1, 0,0, 0,0, 1,1, 0,1, 1,1, 1,0 };
static const uint8_t q_bits[22] = { 1,0, 1,1, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 1,0, 1,1, 0,1 };
static const uint8_t x_bits[30] = { 1,0, 0,1, 1,1, 0,1, 0,0, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,1, 0,0, 0,0, 1,1 };
static const uint8_t y_bits[38] = { 1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1 };

typedef struct {
    int32_t state;
    uint32_t bits_in_buf;
    uint32_t bitbuf_start_bitnum;
    uint32_t next_frame_start_bitnum;
    uint8_t bitbuf[BITBUF_SIZE + 64];      /* + slack: the search reads up to 21 bytes past end_of_in */
} bs_oracle_state_t;

typedef void (*bs_frame_cb)(void* user, const uint8_t* burst, int type, uint32_t start_bitnum);

/* tetra_burst.c:271-341 */
int bs_oracle_find_train_seq(const uint8_t* in, unsigned end_of_in, uint32_t mask, unsigned* offset) {
    uint32_t head[5] = { 0, 0, 0, 0, 0 };
    for (int i = 0; i < 22; i++) {
        head[0] = (head[0] << 1) | y_bits[i];
        head[1] = (head[1] << 1) | n_bits[i];
        head[2] = (head[2] << 1) | p_bits[i];
        head[3] = (head[3] << 1) | q_bits[i];
        head[4] = (head[4] << 1) | x_bits[i];
    }
    uint32_t filter = 0;
    for (int i = 0; i < 20; i++) filter = (filter << 1) | in[i];
    for (unsigned cur = 0; cur < end_of_in; cur++) {
        filter = ((filter << 1) | in[cur + 21]) & 0x3fffffu;
        int match = 0;
        for (int i = 0; i < 5; i++) match |= (filter == head[i]);
        if (!match) continue;
        const unsigned remain = end_of_in - cur;
        const struct { int type; const uint8_t* seq; unsigned len; } order[5] = {
            { TRAIN_SYNC, y_bits, 38 }, { TRAIN_NORM_1, n_bits, 22 }, { TRAIN_NORM_2, p_bits, 22 },
            { TRAIN_NORM_3, q_bits, 22 }, { TRAIN_EXT, x_bits, 30 } };
        for (int s = 0; s < 5; s++) {
            if ((mask & (1u << order[s].type)) && remain >= order[s].len && !memcmp(in + cur, order[s].seq, order[s].len)) {
                *offset = cur;
                return order[s].type;
            }
        }
    }
    return -1;
}

void bs_oracle_reset(bs_oracle_state_t* trs) { memset(trs, 0, sizeof(*trs)); }

/* tetra_burst_sync.c:38-51 */
static void make_bitbuf_space(bs_oracle_state_t* trs, unsigned len) {
    unsigned space = BITBUF_SIZE - trs->bits_in_buf;
    if (space < len) {
        const unsigned delta = len - space;
        memmove(trs->bitbuf, trs->bitbuf + delta, trs->bits_in_buf - delta);
        trs->bits_in_buf -= delta;
        trs->bitbuf_start_bitnum += delta;
    }
}

/* tetra_burst_sync.c:54-155; returns what the reference returns */
int bs_oracle_sync_in(bs_oracle_state_t* trs, const uint8_t* bits, unsigned len, bs_frame_cb cb, void* user) {
    int rc;
    unsigned offs = 0;
    make_bitbuf_space(trs, len);
    memcpy(trs->bitbuf + trs->bits_in_buf, bits, len);
    trs->bits_in_buf += len;

    switch (trs->state) {
    case RX_S_UNLOCKED:
        if (trs->bits_in_buf < BITS_PER_TS * 2) return (int)len;
        rc = bs_oracle_find_train_seq(trs->bitbuf, trs->bits_in_buf, 1u << TRAIN_SYNC, &offs);
        if (rc < 0) return rc;
        trs->state = RX_S_KNOW_FSTART;
        trs->next_frame_start_bitnum = trs->bitbuf_start_bitnum + offs + 296;
        break;
    case RX_S_KNOW_FSTART:
        if (trs->bitbuf_start_bitnum + trs->bits_in_buf < trs->next_frame_start_bitnum) return 0;
        {
            const int offset = (int)(trs->next_frame_start_bitnum - trs->bitbuf_start_bitnum);
            const int remaining = (int)trs->bits_in_buf - offset;
            memmove(trs->bitbuf, trs->bitbuf + offset, (size_t)remaining);
            trs->bits_in_buf = (uint32_t)remaining;
            trs->bitbuf_start_bitnum += (uint32_t)offset;
            trs->next_frame_start_bitnum += BITS_PER_TS;
            trs->state = RX_S_LOCKED;
        }
        /* fall through, as the reference does (:103-104) */
        __attribute__((fallthrough));
    case RX_S_LOCKED:
        if (trs->bits_in_buf < BITS_PER_TS) return (int)len;
        rc = bs_oracle_find_train_seq(trs->bitbuf, trs->bits_in_buf,
                                      (1u << TRAIN_NORM_1) | (1u << TRAIN_NORM_2) | (1u << TRAIN_SYNC), &offs);
        {
            int reported = -1;
            switch (rc) {
            case TRAIN_SYNC:
                if (offs == 214) reported = rc;
                else trs->state = RX_S_UNLOCKED;
                break;
            case TRAIN_NORM_1:
            case TRAIN_NORM_2:
            case TRAIN_NORM_3:
                if (offs == 244) reported = rc;
                break;
            default:
                trs->state = RX_S_UNLOCKED;
                break;
            }
            if (cb) cb(user, trs->bitbuf, reported, trs->bitbuf_start_bitnum);
        }
        trs->bits_in_buf -= BITS_PER_TS;
        memmove(trs->bitbuf, trs->bitbuf + BITS_PER_TS, trs->bits_in_buf);
        trs->bitbuf_start_bitnum += BITS_PER_TS;
        trs->next_frame_start_bitnum += BITS_PER_TS;
        break;
    }
    return (int)len;
}

/* Recorder + driver used by the tests: feeds `bits` in calls of `chunk` bits (chunk 1 = the call pattern the device
 * entry point is defined against, include/tetra_burst_sync.h) and stores every consumed frame. */
typedef struct {
    uint8_t* frames;        /* [max_frames][512] */
    int32_t* types;         /* [max_frames] */
    uint32_t* bitnums;      /* [max_frames] */
    int max_frames, n;
} bs_recorder_t;

static void record_cb(void* user, const uint8_t* burst, int type, uint32_t start_bitnum) {
    bs_recorder_t* r = (bs_recorder_t*)user;
    if (r->n < r->max_frames) {
        memcpy(r->frames + (size_t)r->n * 512, burst, BITS_PER_TS);
        r->types[r->n] = type;
        r->bitnums[r->n] = start_bitnum;
    }
    r->n++;
}

int bs_oracle_feed(bs_oracle_state_t* trs, const uint8_t* bits, int n_bits, int chunk, uint8_t* frames, int32_t* types,
                   uint32_t* bitnums, int max_frames) {
    bs_recorder_t r = { frames, types, bitnums, max_frames, 0 };
    if (chunk < 1) chunk = 1;
    for (int i = 0; i < n_bits; i += chunk)
        bs_oracle_sync_in(trs, bits + i, (unsigned)(n_bits - i < chunk ? n_bits - i : chunk), record_cb, &r);
    return r.n;
}

/* tetra_burst_rx_cb(), tetra_burst.c:343-393, as a table: which bits of a 510-bit burst of training-sequence type
 * `train` form block kind `tpsap` (enum tp_sap_data_type, tetra_burst.h:9-16), block number blk_num (1 or 2; 0 where the
 * reference passes 0).  Writes the block's type-5 bits to out, returns their count, or 0 if that burst type carries no
 * such block.  Offsets: tetra_burst.c:33-49. */
int bs_oracle_demux(const uint8_t* burst, int train, int tpsap, int blk_num, uint8_t* out) {
    enum { SB1 = 0, SB2 = 1, NDB = 2, BBK = 3, SCH_HU = 4, SCH_F = 5 };
    const int SB_BLK1_OFFSET = (6 + 1 + 40) * 2, SB_BBK_OFFSET = (6 + 1 + 40 + 60 + 19) * 2, SB_BLK2_OFFSET = (6 + 1 + 40 + 60 + 19 + 15) * 2;
    const int NDB_BLK1_OFFSET = (5 + 1 + 1) * 2, NDB_BBK1_OFFSET = (5 + 1 + 1 + 108) * 2, NDB_BBK2_OFFSET = (5 + 1 + 1 + 108 + 7 + 11) * 2,
              NDB_BLK2_OFFSET = (5 + 1 + 1 + 108 + 7 + 11 + 8) * 2;
    if (train == TRAIN_SYNC) {
        if (tpsap == SB1 && blk_num == 1) { memcpy(out, burst + SB_BLK1_OFFSET, 120); return 120; }
        if (tpsap == BBK) { memcpy(out, burst + SB_BBK_OFFSET, 30); return 30; }
        if (tpsap == SB2 && blk_num == 2) { memcpy(out, burst + SB_BLK2_OFFSET, 216); return 216; }
        return 0;
    }
    if (train == TRAIN_NORM_1 || train == TRAIN_NORM_2) {
        if (tpsap == BBK) { memcpy(out, burst + NDB_BBK1_OFFSET, 14); memcpy(out + 14, burst + NDB_BBK2_OFFSET, 16); return 30; }
        if (train == TRAIN_NORM_2 && tpsap == NDB && blk_num == 1) { memcpy(out, burst + NDB_BLK1_OFFSET, 216); return 216; }
        if (train == TRAIN_NORM_2 && tpsap == NDB && blk_num == 2) { memcpy(out, burst + NDB_BLK2_OFFSET, 216); return 216; }
        if (train == TRAIN_NORM_1 && tpsap == SCH_F) {
            memcpy(out, burst + NDB_BLK1_OFFSET, 216);
            memcpy(out + 216, burst + NDB_BLK2_OFFSET, 216);
            return 432;
        }
    }
    return 0;
}

int bs_oracle_state_size(void) { return (int)sizeof(bs_oracle_state_t); }

/* ---------------------------------------------------------------------------------------------------------------------
 * The plugin's training-sequence indicator, src/main.cpp:385-414 (_demodSinkHandler without the UDP send), restated
 * literally: sequences :457-468, state :470-472 (tsfind_buffer[45], tsfound, symsbeforeexpire), taken as zero-initialised.
 * PARITY UNPINNED: main.cpp needs SDR++ and cannot be built here; the handler is 25 lines of byte compares.
 * ------------------------------------------------------------------------------------------------------------------- */
static const uint8_t training_seq_n[22] = { 1,1, 0,1, 0,0, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,1, 0,0 };
static const uint8_t training_seq_p[22] = { 0,1, 1,1, 1,0, 1,0, 0,1, 0,0, 0,0, 1,1, 0,1, 1,1, 1,0 };
static const uint8_t training_seq_q[22] = { 1,0, 1,1, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 1,0, 1,1, 0,1 };
static const uint8_t training_seq_N[33] = { 1,1,1, 0,0,1, 1,0,1, 1,1,1, 0,0,0, 1,1,1, 1,0,0, 0,1,1, 1,1,0, 0,0,0, 0,0,0 };
static const uint8_t training_seq_P[33] = { 1,0,1, 0,1,1, 1,1,1, 1,0,1, 0,1,0, 1,0,1, 1,1,0, 0,0,1, 1,0,0, 0,1,0, 0,1,0 };
static const uint8_t training_seq_x[30] = { 1,0, 0,1, 1,1, 0,1, 0,0, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,1, 0,0, 0,0, 1,1 };
static const uint8_t training_seq_X[45] = { 0,1,1,1,0,0,1,1,0,1,0,0,0,0,1,0,0,0,1,1,1,0,1,1,0,1,0,1,0,1,1,1,1,1,0,1,0,0,0,0,0,1,1,1,0 };
static const uint8_t training_seq_y[38] = { 1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1 };

typedef struct {
    uint8_t tsfind_buffer[45];
    int32_t tsfound;
    int32_t symsbeforeexpire;
} ts_indicator_state;

void ts_indicator_init(ts_indicator_state* s) { memset(s, 0, sizeof(*s)); }

void ts_indicator_feed(ts_indicator_state* s, const uint8_t* data, int count) {
    for (int j = 0; j < count; j++) {
        for (int i = 0; i < 44; i++) s->tsfind_buffer[i] = s->tsfind_buffer[i + 1];
        s->tsfind_buffer[44] = data[j];
        if (!memcmp(s->tsfind_buffer, training_seq_n, sizeof(training_seq_n)) ||
            !memcmp(s->tsfind_buffer, training_seq_p, sizeof(training_seq_p)) ||
            !memcmp(s->tsfind_buffer, training_seq_q, sizeof(training_seq_q)) ||
            !memcmp(s->tsfind_buffer, training_seq_N, sizeof(training_seq_N)) ||
            !memcmp(s->tsfind_buffer, training_seq_P, sizeof(training_seq_P)) ||
            !memcmp(s->tsfind_buffer, training_seq_x, sizeof(training_seq_x)) ||
            !memcmp(s->tsfind_buffer, training_seq_X, sizeof(training_seq_X)) ||
            !memcmp(s->tsfind_buffer, training_seq_y, sizeof(training_seq_y))) {
            s->tsfound = 1;
            s->symsbeforeexpire = 2048;
        }
        if (s->symsbeforeexpire > 0) {
            s->symsbeforeexpire--;
            if (s->symsbeforeexpire == 0) s->tsfound = 0;
        }
    }
}
