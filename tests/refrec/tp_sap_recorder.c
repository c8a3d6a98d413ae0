/* Test-side recorder for the ONE downstream callback of the reference's burst path, tp_sap_udata_ind()
 * (declared phy/tetra_burst.h:18; the reference's own definition lives in lower_mac/tetra_lower_mac.c, which needs the
 * upper MAC, crypto and ETSI codec objects and is not built here).  With it the reference's unmodified
 * tetra_burst_sync_in() (phy/tetra_burst_sync.c:54-155) and tetra_burst_rx_cb() (phy/tetra_burst.c:343-393) from
 * oracle/_ref/libtetra_burst_ref.so can RUN, and the tests compare what they hand downstream with
 * oracle/burst_sync_oracle.c and with the device kernels.
 *
 * Our own code (nothing copied); built by oracle/build_ref.sh against the reference's headers into
 * oracle/_ref/libtetra_tpsap_recorder.so and loaded RTLD_GLOBAL so that the reference library's lazy reference to
 * tp_sap_udata_ind binds here.  It owns the structs the reference's functions want (tetra_rx_state, tetra_mac_state,
 * tetra_display_state) so that no struct layout is restated in Python.  Test infrastructure only. */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <tetra_common.h>
#include <phy/tetra_burst.h>
#include <phy/tetra_burst_sync.h>

typedef struct {
    int32_t type, blk_num, len;
    uint32_t frame_bitnum;            /* trs->bitbuf_start_bitnum when the burst was handed over */
    uint32_t tn, fn, mn;              /* t_phy_state.time at that moment */
    uint8_t bits[432];
} rec_event_t;

typedef struct {
    struct tetra_rx_state rx;
    struct tetra_mac_state mac;
    struct tetra_display_state disp;
    rec_event_t* ev;
    int n_ev, cap_ev;
    const struct tetra_phy_state* phy;   /* &t_phy_state of the reference library (set by the test) */
} rec_t;

void tp_sap_udata_ind(enum tp_sap_data_type type, int blk_num, const uint8_t* bits, unsigned int len, void* priv) {
    /* priv is trs->burst_cb_priv = &rec->mac; the recorder sits at a fixed offset around it */
    rec_t* r = (rec_t*)((char*)priv - offsetof(rec_t, mac));
    if (r->n_ev == r->cap_ev) {
        r->cap_ev = r->cap_ev ? 2 * r->cap_ev : 256;
        r->ev = (rec_event_t*)realloc(r->ev, sizeof(rec_event_t) * (size_t)r->cap_ev);
    }
    rec_event_t* e = &r->ev[r->n_ev++];
    memset(e, 0, sizeof(*e));
    e->type = (int32_t)type;
    e->blk_num = blk_num;
    e->len = (int32_t)len;
    e->frame_bitnum = r->rx.bitbuf_start_bitnum;
    if (r->phy) { e->tn = r->phy->time.tn; e->fn = r->phy->time.fn; e->mn = r->phy->time.mn; }
    memcpy(e->bits, bits, len <= sizeof(e->bits) ? len : sizeof(e->bits));
}

rec_t* rec_new(const void* phy_state) {
    rec_t* r = (rec_t*)calloc(1, sizeof(rec_t));
    r->mac.t_display_st = &r->disp;
    r->rx.burst_cb_priv = &r->mac;
    r->phy = (const struct tetra_phy_state*)phy_state;
    return r;
}

void rec_free(rec_t* r) {
    if (r) { free(r->ev); free(r); }
}

void rec_set_traffic(rec_t* r, int is_traffic) { r->mac.cur_burst.is_traffic = is_traffic; }

/* Feeds `bits` to the reference's tetra_burst_sync_in (passed in by the test) in calls of `chunk` bits. */
typedef int (*sync_in_fn)(struct tetra_rx_state*, uint8_t*, unsigned int);
void rec_feed(rec_t* r, sync_in_fn sync_in, const uint8_t* bits, int n_bits, int chunk) {
    uint8_t tmp[4096];
    if (chunk < 1) chunk = 1;
    if (chunk > (int)sizeof(tmp)) chunk = (int)sizeof(tmp);
    for (int i = 0; i < n_bits; i += chunk) {
        const int n = n_bits - i < chunk ? n_bits - i : chunk;
        memcpy(tmp, bits + i, (size_t)n);       /* the reference takes a non-const pointer */
        sync_in(&r->rx, tmp, (unsigned)n);
    }
}

/* rx state as four words (state, bits_in_buf, bitbuf_start_bitnum, next_frame_start_bitnum) + the buffered bits */
void rec_rx_state(const rec_t* r, uint32_t out[4], uint8_t* bitbuf_out) {
    out[0] = (uint32_t)r->rx.state;
    out[1] = r->rx.bits_in_buf;
    out[2] = r->rx.bitbuf_start_bitnum;
    out[3] = r->rx.next_frame_start_bitnum;
    if (bitbuf_out) memcpy(bitbuf_out, r->rx.bitbuf, r->rx.bits_in_buf);
}

int rec_event_count(const rec_t* r) { return r->n_ev; }
int rec_event_size(void) { return (int)sizeof(rec_event_t); }
void rec_events(const rec_t* r, int first, int count, rec_event_t* out) { memcpy(out, r->ev + first, sizeof(rec_event_t) * (size_t)count); }
void rec_clear_events(rec_t* r) { r->n_ev = 0; }
int rec_timeslot_content(const rec_t* r, int tn0) { return r->disp.timeslot_content[tn0 & 3]; }
