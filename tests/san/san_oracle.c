/* Sanitizer run of the CPU checkers (oracle/tetra_oracle.c, burst_sync_oracle.c, chan_oracle.c): this program is built by
 * tests/test_sanitizers.py together with those sources under -fsanitize=address,undefined -fno-sanitize-recover=all and
 * drives every entry point the tests use, with the call shapes that stress buffer ends (1-sample calls, calls shorter than
 * the filters, maximum-size rows, reset/setter sequences, adversarial bit streams).  Exit 0 = clean. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/tetra_oracle.h"

/* burst_sync_oracle.c / chan_oracle.c have no header: their test-facing entry points */
typedef struct bs_oracle_state bs_oracle_state_t;
int bs_oracle_state_size(void);
void bs_oracle_reset(bs_oracle_state_t* trs);
int bs_oracle_feed(bs_oracle_state_t* trs, const uint8_t* bits, int n_bits, int chunk, uint8_t* frames, int32_t* types,
                   uint32_t* bitnums, int max_frames);
int bs_oracle_demux(const uint8_t* burst, int train, int tpsap, int blk_num, uint8_t* out);
int bs_oracle_find_train_seq(const uint8_t* in, unsigned end_of_in, uint32_t mask, unsigned* offset);
void chan_oracle_prototype(int M, int P, double cutoff_rel, float* h);
int chan_oracle_process(int M, int P, int D, const float* h, float* hist, int* phase, int64_t* frame_index, int n_in,
                        const float* x, float* out);

static uint32_t rng_state = 12345u;
static uint32_t rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static float frnd(void) { return (float)(rnd() & 0xffff) / 32768.0f - 1.0f; }

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "san_oracle: check failed line %d: %s\n", __LINE__, #c); return 1; } } while (0)

static int run_demod(void) {
    tetra_oracle_cfg_t cfg;
    tetra_oracle_default_cfg(&cfg);
    tetra_oracle_tables_t* tab = malloc(sizeof(*tab));
    CHECK(tetra_oracle_design(&cfg, tab) == 0);
    const int N = 20000;
    float* iq = malloc(sizeof(float) * 2 * N);
    for (int i = 0; i < N; i++) {   /* pi/4-DQPSK-ish: phase steps every 2 samples + noise; exact shape is irrelevant here */
        static double ph = 0.0;
        if ((i & 1) == 0) ph += (0.25 + 0.5 * (rnd() & 3)) * 3.14159265358979;
        iq[2 * i] = 0.3f * (float)cos(ph + 0.02 * i) + 0.02f * frnd();
        iq[2 * i + 1] = 0.3f * (float)sin(ph + 0.02 * i) + 0.02f * frnd();
    }
    /* exact-size output buffers so that any overrun trips ASan: S <= count/1.9 + 2 */
    static const int chunks[] = { 1, 2, 7, 63, 64, 65, 180, 4001, 20000 };
    for (unsigned k = 0; k < sizeof(chunks) / sizeof(chunks[0]); k++) {
        const int ch = chunks[k];
        tetra_oracle_state_t* st = malloc(sizeof(*st));
        tetra_oracle_reset(tab, st);
        const int cap = (int)(ch / 1.9) + 2;
        float* x = malloc(sizeof(float) * 2 * ch); float* y = malloc(sizeof(float) * 2 * ch);
        float* sym = malloc(sizeof(float) * 2 * cap); uint8_t* dib = malloc(cap); uint8_t* bits = malloc(2 * cap);
        long total = 0;
        for (int pos = 0; pos < N; pos += ch) {
            const int c = N - pos < ch ? N - pos : ch;
            const int S = tetra_oracle_process(tab, st, c, iq + 2 * pos, x, y, sym, dib, bits);
            CHECK(S >= 0 && S <= cap);
            total += S;
            if (pos == 3 * ch) tetra_oracle_reset_reference(tab, st);          /* rrc_valid = 0 path */
        }
        CHECK(total > N / 2 - 200 && total < N / 2 + 200);
        free(x); free(y); free(sym); free(dib); free(bits); free(st);
    }
    /* setters, both modes, interleaved with short calls; tap count down to 2 and up to the maximum */
    for (int quirks = 0; quirks < 2; quirks++) {
        tetra_oracle_state_t* st = malloc(sizeof(*st));
        CHECK(tetra_oracle_design(&cfg, tab) == 0);
        tetra_oracle_reset(tab, st);
        float sym[2 * 600]; uint8_t dib[600], bits[1200];
        static const struct { int id; double v; } seq[] = { { 4, 0.03 }, { 5, 0.004 }, { 6, 0.008 }, { 7, 2e-4 }, { 8, 0.02 },
            { 9, 0.03 }, { 3, 0.5 }, { 2, 2 }, { 2, TETRA_ORACLE_MAX_TAPS }, { 2, 33 }, { 0, 17000 }, { 1, 37000 }, { 3, 1.7 }, { 2, 65 } };
        int pos = 0;
        for (unsigned k = 0; k < sizeof(seq) / sizeof(seq[0]); k++) {
            const int old = tab->ntaps;
            CHECK(tetra_oracle_set_param(tab, seq[k].id, seq[k].v, quirks) == 0);
            if (quirks && tab->ntaps > old) tetra_oracle_rrc_taps_grown(st, old);
            if (seq[k].id < 2) tetra_oracle_reset_timing(tab, st);
            const int c = 17 + 61 * (int)k;
            CHECK(tetra_oracle_process(tab, st, c, iq + 2 * pos, NULL, NULL, sym, dib, bits) <= 600);
            pos += c;
        }
        CHECK(tetra_oracle_set_param(tab, 2, 1, quirks) < 0 && tetra_oracle_set_param(tab, 2, TETRA_ORACLE_MAX_TAPS + 1, quirks) < 0);
        CHECK(tetra_oracle_set_param(tab, 77, 1.0, quirks) < 0);
        free(st);
    }
    /* batch driver with threads, rows exactly as large as the contract says */
    {
        CHECK(tetra_oracle_design(&cfg, tab) == 0);
        const int C = 5, n = 4000, stride = ((int)(n / 0.95) + 16 + 15) / 16 * 16;
        tetra_oracle_state_t* sts = malloc(sizeof(*sts) * C);
        for (int c = 0; c < C; c++) tetra_oracle_reset(tab, &sts[c]);
        uint8_t* bits = malloc((size_t)C * stride); int32_t nb[5]; float* sym = malloc(sizeof(float) * C * stride);
        CHECK(tetra_oracle_process_batch(tab, sts, C, n, 0, 0, iq, bits, stride, nb, sym) == 0);
        CHECK(tetra_oracle_process_batch(tab, sts, C, n, 180, 2, iq, bits, stride, nb, NULL) == 0);
        CHECK(tetra_oracle_process_batch(tab, sts, C, n, 0, 1, iq, bits, 64, nb, NULL) < 0);    /* too-small rows are refused */
        free(sts); free(bits); free(sym);
    }
    float s, c;
    for (int i = -200; i <= 200; i++) tetra_oracle_sincosf(0.05f * i, &s, &c);
    tetra_oracle_sincosf(1e30f, &s, &c); tetra_oracle_sincosf(-1e30f, &s, &c);
    free(iq); free(tab);
    return 0;
}

static int run_bsync(void) {
    /* training sequences as the search knows them are found by the oracle itself in a stream that repeats every 510 bits */
    const int nslots = 60, N = 510 * nslots + 777;
    uint8_t* tx = malloc(N + 64);
    for (int i = 0; i < N + 64; i++) tx[i] = (uint8_t)(rnd() & 1);
    /* ETSI EN 300 392-2 9.4.4.3.4 synchronisation training sequence, 38 bits (public standard) */
    static const uint8_t y[38] = { 1,1,0,0,0,0,0,1,1,0,0,1,1,1,0,0,1,1,1,0,1,0,0,1,1,1,0,0,0,0,0,1,1,0,0,1,1,1 };
    for (int s = 0; s < nslots; s++)
        if (s % 7 != 5) memcpy(tx + 777 + 510 * s + 214, y, 38);       /* every 7th burst loses its sequence: lock loss */
    unsigned off = 0;
    CHECK(bs_oracle_find_train_seq(tx, (unsigned)N, 1u << 3, &off) >= -1);
    static const int chunks[] = { 1, 2, 100, 509, 510 };   /* LOCKED consumes one 510-bit frame per call: longer calls overrun the
                                                             * reference's own 4096-bit buffer (tetra_burst_sync.c:38-51,98-101), so they
                                                             * are outside the state machine's domain here too */
    for (unsigned k = 0; k < sizeof(chunks) / sizeof(chunks[0]); k++) {
        bs_oracle_state_t* st = malloc((size_t)bs_oracle_state_size());
        bs_oracle_reset(st);
        const int cap = N / 510 + 16;
        uint8_t* frames = malloc((size_t)cap * 512); int32_t* types = malloc(sizeof(int32_t) * cap); uint32_t* bn = malloc(sizeof(uint32_t) * cap);
        const int nf = bs_oracle_feed(st, tx, N, chunks[k], frames, types, bn, cap);
        CHECK(nf > 10 && nf <= cap);
        uint8_t out[432];
        for (int f = 0; f < nf; f++)
            for (int tp = 0; tp < 6; tp++)
                for (int blk = 0; blk < 3; blk++) CHECK(bs_oracle_demux(frames + 512 * f, types[f], tp, blk, out) <= 432);
        free(frames); free(types); free(bn); free(st);
    }
    /* searches whose end sits right at the end of the allocation's 21-byte look-ahead */
    for (int n = 1; n < 80; n++) {
        uint8_t* row = malloc((size_t)n + 21);
        for (int i = 0; i < n + 21; i++) row[i] = (uint8_t)(rnd() & 1);
        if (n >= 38) memcpy(row + n - 38, y, 38);
        (void)bs_oracle_find_train_seq(row, (unsigned)n, 0x1f, &off);
        free(row);
    }
    free(tx);
    return 0;
}

static int run_chan(void) {
    const int M = 32, P = 8, D = 25, L = M * P;
    float* h = malloc(sizeof(float) * L);
    chan_oracle_prototype(M, P, 0.8, h);
    float* hist = calloc((size_t)2 * (L - 1), sizeof(float));
    int phase = 0; int64_t fi = 0;
    static const int calls[] = { 1, 3, 24, 25, 26, 1000, 7, 255, 256, 5000 };
    for (unsigned k = 0; k < sizeof(calls) / sizeof(calls[0]); k++) {
        const int n = calls[k];
        float* x = malloc(sizeof(float) * 2 * n);
        for (int i = 0; i < 2 * n; i++) x[i] = frnd();
        const int nf_expect = (phase + n) / D;
        float* out = malloc(sizeof(float) * 2 * (size_t)(nf_expect ? nf_expect : 1) * M);
        CHECK(chan_oracle_process(M, P, D, h, hist, &phase, &fi, n, x, out) == nf_expect);
        free(x); free(out);
    }
    free(h); free(hist);
    return 0;
}

int main(void) {
    if (run_demod()) return 1;
    if (run_bsync()) return 2;
    if (run_chan()) return 3;
    puts("san_oracle: ok");
    return 0;
}
