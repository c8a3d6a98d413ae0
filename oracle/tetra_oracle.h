/*
 * tetra_oracle.h -- CPU restatement of the reference's pi/4-DQPSK demodulator chain.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (include/, the
 * sdrpp-tetra-demodulator_amd/ package, the C-ABI library) may include, link or
 * call anything in oracle/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED: the reference (cropinghigh/sdrpp-tetra-demodulator) ships no
 * tests, golden vectors or fixtures for this path, and its src/dsp sources cannot
 * be compiled in this image without writing stand-ins for the absent SDR++ core
 * headers and VOLK (forbidden), so this restatement is NOT checked against
 * reference outputs.  It follows the reference source line by line (citations on
 * every function) plus the SDR++-core semantics listed in SURVEY.md Appendix A,
 * and is validated functionally: known-answer tests recover the transmitted bits
 * (ETSI bit<->phase map of src/decoder/src/phy/tetra_burst.c:99-117) with the
 * constant lag the reference chain has.
 *
 * Arithmetic contract (this is what the HIP kernels must reproduce bit for bit):
 *   - everything is IEEE-754 binary32, round-to-nearest-even, no contraction
 *     (compile with -ffp-contract=off), subnormals kept;
 *   - every dot product (FLL band-edge FIRs, RRC FIR, 8-tap interpolator rows) is
 *     one fmaf chain per real sum, accumulator starting at +0.0f, taps applied in
 *     ascending tap index (oldest sample first) -- the order of a scalar
 *     `for k: acc += hist[k]*tap[k]` loop (VOLK's own order is SIMD-ISA dependent,
 *     so reference floats are not bit-reproducible across machines anyway);
 *   - the two 65-tap complex band-edge FIRs of the FLL are evaluated through the
 *     four real sums S1=sum xr*a, S2=sum xi*b, S3=sum xr*b, S4=sum xi*a of the
 *     conjugate tap pair tL=a+jb, tH=a-jb (src/dsp/fll.cpp:89-93 builds exactly
 *     conjugate taps);
 *   - all loop arithmetic (AGC, phase-control loops, error functions, complex
 *     multiplies) is plain non-fused mul/add in the order the reference source
 *     writes it;
 *   - cosf/sinf of the run-time loop phases go through tetra_oracle_sincosf()
 *     below (a fixed polynomial, |error| <= 1.6e-7), because host libm and GPU ocml
 *     differ in the last ulp; init-time tap design uses the host libm like the
 *     reference does;
 *   - sqrtf is the correctly rounded IEEE square root.
 */
#ifndef TETRA_ORACLE_H
#define TETRA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TETRA_ORACLE_MAX_TAPS 129   /* rrc tap count limit of the restatement */
#define TETRA_ORACLE_INTERP_PHASES 128
#define TETRA_ORACLE_INTERP_TAPS 8

/* The ten parameters of PI4DQPSK::init (src/dsp/pi4dqpsk.h:36). */
typedef struct {
    double symbolrate;       /* 18000   src/main.cpp:84 */
    double samplerate;       /* 36000   src/main.cpp:35 */
    int    rrc_tap_count;    /* 65      src/main.cpp:40 */
    double rrc_beta;         /* 0.35    src/main.cpp:41 */
    double agc_rate;         /* 0.02    src/main.cpp:42 */
    double costas_bandwidth; /* 0.01    src/main.cpp:43 */
    double fll_bandwidth;    /* 0.006   src/main.cpp:44 */
    double omega_gain;       /* src/main.cpp:82 recov_omega */
    double mu_gain;          /* src/main.cpp:81 recov_mu */
    double omega_rel_limit;  /* 0.02    src/main.cpp:39 */
} tetra_oracle_cfg_t;

/* Per-channel loop state, in the reference's own terms. */
typedef struct {
    float agc_gain;                 /* FastAGC _gain */
    float fll_phase, fll_freq;      /* FLL pcl.phase / pcl.freq (fll.h:58) */
    float hist[2 * (TETRA_ORACLE_MAX_TAPS - 1)]; /* the last 128 FLL outputs (re,im), newest last: delay line of the three FIRs
                                                  * (each uses its own last taps-1 samples) */
    float mu, omega;                /* COMPLEX_FD pcl.phase / pcl.freq (complex_fd.h:57) */
    int32_t offset;                 /* COMPLEX_FD offset (complex_fd.h:72) */
    float ybuf[2 * (TETRA_ORACLE_INTERP_TAPS - 1)]; /* COMPLEX_FD delay buffer */
    float costas_phase, costas_freq;/* PLL pcl */
    float ph2;                      /* pi4dqpsk_costas.h:32 */
    uint8_t prev;                   /* dqpsk_sym_extr.h:42 */
    /* sync/quality statistic of DQPSKSymbolExtractor (dqpsk_sym_extr.h:36-46, .cpp:9-31).  The reference leaves
     * errorbuf uninitialised; a fresh state here starts it at zero. */
    float errorbuf[4096];           /* SYNC_DETECT_BUF */
    int32_t errorptr, errordisplayptr;
    float standarderr;
    int32_t sync;
    /* How many of the newest delay-line samples the RRC FIR may see (saturates at 128).  The reference keeps three FIR
     * objects with a delay line each; PI4DQPSK::reset clears only the RRC's (rrc.reset(), pi4dqpsk.cpp:125 -- FLL::reset,
     * fll.cpp:120-127, resets the loop and leaves its two band-edge FIRs' lines alone).  With ONE shared line here, the
     * reference's reset is "keep the line, RRC sees none of it": rrc_valid = 0.  Found by tests/test_reference_shim.py. */
    int32_t rrc_valid;
} tetra_oracle_state_t;

/* Derived constants + tables, shared by all channels. */
typedef struct {
    tetra_oracle_cfg_t cfg;
    int   ntaps;                              /* rrc_tap_count = length of the RRC FIR */
    int   ntaps_be;                           /* length of the FLL's band-edge FIRs: the tap count PI4DQPSK::init was called with;
                                               * PI4DQPSK's setters never touch the FLL's filters (pi4dqpsk.cpp:32-74), so with
                                               * reference quirks it stays at that value when setRRCTapCount changes ntaps */
    float rrc[TETRA_ORACLE_MAX_TAPS];         /* taps::rootRaisedCosine */
    float be_a[TETRA_ORACLE_MAX_TAPS];        /* band-edge tap real part (tL.re == tH.re) */
    float be_b[TETRA_ORACLE_MAX_TAPS];        /* band-edge tap imag part of tL (tH.im = -b) */
    float bank[TETRA_ORACLE_INTERP_PHASES][TETRA_ORACLE_INTERP_TAPS];
    float agc_rate, agc_set_point, agc_max_gain;
    float fll_alpha, fll_beta, fll_min_freq, fll_max_freq;
    float tr_alpha /*mu gain*/, tr_beta /*omega gain*/, tr_min_freq, tr_max_freq, tr_omega;
    float costas_alpha, costas_beta, costas_min_freq, costas_max_freq;
} tetra_oracle_tables_t;

void tetra_oracle_default_cfg(tetra_oracle_cfg_t* cfg);
/* 0 on success, <0 on bad parameters. */
int  tetra_oracle_design(const tetra_oracle_cfg_t* cfg, tetra_oracle_tables_t* tab);
void tetra_oracle_reset(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st);

/*
 * The setters of PI4DQPSK (pi4dqpsk.h:52-63, pi4dqpsk.cpp:32-118) on an existing design; ids as in the product's
 * include/tetra_demod.h (TETRA_PARAM_*: 0 symbolrate, 1 samplerate, 2 rrc tap count, 3 rrc beta, 4 agc rate, 5 costas
 * bandwidth, 6 fll bandwidth, 7 omega gain, 8 mu gain, 9 omega rel limit).  Loop setters change loop constants only;
 * the rate and RRC setters re-design only the RRC taps (and the timing loop's nominal omega / limits); the band-edge
 * filters are never re-designed from here, except that WITHOUT quirks a new tap count re-designs them to the new length
 * (the product's kernels then keep one length for all three FIRs).  With quirks != 0 the reference is followed to the
 * letter: a new tap count leaves the FLL alone (pi4dqpsk.cpp:56-66) and setRRCBeta(int) truncates its argument
 * (pi4dqpsk.h:56, pi4dqpsk.cpp:72).  Rate setters also reset the timing loop of every state the caller owns
 * (tetra_oracle_reset_timing), like COMPLEX_FD::setOmega (complex_fd.cpp:30-41).  Returns 0, <0 on a bad value.
 */
int  tetra_oracle_set_param(tetra_oracle_tables_t* tab, int param_id, double value, int quirks);
void tetra_oracle_reset_timing(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st);
/*
 * PI4DQPSK::reset (pi4dqpsk.cpp:120-130) as the reference does it: AGC gain, FLL phase/freq, the FIR delay line, PLL
 * phase/freq and the timing loop are reset; ph2 (pi4dqpsk_costas.h:32), COMPLEX_FD's delay buffer and the symbol
 * extractor (another block) keep their values.  The reference clears only the RRC's delay line and leaves the
 * band-edge FIRs' own: with ONE shared line here that is "keep the line, rrc_valid = 0" (see the state struct).
 */
void tetra_oracle_reset_reference(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st);
/* After a quirks-mode setRRCTapCount that made the RRC longer: see the .c file. */
void tetra_oracle_rrc_taps_grown(tetra_oracle_state_t* st, int old_ntaps);

void tetra_oracle_sincosf(float x, float* s, float* c);

/*
 * One PI4DQPSK::process + DQPSKSymbolExtractor::process + BitUnpacker::process
 * call on one channel.  iq = count interleaved (re,im) samples.
 * Optional outputs (NULL to skip):
 *   x_out   [2*count]  FLL output (input of the RRC filter)
 *   y_out   [2*count]  RRC output (input of timing recovery)
 *   sym_out [2*S]      Costas output = PI4DQPSK::process output
 *   dibits  [S]        DQPSKSymbolExtractor output
 *   bits    [2*S]      BitUnpacker output (what tetra_burst_sync_in eats)
 * Output capacity needed: S <= (count + 1) / (tr_min_freq - |tr_alpha|) + 1 (count/1.9 + 2 at the plugin's rates; count + 1
 * whenever every symbol advances by at least one sample).  Returns S (symbols produced).
 */
int tetra_oracle_process(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st,
                         int count, const float* iq,
                         float* x_out, float* y_out, float* sym_out,
                         uint8_t* dibits, uint8_t* bits);

/*
 * The same call with the float recipe chosen:
 *   TETRA_ORACLE_CONTRACT          the arithmetic contract above (what tetra_oracle_process computes and the HIP kernels
 *                                  reproduce bit for bit);
 *   TETRA_ORACLE_REFERENCE_FLOATS  the reference's own recipe as an x86 build without VOLK SIMD kernels executes it: host libm
 *                                  cosf / sinf for every math::phasor (fll.cpp:137, pi4dqpsk_costas.cpp:7,16), the FLL's two
 *                                  band-edge FIRs as two separate complex x complex dot products (fll.cpp:141-142 ->
 *                                  volk_32fc_x2_dot_prod_32fc), every dot product a plain `acc += a * b` in ascending tap order,
 *                                  no fmaf anywhere.  Same tables, same state, same control flow.
 * Purpose: in this mode the restatement must equal the reference's src/dsp objects -- compiled from
 * /root/reference against the stand-in SDR++ core headers of tests/refshim -- in every symbol FLOAT, every bit and the
 * final loop state, bit for bit (tests/test_reference_shim.py), which turns "bits equal, symbols within 3e-3" into an
 * exact statement about the transcription; the contract mode then differs from it only by the documented recipe.
 * Returns S, or -1 for an unknown mode.
 */
enum { TETRA_ORACLE_CONTRACT = 0, TETRA_ORACLE_REFERENCE_FLOATS = 1 };
int tetra_oracle_process_mode(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* st, int mode,
                              int count, const float* iq,
                              float* x_out, float* y_out, float* sym_out,
                              uint8_t* dibits, uint8_t* bits);

/*
 * Batched driver used for the CPU baseline and for big parity tests: C channels,
 * channel-major iq[C][n_samples] (interleaved re,im), processed in `chunk`-sample
 * process() calls (chunk<=0: one call), channels split over `threads` OpenMP
 * threads (<=0: all).  bits[C][bits_stride], n_bits[C]; sym (optional)
 * [C][bits_stride/2] complex.  states[C] carried in/out (must be initialised).
 * Returns 0, or <0 if an output row would overflow.
 */
int tetra_oracle_process_batch(const tetra_oracle_tables_t* tab, tetra_oracle_state_t* states,
                               int n_channels, int n_samples, int chunk, int threads,
                               const float* iq, uint8_t* bits, int bits_stride,
                               int32_t* n_bits, float* sym);

int tetra_oracle_max_threads(void);

/* fmaf-chain matrix product (ascending k from +0): checker for the GPU's matrix-pipe self-test. */
void tetra_oracle_fmaf_chain_matmul(const float* a, const float* b, int m, int n, int k, float* d);

#ifdef __cplusplus
}
#endif
#endif
