/*
 * tetra_demod.h -- C ABI of the MI355X-native batched TETRA pi/4-DQPSK demodulator.
 *
 * This is the drop-in boundary for ONE path of cropinghigh/sdrpp-tetra-demodulator: the
 * src/dsp chain AGC -> FLL -> RRC -> timing recovery -> Costas -> symbol slicer ->
 * differential decoder -> bit unpacker.  A handle owns C independent channels; each channel
 * is exactly one reference chain
 *     dsp::demod::PI4DQPSK        (src/dsp/pi4dqpsk.h:27-81, process() src/dsp/pi4dqpsk.cpp:132-140)
 *  -> dsp::DQPSKSymbolExtractor   (src/dsp/dqpsk_sym_extr.h:19-46, process() src/dsp/dqpsk_sym_extr.cpp:4-55)
 *  -> dsp::BitUnpacker            (src/dsp/bit_unpacker.h:16-34, process() src/dsp/bit_unpacker.cpp:4-10)
 * wired as in src/main.cpp:84,90-91.  The single-channel dsp::block wrapper of
 * sdrpp-tetra-demodulator_amd/host/ is C = 1.
 *
 * Conventions: extern "C", plain pointers and sizes, int status (0 = ok, < 0 = error), no
 * exceptions, no global state.  A handle is driven by one thread at a time (the reference runs
 * one worker thread per block and takes ctrlMtx + tempStop() around setters,
 * src/dsp/pi4dqpsk.cpp:32-42).  All work runs on the GPU; there is no CPU fallback and every
 * entry point fails with TETRA_ERR_NO_DEVICE / TETRA_ERR_HIP when no usable HIP device exists.
 */
#ifndef TETRA_DEMOD_H
#define TETRA_DEMOD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TETRA_DEMOD_ABI_VERSION 6
#define TETRA_DEMOD_MAX_TAPS 129 /* longest filter of this ABI and capacity every tap table handed across it is sized for */

enum {
    TETRA_OK = 0,
    TETRA_ERR_ARG = -1,         /* NULL / out-of-range argument */
    TETRA_ERR_UNSUPPORTED = -2, /* parameter outside what the library implements (> 129 taps; a timing loop that may stall: min_step <= 0) */
    TETRA_ERR_NO_DEVICE = -3,   /* no HIP device / bad ordinal */
    TETRA_ERR_HIP = -4,         /* a HIP runtime call failed (see tetra_demod_last_hip_error) */
    TETRA_ERR_NOMEM = -5,
    TETRA_ERR_SIZE = -6,        /* n_samples > max_samples, or an output stride too small */
    TETRA_ERR_ALIGN = -7,       /* output pointer / stride not 8-byte aligned */
    TETRA_ERR_OVERRUN = -8      /* a channel's output row filled up and the rest of its samples were dropped (only a NaN/Inf-poisoned
                                   channel can do that: rows are sized for the slowest finite timing loop, tetra_demod_bits_stride_for;
                                   with min_step < 1 such a channel stops advancing and fills its row from one offset).
                                   The outputs of every other channel are valid and were delivered; see tetra_demod_get_overruns */
};

/* tetra_demod_config_t.flags */
enum {
    TETRA_FLAG_RETIRED_TWO_KERNEL = 1, /* ABI 1's two-kernel pipeline was retired in ABI 2: tetra_demod_create refuses the flag
                                    (TETRA_ERR_UNSUPPORTED). */
    TETRA_FLAG_KEEP_RRC_OUT = 2, /* also keep the RRC output in an HBM scratch for tetra_demod_debug_read_rrc_out */
    TETRA_FLAG_QUALITY = 4,      /* also compute DQPSKSymbolExtractor's sync/quality statistic (tetra_demod_get_quality) */
    TETRA_FLAG_WIDE_WORKGROUPS = 16,   /* force 32-channel workgroups / ... */
    TETRA_FLAG_NARROW_WORKGROUPS = 32, /* ... or 16-channel ones / ... */
    TETRA_FLAG_SMALL_WORKGROUPS = 64,  /* ... or 4-channel ones (more than one of the three: TETRA_ERR_ARG).  Default: chosen from the
                                    channel count -- 4-channel workgroups (FLL rows of 16 lanes per channel: the shortest
                                    per-sample step) while every workgroup has a CU to itself (<= 4 channels per CU: 1024 channels
                                    on an MI355X); 16-channel ones up to one per CU (4096 channels); beyond, whole rounds of
                                    32-channel workgroups and the rest in whichever shape is through first -- 8192 channels take
                                    1.3x the 4096-channel time instead of 2x.  Results are identical bit for
                                    bit.  Band-edge filters of more than 68 taps (rrc_tap_count 69..72) never run in 32-channel
                                    workgroups. */
    TETRA_FLAG_CONSTELLATION = 256,    /* also keep the constellation diagram's 1024-symbol blocks per channel (tetra_demod_get_constellation) */
    TETRA_FLAG_GENERIC_KERNEL = 128,   /* filters of 73 .. 129 taps and timing loops below 0.27 samples per symbol in the one-lane-per-channel
                                        * kernel instead of the fused kernel's long rows / deepest symbol ring (same results; tests and A/B
                                        * measurements) */
    TETRA_FLAG_REFERENCE_QUIRKS = 8 /* follow the reference to the letter where this library otherwise tidies up (the C++ mirror of
                                    the dsp::block sets it):
                                      - tetra_demod_reset keeps ph2 (src/dsp/pi4dqpsk_costas.h:32 is never reset by
                                        PI4DQPSK::reset, pi4dqpsk.cpp:120-130), COMPLEX_FD's delay buffer (complex_fd.cpp:78-87
                                        does not clear it) and the slicer's previous symbol (another block); it clears the
                                        delay line for the RRC only -- the FLL's band-edge FIRs keep theirs (fll.cpp:120-127);
                                      - TETRA_PARAM_RRC_TAP_COUNT / tetra_demod_set_rrc_params re-design only the RRC and leave the FLL's
                                        band-edge filters at their construction-time length (pi4dqpsk.cpp:56-70).
                                    (The reference's setRRCBeta(int) truncation, src/dsp/pi4dqpsk.h:56, lives in the C++ mirror's
                                    setRRCBeta(int) signature, not here: TETRA_PARAM_RRC_BETA always takes the double.) */
};

/* Input sample layout of process(): element (channel c, sample n) of the complex64 stream. */
enum {
    TETRA_LAYOUT_CHANNEL_MAJOR = 0, /* iq[c][n]: each channel contiguous = the reference's complex_t in[count] per channel */
    TETRA_LAYOUT_TIME_MAJOR = 1     /* iq[n][c]: one frame per sample instant (what a polyphase channeliser emits) */
};

/*
 * Configuration = channel count + the ten arguments of PI4DQPSK::init (src/dsp/pi4dqpsk.h:36),
 * with the reference plugin's values (src/main.cpp:35-44,78-84) as defaults.
 */
typedef struct tetra_demod_config {
    int32_t n_channels;      /* C >= 1 */
    int32_t max_samples;     /* largest n_samples of one process() call (reference: count <= STREAM_BUFFER_SIZE = 1e6) */
    int32_t layout;          /* TETRA_LAYOUT_* */
    int32_t device;          /* HIP device ordinal; -1 = current device */
    double symbolrate;       /* 18000 */
    double samplerate;       /* 36000 */
    int32_t rrc_tap_count;   /* 65 (the reference builds with RRC_TAP_COUNT 65, src/main.cpp:36); 2..129.  Up to 72 taps run in the fused
                              * kernel's regular rows; 73..129 in its LONG variant (FLL rows of 16 x 9 taps in 4-channel workgroups up to
                              * 1024 channels, 8 x 17 in 16-channel ones beyond: about three quarters of the 65-tap rate), or with
                              * TETRA_FLAG_GENERIC_KERNEL in the generic kernel (one lane per channel) -- bit-identical results */
    int32_t flags;           /* TETRA_FLAG_* */
    double rrc_beta;         /* 0.35 */
    double agc_rate;         /* 0.02 */
    double costas_bandwidth; /* 0.01 */
    double fll_bandwidth;    /* 0.006 */
    double omega_gain;       /* timing loop beta, src/main.cpp:82 */
    double mu_gain;          /* timing loop alpha, src/main.cpp:81 */
    double omega_rel_limit;  /* 0.02.  Accepted: 0 <= limit < 1 with min_step = samplerate / symbolrate x (1 - limit) - |mu_gain| > 0
                              * samples per symbol.  Below min_step = 1 the reference emits several symbols from one offset
                              * (floor(mu) = 0, complex_fd.cpp:141-143) and so do the kernels (ABI 4): down to 0.27 in the fused
                              * kernel (16- and 4-channel workgroups with a deeper symbol ring), down to 0.07 in its 4-channel
                              * workgroups with a 1024-deep one, below that in the generic kernel.
                              * Refused (TETRA_ERR_UNSUPPORTED, also from the setters): min_step <= 0, where the reference's own
                              * loop may never leave process() */
    /* Optional caller-supplied tables (NULL = design them like the reference does).  In an SDR++
     * build the host may pass SDR++'s own tap generators' output here. */
    const float* rrc_taps;        /* [rrc_tap_count]                 taps::rootRaisedCosine, pi4dqpsk.cpp:18 */
    const float* bandedge_taps;   /* [2][rrc_tap_count]: re, im of the LOWER band-edge filter, fll.cpp:61-95 */
    const float* interp_bank;     /* [128][8]                        complex_fd.cpp:153-158 */
} tetra_demod_config_t;

/* Per-channel loop state in the reference's own terms (checkpoint / tests). */
typedef struct tetra_demod_channel_state {
    float agc_gain;                  /* FastAGC gain */
    float fll_phase, fll_freq;       /* FLL pcl (src/dsp/fll.h:58) */
    float mu, omega;                 /* COMPLEX_FD pcl.phase / pcl.freq (src/dsp/complex_fd.h:57) */
    int32_t offset;                  /* COMPLEX_FD offset (src/dsp/complex_fd.h:72) */
    float costas_phase, costas_freq; /* PLL pcl of PI4DQPSK_COSTAS */
    float ph2;                       /* src/dsp/pi4dqpsk_costas.h:32 */
    int32_t prev;                    /* DQPSKSymbolExtractor prev (src/dsp/dqpsk_sym_extr.h:42) */
    float hist[2 * 80];              /* last 80 FLL outputs (re,im), newest last: FIR delay lines (only the last taps-1 matter) */
    float ybuf[2 * 7];               /* COMPLEX_FD delay buffer: last 7 RRC outputs */
    int32_t rrc_valid;               /* how many of the newest delay-line samples the RRC FIR may see, 0..128 (>= taps - 1 = all; older
                                      * ones are zeros to it).  The reference keeps a delay line per FIR object; rrc.reset()
                                      * (pi4dqpsk.cpp:125) and a growing FIR::setTaps clear/zero-fill the RRC's only.  The fused
                                      * kernel's regular rows keep 80 samples and saturate the count at 80 (its long rows and the
                                      * generic kernel: 128). */
    float hist_far[2 * 48];          /* the 48 FLL outputs BEFORE hist[] (oldest first): only filters of more than 81 taps look that
                                      * far back.  Kept by the fused kernel's long rows and by the generic kernel; a launch of the fused kernel's regular rows (<= 72 taps) does not
                                      * carry them, after it they read as zeros -- to tetra_demod_get_state and to a filter that a
                                      * setter grows beyond 81 taps. */
} tetra_demod_channel_state_t;

/* IDs for tetra_demod_set_param: the setters of PI4DQPSK (src/dsp/pi4dqpsk.h:52-63). */
enum {
    TETRA_PARAM_SYMBOLRATE = 0,       /* setSymbolrate      pi4dqpsk.cpp:32-42  (also resets timing recovery, complex_fd.cpp:30-41) */
    TETRA_PARAM_SAMPLERATE = 1,       /* setSamplerate      pi4dqpsk.cpp:44-54 */
    TETRA_PARAM_RRC_TAP_COUNT = 2,    /* setRRCTapCount     pi4dqpsk.cpp:68-70.  Without TETRA_FLAG_REFERENCE_QUIRKS a NEW count also re-designs the
                                       * FLL's band-edge filters to that length (what a fresh init with that count gives); with the flag only
                                       * the RRC changes, like the reference */
    TETRA_PARAM_RRC_BETA = 3,         /* the roll-off of setRRCParams, pi4dqpsk.cpp:56-66: a double, never truncated here (the reference's
                                       * setRRCBeta(int) truncates in its signature; so does the C++ mirror's) */
    TETRA_PARAM_AGC_RATE = 4,         /* setAGCRate         pi4dqpsk.cpp:76-80 */
    TETRA_PARAM_COSTAS_BANDWIDTH = 5, /* setCostasBandwidth pi4dqpsk.cpp:82-86 */
    TETRA_PARAM_FLL_BANDWIDTH = 6,    /* setFllBandwidth    pi4dqpsk.cpp:88-92 */
    TETRA_PARAM_OMEGA_GAIN = 7,       /* setOmegaGain       pi4dqpsk.cpp:102-106 */
    TETRA_PARAM_MU_GAIN = 8,          /* setMuGain          pi4dqpsk.cpp:108-112 */
    TETRA_PARAM_OMEGA_REL_LIMIT = 9   /* setOmegaRelLimit   pi4dqpsk.cpp:114-118 */
};

typedef struct tetra_demod tetra_demod_t;

/* Fill cfg with the reference plugin's parameters (src/main.cpp:35-44,78-84), C = 1. */
int tetra_demod_default_config(tetra_demod_config_t* cfg);

/* Number of HIP devices visible (0 if none); the per-GPU channel ranges of a multi-GPU host are
 * one handle per device (cfg.device). */
int tetra_demod_device_count(void);

/* Replaces PI4DQPSK::init + DQPSKSymbolExtractor::init + BitUnpacker::init (src/main.cpp:84,90-91)
 * for C channels: designs the taps, allocates device state. */
int tetra_demod_create(const tetra_demod_config_t* cfg, tetra_demod_t** out);
int tetra_demod_destroy(tetra_demod_t* h);

/* Row length (bytes = bits) that holds any call of n_samples on THIS handle: 2 x the largest symbol count its timing loop
 * can emit, K <= (n + 1) / (samplerate / symbolrate x (1 - omega_rel_limit) - |mu_gain|) + 1, plus margin, a multiple of 16.
 * process* refuse a smaller bits_stride (TETRA_ERR_SIZE).  Changes with the rate and timing setters.  The reference has no
 * such limit because its output is a STREAM_BUFFER_SIZE stream buffer (complex_fd.cpp:89-151). */
int tetra_demod_bits_stride_for(tetra_demod_t* h, int n_samples);
/* The same without a handle, for the common case: n/0.95 + 16 rounded up to a multiple of 16 covers every handle whose
 * slowest timing step is >= 1.9 samples per symbol (the plugin's parameters: 2 x 0.98 - 0.0176 = 1.9424), so a caller that
 * stays at ~2 samples per symbol may size rows with this one; it is never smaller than tetra_demod_bits_stride_for there. */
int tetra_demod_bits_stride(int n_samples);

/*
 * Replaces one PI4DQPSK::process + DQPSKSymbolExtractor::process + BitUnpacker::process call per
 * channel (src/dsp/pi4dqpsk.cpp:132-140, dqpsk_sym_extr.cpp:4-55, bit_unpacker.cpp:4-10).
 *   iq      n_channels x n_samples complex64 (interleaved re,im) in cfg.layout
 *   bits    [n_channels][bits_stride] uint8, one bit per byte, MSB of each dibit first -- the stream
 *           tetra_burst_sync_in() consumes (src/decoder/src/phy/tetra_burst_sync.c:54)
 *   n_bits  [n_channels] int32: bits written per channel this call (2 x symbols; varies per channel
 *           because the timing loop's omega floats within +-omega_rel_limit)
 *   sym     optional [n_channels][bits_stride/2] complex64: PI4DQPSK::process output (constellation
 *           points after the Costas loop), NULL to skip
 * Loop state is carried across calls; the result is independent of how a stream is cut into calls.
 * _device: all pointers are device pointers on the handle's GPU, work is enqueued on `hip_stream`
 * (a hipStream_t, NULL = default stream) and the call returns without synchronising.
 * Host variant: pointers are host memory; copies in, runs, copies out, synchronises.  Hand it page-locked buffers
 * (hipHostMalloc / hipHostRegister) and the copies run as DMA at PCIe rate (measured 2x the pageable rate).
 */
int tetra_demod_process_device(tetra_demod_t* h, const float* d_iq, int n_samples, uint8_t* d_bits,
                               int bits_stride, int32_t* d_n_bits, float* d_sym, void* hip_stream);
int tetra_demod_process(tetra_demod_t* h, const float* iq, int n_samples, uint8_t* bits, int bits_stride,
                        int32_t* n_bits, float* sym);
/* Device-resident and synchronous: the pointers of tetra_demod_process_device, enqueued on the handle's OWN stream (created on
 * first use, non-blocking: handles driven from different host threads -- one per GPU, or several on one GPU -- overlap), and
 * the call returns when the launch has finished, with TETRA_ERR_OVERRUN if it cut a channel off (below).  This is what a
 * per-GPU worker thread of a multi-GPU host calls when the samples are already in that GPU's memory. */
int tetra_demod_process_resident(tetra_demod_t* h, const float* d_iq, int n_samples, uint8_t* d_bits, int bits_stride,
                                 int32_t* d_n_bits, float* d_sym);
/* Only bits[c][0 .. n_bits[c]) (and sym[c][0 .. n_bits[c]/2)) are defined by a call; the rest of a row keeps whatever it held.
 *
 * A channel can only fill its row if NaN/Inf has poisoned its timing loop (then every symbol advances one sample).  Such a
 * channel is cut off at the row's capacity, the rest of its samples of that call are dropped, and the event is REPORTED:
 * tetra_demod_process and tetra_demod_wait return TETRA_ERR_OVERRUN (every output was delivered; the other channels'
 * are valid), and tetra_demod_get_overruns gives the number of (channel, launch) events since create -- the way to learn
 * of it after tetra_demod_process_device.  Synchronises. */
int tetra_demod_get_overruns(tetra_demod_t* h, long long* total);

/*
 * Asynchronous host entry point: the same call as tetra_demod_process (host buffers in, host buffers out, state carried),
 * enqueued and pipelined instead of copy -> kernel -> copy.  The call is cut along the time axis into up to eight chunks;
 * while chunk k is demodulated, chunk k+1 crosses PCIe into the other half of a double buffer, and the call's bits go back on
 * a third stream -- so up to two calls may be in flight and the steady state is bounded by the input copy alone.  The bits
 * are exactly those of tetra_demod_process on the same samples (chunks carry state like calls do).
 *   iq         host memory, in cfg.layout.  PAGE-LOCKED memory (tetra_demod_host_alloc, hipHostMalloc, hipHostRegister)
 *              is what makes the copies asynchronous; pageable memory works but serialises.
 *   iq_format  TETRA_IQ_CF32: complex float like the reference's complex_t stream;
 *              TETRA_IQ_CS16: interleaved int16 (re, im) as SDR hardware delivers it -- converted on the GPU as x / 32768
 *              (exact in binary32, the scaling SDR++'s sources apply on the host), half the PCIe bytes;
 *              TETRA_IQ_CS8: interleaved int8 (RTL-SDR / HackRF class front-ends), converted as x / 128, a quarter.
 *   bits / n_bits   host memory, [n_channels][bits_stride] / [n_channels]; valid after tetra_demod_wait().
 * The input and output buffers must stay untouched until tetra_demod_wait() returns.  tetra_demod_process, _reset,
 * _set_param, _get/_set_state and _destroy wait for calls in flight themselves; tetra_demod_process_device on a caller's
 * stream does not -- wait first.  No symbol output on this path.
 */
enum { TETRA_IQ_CF32 = 0, TETRA_IQ_CS16 = 1, TETRA_IQ_CS8 = 2 };
int tetra_demod_process_async(tetra_demod_t* h, const void* iq, int iq_format, int n_samples, uint8_t* bits, int bits_stride,
                              int32_t* n_bits);
/* Blocks until every tetra_demod_process_async call enqueued on this handle has delivered its output. */
int tetra_demod_wait(tetra_demod_t* h);
/* Page-locked host memory for the two calls above, for callers that do not link HIP themselves (NULL on failure). */
void* tetra_demod_host_alloc(size_t bytes);
void tetra_demod_host_free(void* p);

/* PI4DQPSK::reset (src/dsp/pi4dqpsk.cpp:120-130); channel = -1 resets all.  Resets the AGC gain, the FLL and PLL loop states,
 * the timing loop and the FIR delay line.  Without TETRA_FLAG_REFERENCE_QUIRKS it also zeroes ph2, COMPLEX_FD's delay buffer,
 * the slicer's previous symbol and the quality statistic (= a fresh chain); with the flag those keep their values like in the
 * reference.  The delay line: the reference has one per FIR object and its reset clears the RRC's only (rrc.reset(),
 * pi4dqpsk.cpp:125; FLL::reset, fll.cpp:120-127, leaves the two band-edge FIRs' lines alone).  The kernel keeps ONE line for the
 * three FIRs: without the flag it is cleared (fresh chain); with the flag it is kept and channel_state.rrc_valid = 0 hides it
 * from the RRC, which is the reference's behaviour to the letter. */
int tetra_demod_reset(tetra_demod_t* h, int channel);

/* The twelve PI4DQPSK setters collapse to one call (IDs above).  Like the reference: the loop setters (AGC rate, Costas / FLL
 * bandwidth, timing gains and limit) change loop constants and nothing else; changing a rate or the RRC design re-designs the
 * RRC taps only and keeps loop state -- no setter re-designs the FLL's band-edge filters (pi4dqpsk.cpp:32-118; the tap-count
 * exception without the quirks flag is described at TETRA_PARAM_RRC_TAP_COUNT); TETRA_PARAM_SYMBOLRATE / _SAMPLERATE
 * additionally reset the timing loop (COMPLEX_FD::setOmega, complex_fd.cpp:30-41).  Caller-supplied tables (cfg.rrc_taps,
 * cfg.bandedge_taps, cfg.interp_bank, tetra_demod_set_tables) are never re-designed by this library: the rate setters keep a
 * caller's RRC table and do the rest of their work (ABI 5; the caller follows with tetra_demod_set_tables -- ABI 4 refused the
 * call); the setters that ONLY re-design a caller's table (TETRA_PARAM_RRC_BETA / _RRC_TAP_COUNT, tetra_demod_set_rrc_params with
 * a caller's RRC table; a tap count without the quirks flag with a caller's band-edge table) return TETRA_ERR_UNSUPPORTED and
 * change nothing -- tetra_demod_set_tables is their replacement. */
int tetra_demod_set_param(tetra_demod_t* h, int param_id, double value);
/* PI4DQPSK::setRRCParams (pi4dqpsk.cpp:56-66): tap count and roll-off applied in one re-design of the RRC (same rules as
 * TETRA_PARAM_RRC_TAP_COUNT + TETRA_PARAM_RRC_BETA). */
int tetra_demod_set_rrc_params(tetra_demod_t* h, int rrc_tap_count, double rrc_beta);

/* ABI 5.  FIR::setTaps with tables the CALLER designed -- the route by which an SDR++ build runs the kernels on the output of
 * SDR++'s own generators instead of this library's restatement of them: taps::rootRaisedCosine (pi4dqpsk.cpp:18,38,50,63),
 * FLL::createBandedgeFilters through math::sinc / math::phasor (fll.cpp:61-95), taps::windowedSinc + window::nuttall +
 * multirate::buildPolyphaseBank (complex_fd.cpp:153-158).  host/pi4dqpsk_gpu.cpp does exactly that under TETRA_WITH_SDRPP, in
 * init() and in every setter that re-designs (setSymbolrate / setSamplerate / setRRCParams / setRRCTapCount / setRRCBeta).
 *   rrc_taps       [n_rrc] or NULL (keep); 2 <= n_rrc <= 129.  Replaces the RRC FIR's taps and its length (the handle's
 *                  rrc_tap_count becomes n_rrc) with the reference's setTaps rule for the delay line: a longer filter keeps its old
 *                  taps - 1 history samples and sees zeros before them (TETRA_FLAG_REFERENCE_QUIRKS; without the flag the one
 *                  delay line of the three FIRs is visible in full, as after TETRA_PARAM_RRC_TAP_COUNT).
 *   bandedge_taps  [2][n_be] or NULL (keep): re, im of the LOWER band-edge filter (the upper one is its conjugate, fll.cpp:89-93).
 *   interp_bank    [128][8] or NULL (keep).
 * Loop state, loop constants and rates are untouched.  From then on the replaced tables count as caller-supplied (see the
 * setter rules above).  Synchronises.  TETRA_ERR_ARG if all three are NULL, TETRA_ERR_UNSUPPORTED for a length outside 2..129. */
int tetra_demod_set_tables(tetra_demod_t* h, const float* rrc_taps, int n_rrc, const float* bandedge_taps, int n_be,
                           const float* interp_bank);

/* Checkpoint / restore of one channel's loop state.  set_state accepts what the chain can be in: |fll_phase| <= pi,
 * |costas_phase| <= pi, |ph2| < 2 pi (the reference's loops wrap to these ranges on every step), offset >= 0 and mu not
 * infinite (complex_fd.cpp:141-148 leaves both so; NaN = a poisoned channel passes) -- anything else is TETRA_ERR_ARG;
 * rrc_valid clamped to 0..128. */
int tetra_demod_get_state(tetra_demod_t* h, int channel, tetra_demod_channel_state_t* out);
int tetra_demod_set_state(tetra_demod_t* h, int channel, const tetra_demod_channel_state_t* in);

/* Copies of the designed tables (any pointer may be NULL): rrc[*taps], be_re[*be_taps], be_im[*be_taps] (lower band-edge
 * filter), bank[128*8].  The two lengths differ after a tap-count change under TETRA_FLAG_REFERENCE_QUIRKS (the FLL keeps its
 * construction-time filters), so a caller cannot size be_re / be_im from *taps: every tap buffer handed in must hold
 * TETRA_DEMOD_MAX_TAPS (129) floats, or call once with NULL buffers to learn both lengths first. */
int tetra_demod_get_tables(tetra_demod_t* h, int* taps, float* rrc, int* be_taps, float* be_re, float* be_im, float* bank);
int tetra_demod_bandedge_tap_count(tetra_demod_t* h);

/* DQPSKSymbolExtractor's public `standarderr` / `sync` members (src/dsp/dqpsk_sym_extr.h:36-37; computed at
 * dqpsk_sym_extr.cpp:8-31, read by the GUI at src/main.cpp:211,215) for every channel: the mean angular distance of
 * the last 4096 symbols from their ideal constellation points, refreshed every 256 symbols, and sync = that < 0.35.
 * standarderr[n_channels] / sync[n_channels] host arrays (either may be NULL).  A GUI float, not on the bit path: held
 * to a tolerance (2e-6) rather than bit equality: the ring is summed in the reference's order and precision (float, index
 * order); the distance itself is pi/4 - atan(min/max) by a polynomial instead of two libm atan2f.  Needs TETRA_FLAG_QUALITY
 * (TETRA_ERR_UNSUPPORTED otherwise).  Only the value at the last 256-symbol boundary of a call is observable, so it is
 * computed by a small kernel after the chain's launch from the symbols the chain wrote (kept in a scratch buffer of
 * n_channels x max_samples/2 complex64 when the caller does not ask for them); costs ~2.5 % of a call's time when
 * enabled (profiles/r02/r02_p_quality_statistic.md). */
int tetra_demod_get_quality(tetra_demod_t* h, float* standarderr, uint8_t* sync);

/* The plugin's constellation tap for every channel (ABI 6).  The reference splits PI4DQPSK's symbol stream into a second branch,
 * regroups it with dsp::buffer::Reshaper<complex_t>(.., keep 1024, skip 0) and copies every 1024-symbol block it delivers into the
 * GUI's diagram buffer (src/main.cpp:85-89 wiring, :376-383 _constDiagSinkHandler; drawn at :337): what is on screen is the LAST
 * COMPLETE block of 1024 consecutive symbols, blocks counted from the start of the stream.  With thousands of channels on the
 * device, streaming every symbol to the host for that costs 4 B per input sample; this keeps the regrouping on the device -- a
 * small kernel after the chain's launch carries each channel's partial block across calls -- and the host fetches 8 KB per channel
 * when it wants to draw one.
 *   symbols   [count][TETRA_CONSTELLATION_SYMBOLS][2] float (re, im), host: the last complete block of channels first ..
 *             first + count - 1 (all zero before a channel's first block completes); may be NULL
 *   n_blocks  [count] int32, host: blocks completed so far (a caller redraws when it changed); may be NULL
 * Symbols are exactly the ones the optional `sym` output of the process calls carries (bit patterns).  The block phase follows
 * the symbol count since create; tetra_demod_reset leaves the tap alone under TETRA_FLAG_REFERENCE_QUIRKS (the Reshaper is another
 * block, untouched by PI4DQPSK::reset, pi4dqpsk.cpp:119-130) and restarts it otherwise.  SDR++'s Reshaper is core code outside the
 * reference repository; keep 1024 / skip 0 regrouping is restated from its call site.  Needs
 * TETRA_FLAG_CONSTELLATION (TETRA_ERR_UNSUPPORTED otherwise); first / count outside the handle's channels: TETRA_ERR_ARG.
 * Synchronises the device.  Cost when enabled: +0.5 % of a call's time on the 4096 x 36000 workload (the chain also writes its symbols
 * to a scratch row when the caller does not ask for them; profiles/r05/r05_p_tap_cost.json). */
#define TETRA_CONSTELLATION_SYMBOLS 1024
int tetra_demod_get_constellation(tetra_demod_t* h, int first, int count, float* symbols, int32_t* n_blocks);

/* Debug/verification tap: RRC output (timing-recovery input) of the last process call,
 * y[n_channels][n_samples] complex64 channel-major, copied to host memory.  Needs
 * TETRA_FLAG_KEEP_RRC_OUT (TETRA_ERR_UNSUPPORTED otherwise). */
int tetra_demod_debug_read_rrc_out(tetra_demod_t* h, float* y, int n_samples);

/* GPU time (ms) of the most recent process call's launches (k_fused, once or twice, plus k_quality when enabled), from HIP
 * events recorded on the call's stream around them (synchronises on them). */
int tetra_demod_last_kernel_ms(tetra_demod_t* h, float* ms);
/* Same for the n (1..64) most recent kernel-launching process calls, oldest first: ms[n].  The events are recorded on each
 * call's own stream, so a benchmark can read the per-launch durations of its timed region after the region ends, without
 * synchronising inside it. */
int tetra_demod_kernel_ms_history(tetra_demod_t* h, int n, float* ms);

/* Device self-test of the primitives the bit-exact contract rests on: in[0..63] -> sqrt, in[64..127] ->
 * sin/cos; out[0..63] = row_shr:1 DPP of lane ids (old = 100+lane), out[64..127] = row_shl:1 (old = 200+lane),
 * out[128..191] = sqrt, out[192..255] = sin, out[256..319] = cos.  Used by the GPU tests. */
int tetra_demod_debug_selftest(tetra_demod_t* h, const float* in128, float* out320);

/* Device self-test of the matrix pipe's f32 arithmetic: d[M][M] = a[M][k] . b[k][M] (row-major, M = shape = 16 or 32, k a
 * multiple of 4, <= 4096) accumulated over ascending k from +0 by chained v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32.
 * The GPU tests compare it bit for bit with the fmaf chain the arithmetic contract prescribes for every FIR sum. */
int tetra_demod_debug_mfma_selftest(tetra_demod_t* h, int shape, int k, const float* a, const float* b, float* d);

const char* tetra_demod_strerror(int status);
/* hipError_t of the last failing HIP call on this handle (0 if none). */
int tetra_demod_last_hip_error(tetra_demod_t* h);
int tetra_demod_abi_version(void);
/* 64 hex digits: sha256 of the sources and compile flags this library was built from (sdrpp-tetra-demodulator_amd/build.py
 * source_hash()).  The Python binding, the tests and smoke() refuse a library whose id is not the tree's: a stale .so is never
 * tested or benchmarked as if it were current. */
const char* tetra_demod_build_id(void);
/* Shader clock (kHz) and compute-unit count of a device, for callers that price a launch in clocks (either may be NULL). */
int tetra_demod_device_info(int device, int* clock_khz, int* compute_units);

#ifdef __cplusplus
}
#endif
#endif /* TETRA_DEMOD_H */
