/*
 * tetra_chan.h -- C ABI of the polyphase channeliser front-end (BASELINE.json config 5).
 *
 * The reference plugin has no channeliser: every instance asks SDR++ core for its own VFO/DDC
 * (src/main.cpp:75, sigpath::vfoManager.createVFO(name, ..., VFO_BANDWIDTH, VFO_SAMPLERATE, ...)) and demodulates
 * that one 36 ksps stream.  For a wideband capture carrying hundreds of TETRA carriers this front-end replaces the
 * N VFOs by one analysis filter bank on the GPU and hands the demodulator (include/tetra_demod.h,
 * TETRA_LAYOUT_TIME_MAJOR) one frame per output instant:
 *
 *   out[m][k] = sum_{l<L} h[l] * x[n_m - l] * exp(-j 2 pi k (n_m - l) / M),   n_m = (m+1)*D - 1,  k = 0..M-1
 *
 * M channels at spacing Fs/M (channel k centred at k*Fs/M; k > M/2 are the negative frequencies), prototype low-pass
 * h of L = P*M taps, decimation D (output rate Fs/D per channel; D = M/2 = 2x oversampled is the intended use:
 * 20 MHz / 800 channels = 25 kHz spacing, 50 ksps per channel, demodulator run with samplerate 50000).
 * Floating point throughout (float32 DFT): results are held to a tolerance against the double-precision definition
 * (oracle/chan_oracle.c), see tests/test_chan.py.  Same conventions as tetra_demod.h: extern "C", int status
 * (TETRA_OK / TETRA_ERR_*), no exceptions, one thread per handle, GPU only.
 */
#ifndef TETRA_CHAN_H
#define TETRA_CHAN_H

#include <stdint.h>

#include "tetra_demod.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tetra_chan_config {
    int32_t n_channels;        /* M: must factor as N1*N2 with N1, N2 <= 64 (e.g. 800 = 25*32, 32 = 4*8) */
    int32_t taps_per_channel;  /* P: prototype length L = P*M (1..32) */
    int32_t decimation;        /* D >= 1 */
    int32_t max_in;            /* largest n_in of one process call */
    int32_t device;            /* HIP device ordinal, -1 = current */
    int32_t reserved;          /* flags: TETRA_CHAN_FLAG_* (0 = default); any other bit: TETRA_ERR_ARG */
    double cutoff_rel;         /* prototype cutoff relative to half the channel spacing (1.0 = Fs/(2M)); default 1.2 */
    const float* prototype;    /* optional caller-supplied prototype [P*M]; NULL = Kaiser(beta 9)-windowed sinc */
} tetra_chan_config_t;

/* tetra_chan_config_t.reserved */
enum {
    TETRA_CHAN_FLAG_VALU_DFT = 1,  /* keep the direct-sum DFT kernel even where a faster form exists; same results within the
                                     documented float32 tolerance.  For A/B measurements and tests. */
    TETRA_CHAN_FLAG_MATRIX_DFT = 2 /* M = 800: the two DFT stages as chained v_mfma_f32_16x16x4_f32 (25 x 32 matrix products: round 4's
                                     form, any decimation) even at D = M / 2, where the default is the 32 x 5 x 5 mixed-radix FFT in
                                     registers / LDS (a tenth of the flops: the kernel is then bound by its HBM traffic).  A/B, tests. */
};

typedef struct tetra_chan tetra_chan_t;

int tetra_chan_default_config(tetra_chan_config_t* cfg);   /* M 800, P 8, D 400, cutoff 1.2 */
int tetra_chan_create(const tetra_chan_config_t* cfg, tetra_chan_t** out);
int tetra_chan_destroy(tetra_chan_t* h);
/* Frames the next process call with n_in samples will emit (depends on the carried sub-frame phase). */
int tetra_chan_frames_for(tetra_chan_t* h, int n_in);
/* x: n_in wideband complex64 samples (device pointer, 8-byte aligned: TETRA_ERR_ALIGN otherwise, also for out); out: [frames][M] complex64 (device pointer, capacity >=
 * tetra_chan_frames_for(n_in) frames); *n_frames receives the frame count.  Enqueued on hip_stream, no sync: d_x is read by the
 * work enqueued here (at M = 800, D = M / 2 the kernel reads it IN PLACE -- no staging copy, every sample crosses HBM once -- and
 * the last L - 1 samples are copied into the handle's delay line behind it), so it must stay untouched until that work has run.
 * The filter history and the sub-frame phase are carried across calls (results independent of the chunking). */
int tetra_chan_process_device(tetra_chan_t* h, const float* d_x, int n_in, float* d_out, int* n_frames, void* hip_stream);
/* The same for what an SDR's DMA delivers (round 6; the plugin's VFO reads the device's native stream, src/main.cpp:75): interleaved
 * int16 / int8 I, Q pairs, sample value = integer / 32768 (/ 128) -- exact in binary32, so the result equals tetra_chan_process_device
 * on the converted samples bit for bit.  At M = 800, D = M / 2 the kernel converts in its fold's loads: the capture is read in
 * place at 4 (2) bytes per sample instead of 8 and no float copy of it ever exists; the other kernels convert into their staging
 * buffer.  d_x: 4-byte (2-byte) aligned.  Calls of different formats may be mixed on one handle (the delay line is complex64). */
int tetra_chan_process_device_cs16(tetra_chan_t* h, const int16_t* d_x, int n_in, float* d_out, int* n_frames, void* hip_stream);
int tetra_chan_process_device_cs8(tetra_chan_t* h, const int8_t* d_x, int n_in, float* d_out, int* n_frames, void* hip_stream);
/* Host-pointer variant: copies in, runs, copies out, synchronises. */
int tetra_chan_process(tetra_chan_t* h, const float* x, int n_in, float* out, int* n_frames);
int tetra_chan_reset(tetra_chan_t* h);
/* Copy of the prototype filter [P*M]. */
int tetra_chan_get_prototype(tetra_chan_t* h, float* proto);
/* GPU time (ms) of the channeliser kernel of the most recent process call (HIP events on its stream). */
int tetra_chan_last_kernel_ms(tetra_chan_t* h, float* ms);

/* ---------------------------------------------------------------------------------------------------------------------------
 * Rational resampler I / DN on time-major frames (round 6).  The reference instance runs at VFO_SAMPLERATE 36000 with 2 samples per
 * symbol (src/main.cpp:35,75,84): SDR++'s VFO hands every plugin instance a 36 ksps stream.  The 2 x oversampled bank above emits
 * 50 ksps per channel, so config 5's 800 demodulator instances get their frames through an 18 / 25 polyphase resampler (SURVEY.md
 * section 8(f) #1: "PFB at 50 ksps + 18/25 rational resampler per channel") and are then created with the DEFAULT demodulator
 * configuration -- the plugin's own operating point.
 *
 *   y[m][c] = sum_{j < T} h[r_m + I j] * x[q_m - j][c],     I q_m + r_m = DN m,  0 <= r_m < I
 *
 * = zero-stuff by I, low-pass with the real prototype h (I*T taps, DC gain I), keep every DN-th sample; the same filter for every
 * channel.  Output m exists as soon as frame q_m = floor(DN m / I) has arrived: a call that brings the stream to n frames leaves
 * ceil(I n / DN) outputs emitted in total.  The T - 1 newest frames and the output position are carried across calls (results are
 * independent of the chunking).  float32 arithmetic, held to a tolerance against the double-precision definition
 * (oracle/chan_oracle.c: resamp_oracle_process), see tests/test_resamp.py.  18 / 25 with 8 / 12 / 16 / 24 taps per phase (and
 * 2 / 3, 3 / 2, 1 / 2 with 8) run a kernel specialised at compile time; every other ratio or length a generic one.
 * --------------------------------------------------------------------------------------------------------------------------- */
typedef struct tetra_resamp_config {
    int32_t n_channels;        /* C: complex channels per frame (row = C complex64) */
    int32_t interp;            /* I  >= 1 (18) */
    int32_t decim;             /* DN >= 1 (25) */
    int32_t taps_per_phase;    /* T: prototype length I*T (2..64) */
    int32_t max_in;            /* largest n_in (frames) of one process call */
    int32_t device;            /* HIP device ordinal, -1 = current */
    int32_t flags;             /* TETRA_RESAMP_FLAG_* */
    int32_t reserved;          /* must be 0 */
    double cutoff_rel;         /* prototype cutoff relative to the narrower Nyquist band min(in, out) / 2; default 1.0 */
    double kaiser_beta;        /* Kaiser window parameter of the designed prototype; default 6.0 */
    const float* prototype;    /* optional caller-supplied prototype [I*T] (replaces the designed one); NULL = Kaiser-windowed sinc */
} tetra_resamp_config_t;

enum {
    TETRA_RESAMP_FLAG_GENERIC = 1,      /* keep the generic (run-time ratio) kernel where a specialised one exists; A/B and tests */
    TETRA_RESAMP_FLAG_NARROW_UNITS = 2  /* one channel (8 bytes) per lane even where a row divides into 16-byte units (the default for
                                          an even channel count); same results; A/B and tests */
};

typedef struct tetra_resamp tetra_resamp_t;

int tetra_resamp_default_config(tetra_resamp_config_t* cfg);   /* 800 channels, 18 / 25, 16 taps per phase, cutoff 1.0, beta 6 */
int tetra_resamp_create(const tetra_resamp_config_t* cfg, tetra_resamp_t** out);
int tetra_resamp_destroy(tetra_resamp_t* h);
/* Frames the next process call with n_in input frames will emit. */
int tetra_resamp_frames_for(tetra_resamp_t* h, int n_in);
/* d_in: [n_in][C] complex64 frames (device pointer, 16-byte aligned, read IN PLACE by the work enqueued here: it must stay untouched
 * until that work has run); d_out: [>= tetra_resamp_frames_for(n_in)][C] complex64 (device pointer, 16-byte aligned, must not overlap
 * d_in); *n_out receives the frame count.  Enqueued on hip_stream, no sync. */
int tetra_resamp_process_device(tetra_resamp_t* h, const float* d_in, int n_in, float* d_out, int* n_out, void* hip_stream);
/* Host-pointer variant: copies in, runs, copies out, synchronises. */
int tetra_resamp_process(tetra_resamp_t* h, const float* in, int n_in, float* out, int* n_out);
int tetra_resamp_reset(tetra_resamp_t* h);
/* Copy of the prototype [I*T]. */
int tetra_resamp_get_prototype(tetra_resamp_t* h, float* proto);
/* GPU time (ms) of the resampler kernel of the most recent process call (HIP events on its stream). */
int tetra_resamp_last_kernel_ms(tetra_resamp_t* h, float* ms);

#ifdef __cplusplus
}
#endif
#endif
