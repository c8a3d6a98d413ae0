#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s demodulated to bits, batched TETRA channels, on N MI355X.

A "step" is one pass of the hot path (tetra_demod_process_device: AGC -> FLL -> RRC -> timing
recovery -> Costas -> slicer -> differential decoder -> bit unpacker) over one batch of synthetic
input that is already resident in HBM: BASELINE.json configs[2], 4096 channels x 36000 complex64
samples (1 s @ 36 ksps) per GPU, 65-tap RRC, loop state carried from step to step.  With N > 1
(launched by torch.distributed.run, one process per GPU) every rank demodulates its own 4096-channel
range -- channels are independent, so there is no data-path collective; RCCL is only used for the
barrier and the max-over-ranks of the elapsed time ("weak" scaling).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel (k_fused, the whole chain in one launch):
                algorithmic bytes (9 B per input sample: 8 B IQ read + 1 B bit written, SURVEY.md section
                8(d)) / its mean launch duration from HIP events recorded on the launch stream inside the
                timed region, against 8 TB/s HBM peak;
  cpu_baseline  the CPU oracle (oracle/, "port") timed on this host's cores on a bounded sample of the
                same workload.  The oracle is only the baseline/checker here, never the measured path.
  time_major    the same workload handed over as iq[n][c] (TETRA_LAYOUT_TIME_MAJOR, what a channeliser emits), A/B beside the
                channel-major headline (informational);
  host_path_*   PCIe-inclusive rates of the host entry points (informational);
  large_batch   the same call with twice the channels on this GPU (32-channel workgroup shape; informational);
  config5       BASELINE config 5 (20 MHz wideband -> channeliser -> 18/25 resampler -> 800 channels at the plugin's 36 ksps -> bits;
                informational; also --config5 alone);
  chain         the receive chain behind one handle (include/tetra_rx.h) on coded downlinks: steady-state ms per second of 4096
                channels on two streams / one stream, stage times with rooflines, CRC-good and exact-block counters (informational;
                also --chain-only).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHANNELS_PER_GPU = 4096
SAMPLES = 36000
CHECK_CHANNELS = 256               # channels of every rank's bank the known-answer check demodulates back to their transmitted bits
ALGO_BYTES_PER_SAMPLE = 9.0
HBM_PEAK_GBS = 8000.0
# Useful arithmetic of the chain per input sample at 2 samples/symbol and 65 taps, an fma counted as 2 (DESIGN.md section 4
# itemises it): AGC 9 + rotation (sine/cosine polynomial + complex multiply) 36 + two band-edge FIRs as four 65-tap real sums
# 520 + error/loop 12 + RRC 260 + per symbol (3 x 8-tap complex interpolator sums 96, timing loop 12, two rotations 72,
# Costas loop + slicer 16) / 2 = 98  ->  935 flop per sample.
FLOP_PER_SAMPLE = 935.0
VALU_PEAK_TFLOPS = 157.3          # MI355X FP32 vector peak (MI355X_MICROARCH.md)
RAMP_STEPS = 8                    # untimed passes that bring the shader clock up before warm-up (see main)


def make_input(torch, pkg, device, n_channels, n_samples, seed):
    """Synthetic batch in HBM, every channel from its OWN seed (seed + channel index; BASELINE.md's generator): bits and channel
    parameters (carrier offset, timing offset, amplitude, phase) are a hash of (channel seed, position), shaped and rotated on the
    GPU in float64 (synth_gpu.gen_bank = synth.gen_channel's signal model), AWGN at Es/N0 25 dB.  Returns (iq [C][N] complex64 on
    device, seeds int64 [C]); synth.hash_bits(seeds[c], ..) regenerates channel c's transmitted bits on the CPU."""
    return pkg.synth_gpu.gen_bank(torch, device, n_channels, n_samples, seed)


KERNEL_SOURCES = ("kernel_fused.hpp", "demod_core.hpp", "fll_asm.inc", "fll4_asm.inc", "fll16_asm.inc", "fll16l_asm.inc", "fll8l_asm.inc")      # what k_fused is compiled from (+ the flags below)


def kernel_source_hash():
    """sha256 over the sources the dominant kernel is built from (and its arithmetic-relevant compile flags): a counter
    measurement is only as current as this.  Host-side files (the C ABI around the launch) do not enter."""
    import hashlib
    csrc = os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "csrc")
    hsh = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(csrc, name), "rb") as f:
            hsh.update(name.encode() + b"\0" + f.read())
    hsh.update(b"-O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950")
    return hsh.hexdigest()


def pmc_traffic(pipeline, channels, samples):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (profiles/run_rocprof.sh: separate --pmc
    FETCH_SIZE / WRITE_SIZE runs of this same command, corrected as MI355X_MICROARCH.md prescribes).  Counters cannot be
    collected from inside this process, so the committed measurement for the same workload is reported -- but ONLY if it
    was taken on the kernel sources of this build (run_rocprof.sh stores kernel_source_hash() in the file); a measurement
    of other sources is refused: (None, None, reason)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_%s_%dx%d.json" % (pipeline, channels, samples))
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel_source_sha256") != kernel_source_hash():
            return None, None, "stale: %s was measured on other kernel sources (re-run profiles/run_rocprof.sh)" % os.path.basename(path)
        return float(d["traffic_bytes_per_launch"]), d.get("issue"), "bytes per launch (rocprofv3 PMC, %s)" % os.path.basename(path)
    except (OSError, KeyError, ValueError):
        return None, None, "no counter measurement for this workload under profiles/"


def cpu_baseline(synth, n_samples, budget_s=8.0):
    """Time the CPU legs on a bounded sample of the same workload: batches of 4 channels per thread x n_samples, streamed
    back to back (state carried) for ~budget_s seconds each, OpenMP over channels on all host threads.
      port       oracle/tetra_oracle.c -- the bit-exact checker (-O2, one serial fmaf chain per FIR: built for parity)
      port-fast  oracle/tetra_fast.c   -- the same chain built for speed (-O3 -march=native -ffast-math, FIR sums over
                 independent accumulators, blocked RRC); its bits after lock equal the oracle's (tests/test_oracle.py)
    `threads` = OpenMP threads used = `cores` = the CPUs this container may use (scheduler affinity cut by the cgroup CPU
    quota; `host_logical_cpus` and `cgroup_cpu_quota` say what the machine has and what the quota is)."""
    from oracle import binding as ob
    usable, quota = ob.usable_cpus()          # affinity cut by the cgroup CPU quota: more threads than that only throttle
    threads = max(1, min(ob.max_threads(), usable))
    cores = min(ob.physical_cores(), usable)
    n_ch = min(4096, max(64, 16 * threads))
    base, _, _ = synth.gen_batch(min(n_ch, 32), n_samples, base_seed=999)
    iq = np.ascontiguousarray(np.tile(base, ((n_ch + base.shape[0] - 1) // base.shape[0], 1))[:n_ch])

    def leg(kind, call):
        states = call(None)                                            # warm-up (also locks the loops)
        reps, t0 = 0, time.perf_counter()
        while True:
            states = call(states)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget_s or reps >= 4000:
                break
        return dict(value=round(reps * n_ch * n_samples / el / 1e6, 3), unit="Msamples/s", cores=cores, threads=threads,
                    kind=kind, per_thread_msamples_s=round(reps * n_ch * n_samples / el / 1e6 / threads, 3),
                    host_logical_cpus=os.cpu_count(), cgroup_cpu_quota=quota,
                    sample="%d x (%d channels x %d samples), %d OpenMP threads on the %d cores this container may use "
                           "(host: %d logical CPUs), %.1f s" % (reps, n_ch, n_samples, threads, cores, os.cpu_count() or 0, el))

    def one_thread(call):
        """The 1-thread figure SURVEY.md 8(d) asks for beside the all-cores one: 4 channels, ~2 s."""
        small = np.ascontiguousarray(iq[:4])
        st = call(small, None)
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 2.0:
            st = call(small, st)
            reps += 1
        return round(reps * 4 * n_samples / (time.perf_counter() - t0) / 1e6, 3)

    port = leg("port", lambda st: ob.process_batch(iq, threads=threads, states=st)[3])
    port["single_thread_msamples_s"] = one_thread(lambda x, st: ob.process_batch(x, threads=1, states=st)[3])
    try:
        fast = leg("port-fast", lambda st: ob.fast_process_batch(iq, threads=threads, states=st)[2])
        fast["single_thread_msamples_s"] = one_thread(lambda x, st: ob.fast_process_batch(x, threads=1, states=st)[2])
    except Exception as e:          # the baseline must never take the bench line down
        fast = dict(error=str(e)[:200], kind="port-fast")
    return port, fast


def wideband_tetra(torch, synth, device, M, n_in, carriers, seed=5):
    """A wideband capture at Fs = M x 25 kHz holding pi/4-DQPSK TETRA carriers {channel k: seed} (18 ksymbols/s, RRC 0.35,
    i.e. Fs / 18000 samples per symbol) over a noise floor: synth.modulate's formula evaluated on the GPU in float64 (the
    numpy form takes seconds per carrier at 5e6 samples).  Returns (x complex64 [n_in] on device, {k: tx bits})."""
    import math
    fs = M * 25000.0
    sps = fs / 18000.0
    beta = 0.35
    n = torch.arange(n_in, device=device, dtype=torch.float64)
    x = torch.zeros(n_in, device=device, dtype=torch.complex128)
    tx = {}

    def rrc(t):          # synth.rrc_pulse (unit-energy root-raised cosine, t in symbol periods)
        z = t.abs() < 1e-9
        sg = (t.abs() - 1.0 / (4.0 * beta)).abs() < 1e-9
        ts = torch.where(z | sg, torch.full_like(t, 0.123), t)
        v = (torch.sin(math.pi * ts * (1 - beta)) + 4 * beta * ts * torch.cos(math.pi * ts * (1 + beta))) / (
            math.pi * ts * (1 - (4 * beta * ts) ** 2))
        v = torch.where(z, torch.full_like(t, 1 - beta + 4 * beta / math.pi), v)
        return torch.where(sg, torch.full_like(t, (beta / math.sqrt(2)) * ((1 + 2 / math.pi) * math.sin(math.pi / (4 * beta)) +
                                                                       (1 - 2 / math.pi) * math.cos(math.pi / (4 * beta)))), v)

    for k, sd in carriers.items():
        bits = np.random.default_rng(sd).integers(0, 2, synth.needed_bits(n_in, sps), dtype=np.uint8)
        syms = torch.from_numpy(synth.bits_to_symbols(bits)).to(device)
        K = syms.shape[0]
        t = (n + 0.37 * sps) / sps - synth.SPAN
        k0 = torch.floor(t).to(torch.int64)
        s = torch.zeros(n_in, device=device, dtype=torch.complex128)
        for j in range(-synth.SPAN, synth.SPAN + 1):
            kk = k0 + j
            valid = (kk >= 0) & (kk < K)
            s += torch.where(valid, syms[kk.clamp(0, K - 1)], torch.zeros((), device=device, dtype=torch.complex128)) * rrc(t - kk)
        kc = k if k <= M // 2 else k - M
        x += 0.3 * s * torch.polar(torch.ones_like(n), 2.0 * math.pi * kc / M * n)
        tx[k] = bits
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x += 0.003 * torch.view_as_complex(torch.randn((n_in, 2), device=device, generator=g, dtype=torch.float64))
    return x.to(torch.complex64).contiguous(), tx


def wideband_config5(args, torch, pkg, device, local_rank):
    """BASELINE config 5 (informational, NOT the metric line): 20 MHz wideband capture -> 800 x 25 kHz channels (2x oversampled
    bank, 50 ksps each) -> 18 / 25 rational resampler -> 36 ksps -> demodulator created with the DEFAULT configuration = the plugin's
    own operating point (VFO_SAMPLERATE 36000, 2 samples per symbol: /root/reference/src/main.cpp:35,75,84), time-major layout -> bits.
    One step = 0.25 s of capture.  `--config5-rate 50000` keeps round 5's route (no resampler, the demodulator run at 50 ksps).
    The capture holds 16 TETRA carriers spread over the band (one at a negative frequency index, one at each band edge region) over a
    noise floor; the known-answer check demodulates them back to their transmitted bits."""
    M, P, D = 800, 8, 400
    n_in = 5000000
    frames = n_in // D
    rate = int(getattr(args, "config5_rate", 36000))
    if rate not in (36000, 50000):
        raise SystemExit("--config5-rate: 36000 (the plugin's rate, through the 18/25 resampler) or 50000 (the bank's own rate)")
    resample = rate == 36000
    carriers = {k: 500 + i for i, k in enumerate((3, 57, 101, 150, 199, 250, 313, 377, 423, 480, 531, 590, 644, 700, 751, 797))}
    x, tx = wideband_tetra(torch, pkg.synth, device, M, n_in, carriers)
    # what the SDR's DMA delivers: interleaved int16 (int8) I / Q pairs, read by the channeliser in place (tetra_chan_process_device_cs16 /
    # _cs8); the capture is scaled to a quarter of full scale at its largest component.  --config5-input complex64 keeps the float capture.
    in_fmt = getattr(args, "config5_input", "cs16")
    in_bytes = {"complex64": 8.0, "cs16": 4.0, "cs8": 2.0}[in_fmt]
    if in_fmt != "complex64":
        full = 32768.0 if in_fmt == "cs16" else 128.0
        peak = float(torch.view_as_real(x).abs().max())
        x = torch.clamp(torch.round(torch.view_as_real(x) * (0.25 * full / peak)), -full, full - 1).to(torch.int16 if in_fmt == "cs16" else torch.int8).contiguous()
    ch = pkg.Channeliser(M, P, D, max_in=n_in, device=local_rank)
    rs = pkg.Resampler(M, 18, 25, 16, max_in=frames, device=local_rank) if resample else None
    n_dem = frames * 18 // 25 if resample else frames          # 9000 frames at 36 ksps
    if resample:
        dem = pkg.Demodulator(M, n_dem + 1, layout=pkg.binding.LAYOUT_TIME_MAJOR, device=local_rank)            # default configuration
    else:
        dem = pkg.Demodulator(M, frames, layout=pkg.binding.LAYOUT_TIME_MAJOR, device=local_rank, samplerate=50000.0)
    out = torch.zeros((frames, M), dtype=torch.complex64, device=device)
    out36 = [torch.zeros((n_dem + 1, M), dtype=torch.complex64, device=device) for _ in range(2)] if resample else None
    stride = dem.bits_stride(n_dem + 1)
    bits = torch.zeros((M, stride), dtype=torch.uint8, device=device)
    nbits = torch.zeros(M, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream(device)

    def front(dst50, dst36, s):
        """channeliser (+ resampler) of one block on stream s; returns (frame buffer for the demodulator, frames in it)"""
        nf = ch.process_device(x, n_in, dst50, s)
        if not resample:
            return dst50, nf
        return dst36, rs.process_device(dst50, nf, dst36, s)

    def step():
        buf, n = front(out, out36[0] if resample else None, stream)
        dem.process_device(buf, n, bits, stride, nbits, None, stream)

    for _ in range(args.warmup + 4 * RAMP_STEPS):      # a step is ~1 ms: this many bring the shader clock to its steady value
        step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    el = time.perf_counter() - t0
    k1 = dem.kernel_ms_history(1)
    ch_ms = ch.last_kernel_ms()
    rs_ms = rs.last_kernel_ms() if resample else None
    # steady-state streaming: the front-end works on block k+1 (its own stream, the other frame buffers) while the
    # demodulator -- 800 channels = 200 of the 256 CUs, latency-bound -- is on block k
    s_ch, s_dem = torch.cuda.Stream(device), torch.cuda.Stream(device)
    outs = [out, torch.zeros_like(out)]
    ev_ch = [torch.cuda.Event() for _ in range(2)]
    ev_dem = [torch.cuda.Event() for _ in range(2)]

    def pipelined(k):
        b = k & 1
        if k >= 2:
            s_ch.wait_event(ev_dem[b])
        buf, n = front(outs[b], out36[b] if resample else None, s_ch)
        ev_ch[b].record(s_ch)
        s_dem.wait_event(ev_ch[b])
        dem.process_device(buf, n, bits, stride, nbits, None, s_dem)
        ev_dem[b].record(s_dem)

    torch.cuda.synchronize(device)
    for k in range(2 + 4 * RAMP_STEPS):
        pipelined(k)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(2 + 4 * RAMP_STEPS, 2 + 4 * RAMP_STEPS + args.steps):
        pipelined(k)
    torch.cuda.synchronize(device)
    el2 = time.perf_counter() - t0
    # known answer: fresh loops, two passes over the capture (0.5 s of signal per carrier; the block repeats, so the loops see one
    # phase jump and stay in lock), then the second pass's second half against the transmitted bits of every carrier
    check = None
    if not args.no_check:
        dem.reset()
        step()
        step()
        torch.cuda.synchronize(device)
        hb, hn = bits.cpu().numpy(), nbits.cpu().numpy()
        errs = ncmp = 0
        for k, b in tx.items():
            lag, e, n = pkg.synth.align_and_count_errors(hb[k][: hn[k]], b, skip=hn[k] // 2, max_lag=600)
            errs += e
            ncmp += n
        idle = [k for k in range(M) if k not in tx and all(abs(k - c) > 1 for c in tx)]
        check = dict(carriers=len(tx), bits_compared_second_half=int(ncmp), bit_errors=int(errs), idle_channels=len(idle))
        if ncmp < 2000 * len(tx) or errs > 1e-3 * ncmp:
            raise SystemExit("config 5 known-answer check failed: %d bit errors in %d bits of %d carriers" % (errs, ncmp, len(tx)))
    # Rooflines of the leg's kernels.  Channeliser: algorithmic bytes = the capture read once + the frames written once
    # (8 / 4 / 2 B per wideband sample in for complex64 / cs16 / cs8, 8 B per channel-sample out) -- HBM is what bounds the kernel since round 5; algorithmic flops =
    # the weighted overlap-add (L = P M taps, 4 flop each) + an M-point complex FFT (5 M log2 M), per frame.  Resampler: every input
    # frame read once, every output frame written once (8 B per channel-sample each way); 4 T flop per complex output.  Demodulator:
    # 9 B per channel-sample.
    def hbm(nbytes, ms, **extra):
        d = {"achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": nbytes}
        d.update(extra)
        return d

    ch_bytes = in_bytes * n_in + 8.0 * frames * M
    ch_flop = frames * (4.0 * P * M + 5.0 * M * math.log2(M))
    dm_bytes = ALGO_BYTES_PER_SAMPLE * n_dem * M
    roof = {"channeliser": {"kernel": "k_channelise_fft", "kernel_ms": round(ch_ms, 4), "bound": "hbm",
                            "hbm": hbm(ch_bytes, ch_ms, achievable_gbs=6290.0, frac_of_achievable=round(ch_bytes / (ch_ms * 1e-3) / 1e9 / 6290.0, 4)),
                            "fp32": {"achieved": round(ch_flop / (ch_ms * 1e-3) / 1e12, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(ch_flop / (ch_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4), "algorithmic_flop": ch_flop,
                                     "note": "32 x 5 x 5 mixed-radix FFT in registers / LDS (round 4: 25 x 32 matrix products, 9.5x the flops)"}},
            "demodulator": {"kernel": "k_fused<.., 4>", "kernel_ms": round(float(k1[0]), 4), "hbm": hbm(dm_bytes, float(k1[0])),
                            "note": "800 channels = 200 four-channel workgroups: 200 of the 256 CUs, each at the per-sample recurrence's pace"}}
    if resample:
        rs_bytes = 8.0 * frames * M + 8.0 * n_dem * M
        rs_flop = 4.0 * 16 * n_dem * M
        roof["resampler"] = {"kernel": "k_resample<18, 25, 16, 4>", "kernel_ms": round(rs_ms, 4), "bound": "hbm",
                             "hbm": hbm(rs_bytes, rs_ms, achievable_gbs=6290.0, frac_of_achievable=round(rs_bytes / (rs_ms * 1e-3) / 1e9 / 6290.0, 4)),
                             "fp32": {"achieved": round(rs_flop / (rs_ms * 1e-3) / 1e12, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": round(rs_flop / (rs_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4), "algorithmic_flop": rs_flop}}
    workload = ("5e6 %s samples @ 20 MHz (16 TETRA carriers over a noise floor) -> 800 ch x 12500 frames @ 50 ksps -> 18/25 resampler -> "
                "800 ch x 9000 frames @ 36 ksps (the plugin's rate, default demodulator parameters) -> bits") % in_fmt if resample else \
               "5e6 %s samples @ 20 MHz (16 TETRA carriers over a noise floor) -> 800 ch x 12500 frames @ 50 ksps -> bits (demodulator at 50 ksps)" % in_fmt
    res = {"metric": "wideband IQ Msamples/s channelised and demodulated to bits (BASELINE config 5)",
           "value": round(args.steps * n_in / el / 1e6, 2), "unit": "Msamples/s (20 MHz capture)",
           "ms_per_step": round(el / args.steps * 1e3, 3), "realtime_factor": round(args.steps * n_in / el / 20e6, 1),
           "channeliser_kernel_ms": round(ch_ms, 3), "resampler_kernel_ms": None if rs_ms is None else round(rs_ms, 4),
           "demod_kernel_ms": round(float(k1[0]), 3),
           "two_streams_ms_per_step": round(el2 / args.steps * 1e3, 3),
           "two_streams_realtime_factor": round(args.steps * n_in / el2 / 20e6, 1),
           "check": check, "roofline": roof,
           "config": {"workload": workload, "channels": M, "taps_per_channel": P, "decimation": D, "channel_rate_sps": rate,
                      "resampler": {"interp": 18, "decim": 25, "taps_per_phase": 16} if resample else None,
                      "input_format": {"complex64": "complex64", "cs16": "interleaved int16 I/Q (4 B per sample), read in place by the channeliser's fold",
                                       "cs8": "interleaved int8 I/Q (2 B per sample), read in place by the channeliser's fold"}[in_fmt]}}
    del outs, out36
    ch.close()
    if rs is not None:
        rs.close()
    dem.close()
    del x, out, bits, nbits
    return res


CHAIN_DISTINCT = 64          # distinct coded downlinks (cells) of the chain leg, each used for C / 64 channels with its own rotation


def receive_chain(args, torch, pkg, device, local_rank, C=CHANNELS_PER_GPU, N=SAMPLES, seconds=4):
    """The receive chain behind the demodulator, through the ONE handle of include/tetra_rx.h (informational, NOT the metric line):
    IQ -> demodulator || burst synchroniser -> frame lists -> SB1 decoded straight from the frames -> SYNC-PDU tracker -> every other
    block kind decoded straight from the frames in one launch -> labelled type-1 blocks (reference: tetra_burst_sync_in -> tetra_burst_rx_cb -> tp_sap_udata_ind, phy/tetra_burst_sync.c:54-155,
    phy/tetra_burst.c:343-393, lower_mac/tetra_lower_mac.c:148-275).  Input: CODED continuous downlinks (synth.gen_downlink: SYNC
    bursts with SB1 + AACH + SB2, two-channel normal bursts with NDB 1 + 2, one-channel normal bursts with SCH/F; every block carries
    a CRC and its type-1 bits are known), CHAIN_DISTINCT cells x C / CHAIN_DISTINCT channels each with its own amplitude, carrier
    offset and phase; `seconds` consecutive blocks of N samples per channel resident in HBM, streamed round and round with the state
    carried.  Reports steady-state ms per block (= per second of C channels) with the tail overlapped on its own stream and on one
    stream, the stage times of the one-stream run with their rooflines, and the known-answer counters of the last block."""
    R, synth = pkg.rx_binding, pkg.synth
    n_slots = seconds * N // 510 + 2
    cells = [(200 + c, 3000 + 7 * c, (11 * c + 5) % 64) for c in range(CHAIN_DISTINCT)]
    down = [synth.gen_downlink(n_slots, 7000 + c, cell=cells[c]) for c in range(CHAIN_DISTINCT)]
    nb = synth.needed_bits(seconds * N)
    bits = np.zeros((CHAIN_DISTINCT, nb), np.uint8)
    for c in range(CHAIN_DISTINCT):
        bits[c, : min(nb, down[c][0].size)] = down[c][0][:nb]
    seeds = torch.arange(CHAIN_DISTINCT, dtype=torch.int64, device=device) + 31000
    prm = pkg.synth_gpu.hash_params(torch, device, seeds)
    one = torch.ones(CHAIN_DISTINCT, dtype=torch.float64, device=device)
    base = pkg.synth_gpu.modulate_batch(torch, device, torch.from_numpy(bits).to(device), seconds * N, prm["tau"], 0 * one, one, 0 * one,
                                        esn0_db=25.0, noise_seed=31000)          # unit amplitude, no rotation: applied per channel below
    g = torch.Generator(device="cpu")
    g.manual_seed(31001)
    amp = torch.empty(C).uniform_(0.05, 1.0, generator=g).to(device).double()
    dw = torch.empty(C).uniform_(-0.05, 0.05, generator=g).to(device).double()
    ph = torch.empty(C).uniform_(-3.14159, 3.14159, generator=g).to(device).double()
    idx = torch.arange(C, device=device) % CHAIN_DISTINCT
    n = torch.arange(seconds * N, device=device, dtype=torch.float64)
    d_iq = [torch.empty((C, N), dtype=torch.complex64, device=device) for _ in range(seconds)]
    for c0 in range(0, C, 256):
        c1 = min(C, c0 + 256)
        rot = torch.polar(amp[c0:c1, None].expand(-1, seconds * N).contiguous(), dw[c0:c1, None] * n[None, :] + ph[c0:c1, None]).to(torch.complex64)
        x = base[idx[c0:c1]] * rot
        for k in range(seconds):
            d_iq[k][c0:c1] = x[:, k * N:(k + 1) * N]
    del base, x, rot
    stream = torch.cuda.current_stream(device)
    res = {"channels": C, "samples_per_channel": N, "blocks_resident": seconds,
           "workload": "%d coded downlinks (cells) x %d channels each with its own amplitude / carrier offset / phase; every slot a burst: "
                       "SYNC (SB1 + AACH + SB2) / NORM_2 (NDB 1 + 2 + AACH) / NORM_1 (SCH/F + AACH) x 2 per four slots; Es/N0 25 dB" % (CHAIN_DISTINCT, C // CHAIN_DISTINCT)}
    reps = 2 * seconds

    def run(rx, n_calls):
        for k in range(n_calls):
            rx.process_device(d_iq[k % seconds], N, stream)

    stage = None
    for name, flags in (("two_streams", 0), ("one_stream", R.FLAG_ONE_STREAM)):
        rx = pkg.RxChain(C, N, device=local_rank, flags=flags)
        run(rx, seconds + 2)                      # locks the loops and the synchronisers, fills the allocator pools, ramps the clock
        rx.wait()
        t0 = time.perf_counter()
        run(rx, reps)
        rx.wait()
        ms = (time.perf_counter() - t0) * 1e3 / reps
        res[name + "_ms_per_second"] = round(ms, 3)
        res[name + "_x_real_time"] = round(1000.0 / ms, 1)
        if flags:
            stage = rx.stage_ms()
        else:
            # known answer on the last block of the overlapped run: every decoded block of every channel has a good CRC, and on the
            # CHAIN_DISTINCT distinct streams the type-1 bits are those sent in the slot the block's TDMA time names
            names = {R.KIND_SB1: "sb1", R.KIND_BBK: "bbk", R.KIND_SB2: "sb2", R.KIND_NDB1: "ndb1", R.KIND_NDB2: "ndb2", R.KIND_SCH_F: "schf"}
            by_time = {}
            for sl in range(n_slots):
                tn, fn, mn = synth.tdma_time_of_slot(sl)
                by_time[tn | fn << 8 | mn << 16] = sl
            counters, rows_k = {}, {}
            bad = 0
            t_fetch = 0.0
            for k, nm in names.items():
                tf = time.perf_counter()
                blocks, t1 = rx.fetch(k)
                t_fetch += time.perf_counter() - tf
                rows_k[k] = len(blocks)
                good = int((blocks["crc_ok"] != 0).sum())
                sel = np.nonzero(blocks["channel"] < CHAIN_DISTINCT)[0]
                exact = 0
                sent = [{sl: v.tobytes() for sl, v in down[c][1][nm]} for c in range(CHAIN_DISTINCT)]
                for j in sel:
                    sl = by_time.get(int(blocks["tdma_time"][j]))
                    exact += int(sl is not None and sent[int(blocks["channel"][j])].get(sl) == t1[j].tobytes())
                counters[nm] = {"rows": len(blocks), "crc_good": good, "checked_rows": int(len(sel)), "type1_bits_and_tdma_slot_exact": exact}
                bad += (len(blocks) - good) + (len(sel) - exact)
            cellok = sum(1 for c, st in enumerate(rx.cells()) if (st.mcc, st.mnc, st.colour_code) == cells[c % CHAIN_DISTINCT])
            locked = sum(1 for st in rx.sync_states() if st[0] == 2)
            res["check"] = {"blocks": counters, "channels_locked": locked, "cells_read": cellok, "channels": C}
            if bad or cellok != C or locked != C:
                raise SystemExit("receive chain known-answer check failed: %s" % json.dumps(res["check"]))
            res["rows_per_kind"] = {names[k]: v for k, v in rows_k.items()}
            # informational, PCIe-inclusive: every block of the call fetched to host memory through tetra_rx_fetch (labels + type-1 bits, one
            # byte per bit as the reference's upper MAC takes them; pageable numpy buffers, one kind after the other, nothing overlapped)
            res["host_fetch_all_kinds_ms"] = round(t_fetch * 1e3, 2)
            res["host_fetch_bytes"] = int(sum(rows_k[k] * (24 + pkg.rx_binding.type1_bits(k)) for k in rows_k))
        rx.close()
    # Rooflines of the one-stream run's stages (round 6 chain: no byte rows between the stages).  Bytes: demodulator 9 B per sample;
    # synchroniser = the bit rows it scans + 64 B per packed frame + 8 B per frame slot; SB1 stage = the frame-list pass (frame types
    # read twice, 4 B per list entry written) + per SB1 row (frame 64, list entry 4, type-2 row 80 + crc 4 written, then read by the
    # tracker with the label 24 written) + the tracker's 12 B per frame slot; other kinds = per decoded row (frame 64, list entry 4,
    # code 4, times + bit number 12 read; type-2 row + crc + label out + 28 written).  The decoder's decision scratch (2 B per trellis
    # step and block, written once and read once) normally lives in L2 / MALL and is not counted.  The decoder is integer vector work:
    # 99 vector instructions per step PAIR in the forward recursion + 6.1 per step in the traceback (ISA of k_lmac_frames,
    # DESIGN.md 8.3) = 55.6 per trellis step per 64-block wave, against the chip's issue peak 256 CUs x 4 SIMDs x 2.4 GHz / 4 clocks
    # = 614 G wave-instructions / s.
    F = (4096 + pkg.binding.bits_stride(N)) // 510 + 2
    geo = {"sb1": (80, 84), "bbk": (32, 0), "sb2": (144, 148), "ndb1": (144, 148), "ndb2": (144, 148), "schf": (288, 292)}      # type-2 row bytes, trellis steps
    rows = res["rows_per_kind"]
    frames = sum(rows[k] for k in ("sb1", "ndb1", "schf"))          # every frame with a callback carries exactly one of these
    lists = rows["sb1"] + rows["ndb1"] + rows["schf"] + rows["bbk"]
    by = [9.0 * C * N, float(C * N) + 64.0 * frames + 8.0 * C * F,
          8.0 * C * F + 4.0 * lists + rows["sb1"] * (64.0 + 4.0 + 2 * 84.0 + 24.0) + 12.0 * C * F,
          sum(rows[k] * (64.0 + 4.0 + 4.0 + 12.0 + geo[k][0] + 28.0) for k in geo if k != "sb1")]
    steps = [0, 0, rows["sb1"] * geo["sb1"][1], sum(rows[k] * geo[k][1] for k in geo if k != "sb1")]
    stages = {}
    for i, nm in enumerate(("demodulator", "burst_sync", "frame_lists_sb1_decode_track", "other_kinds_decode_label")):
        d = {"ms": round(stage[i], 4), "algorithmic_bytes": round(by[i]), "GBps": round(by[i] / (stage[i] * 1e-3) / 1e9, 1),
             "frac_hbm_8TBps": round(by[i] / (stage[i] * 1e-3) / 8e12, 4)}
        if steps[i]:
            wi = 55.6 * steps[i] / 64.0
            d.update(trellis_steps=int(steps[i]), vector_instructions_per_step_and_wave=55.6,
                     frac_valu_issue_614G=round(wi / (stage[i] * 1e-3) / 614.4e9, 4), bound="valu-issue (integer add-compare-select)")
        stages[nm] = d
    res["stages_one_stream"] = stages
    res["tail_ms_one_stream"] = round(sum(stage[1:]), 4)
    del d_iq
    return res


def informational(leg, *a):
    """An informational leg of the default line (config 5, the receive chain) must not take the metric line down with it: a leg that
    fails -- its own known-answer check included -- is reported as {"error": ...} in its place (run alone, --config5 / --chain-only,
    the same failure ends the run with a non-zero status)."""
    try:
        return leg(*a)
    except (SystemExit, Exception) as e:      # noqa: BLE001 -- SystemExit is what the legs' own checks raise
        return {"error": "%s: %s" % (type(e).__name__, e)}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks of this same command under
    torch.distributed.run (one process per GPU; on a box with fewer GPUs than ranks they share -- functional runs only) and
    pass the line rank 0 prints through.  Under a launcher (WORLD_SIZE set) main() runs the rank itself."""
    import subprocess
    # --standalone: the launcher picks the rendezvous port itself (c10d store on port 0), so there is no window between "found a
    # free port" and "bound it" for another process to take it; --local-addr keeps the rendezvous on 127.0.0.1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--channels", type=int, default=CHANNELS_PER_GPU, help="channels per GPU")
    ap.add_argument("--samples", type=int, default=SAMPLES, help="samples per channel per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--config5", action="store_true",
                    help="run BASELINE config 5 instead (wideband -> channeliser -> 800-channel demod); informational")
    ap.add_argument("--config5-rate", type=int, default=36000,
                    help="per-channel rate the config 5 demodulator instances run at: 36000 (default: the plugin's VFO_SAMPLERATE, through "
                         "the 18/25 resampler) or 50000 (the 2x oversampled bank's own rate, round 5's route)")
    ap.add_argument("--config5-input", choices=["cs16", "cs8", "complex64"], default="cs16",
                    help="sample format of config 5's wideband capture (default: int16 I/Q pairs, what an SDR delivers)")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the PCIe-inclusive host-path legs (tetra_demod_process / tetra_demod_process_async on page-locked "
                         "buffers; informational fields host_path_*, never the metric value)")
    ap.add_argument("--host-path", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the informational BASELINE config 5 leg (wideband -> channeliser -> 800-channel demod, field config5)")
    ap.add_argument("--no-large-batch", action="store_true",
                    help="skip the informational leg with twice the channels per GPU (32-channel workgroup shape, field large_batch)")
    ap.add_argument("--chain", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-chain", action="store_true",
                    help="skip the informational receive-chain leg (include/tetra_rx.h: demodulator || synchroniser -> demultiplexer -> "
                         "lower-MAC decoder -> SYNC-PDU tracker on coded downlinks; field chain)")
    ap.add_argument("--chain-only", action="store_true", help="run the receive-chain leg alone and print its object (profiling)")
    ap.add_argument("--force-dist", action="store_true",
                    help="build the torch.distributed group even for ONE rank: runs the RCCL set-up, barrier and MAX reduction of the "
                         "N > 1 path on a one-GPU box (RCCL refuses several ranks on one device, so this is how its half is exercised there)")
    ap.add_argument("--no-time-major", action="store_true",
                    help="skip the informational leg that hands the same workload over time-major (iq[n][c], field time_major)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import tetra_amd
    pkg = tetra_amd.pkg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d is running as one of %d rank(s): launch it with --nproc-per-node %d, or without a "
                         "launcher (it starts its ranks itself)" % (args.gpus, world, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the demodulator has no CPU path")
    # More ranks than GPUs (a functional run of the N > 1 path on a smaller box): ranks are folded onto the GPUs there are, and
    # the line says so where no parser can miss it -- "functional_only": true, "value": null, "n_gpus" = the GPUs that really
    # ran ("n_ranks" = the ranks), the folded throughput only as "value_functional".
    gpus_physical = torch.cuda.device_count()
    folded = world > gpus_physical
    local_rank %= max(1, gpus_physical)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    backend_note = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise SystemExit("bench.py: WORLD_SIZE %d without MASTER_PORT -- start the ranks with torch.distributed.run (or let "
                                 "`python bench.py --gpus N` start them)" % world)
            os.environ["MASTER_PORT"] = "0"          # a one-rank group (--force-dist): the store binds a free port itself
        # RCCL carries only the barrier and the MAX of the elapsed time (no data-path collective).  One rank per GPU: RCCL
        # refuses ranks that share a device ("Duplicate GPU detected") -- a functional run of the N > 1 path on fewer GPUs
        # than ranks takes --backend gloo.  A communicator that cannot be built fails the run at once, loudly.
        if args.backend == "nccl" and world > torch.cuda.device_count():
            # more ranks than GPUs (a functional run of the N > 1 path on a smaller box): RCCL needs a GPU per rank, so the ranks
            # line up over gloo instead -- decided from the device count, the same on every rank, before any communicator exists
            args.backend = "gloo"
            backend_note = "gloo: %d ranks on %d GPU(s), RCCL needs one GPU per rank" % (world, torch.cuda.device_count())
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    if args.config5:
        print(json.dumps(wideband_config5(args, torch, pkg, device, local_rank)))
        return
    if args.chain_only:
        print(json.dumps(receive_chain(args, torch, pkg, device, local_rank)))
        return
    C, N = args.channels, args.samples
    # rank r demodulates global channels [r*C, (r+1)*C) of a (world*C)-channel bank: independent channels,
    # per-GPU ranges, nothing exchanged on the data path
    ch_lo, ch_hi = pkg.shard.channel_range(C * world, world, rank)
    assert ch_hi - ch_lo == C
    iq, seeds = make_input(torch, pkg, device, C, N, seed=20260000 + ch_lo)
    stride = pkg.binding.bits_stride(N)
    bits = torch.zeros((C, stride), dtype=torch.uint8, device=device)
    nbits = torch.zeros(C, dtype=torch.int32, device=device)
    dem = pkg.Demodulator(C, N, device=local_rank, flags=0)
    stream = torch.cuda.current_stream(device)

    def step():
        dem.process_device(iq, N, bits, stride, nbits, None, stream)

    # Clock ramp: after ~1 s without work the GPU takes about five launches (25 ms) to reach its steady shader clock
    # (profiles/r02/r02_f_clock_ramp.md: 5.28, 4.92, 4.70, 4.57, 4.50, 4.48, 4.48 ... ms, the same again after every idle
    # second, independent of the data).  A receiver streams continuously, so the measurement belongs on steady clocks: the
    # device is first kept busy with RAMP_STEPS untimed passes and the loop state is reset to fresh, THEN the W warm-up and
    # the K timed steps run as the contract says.
    for _ in range(RAMP_STEPS):
        step()
    dem.reset()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = pkg.shard.max_over_ranks(dist, elapsed, device=device if args.backend == "nccl" else None)

    # per-launch kernel durations of the timed region (HIP events on the launch stream)
    nh = min(args.steps, 64)
    k1_ms = float(dem.kernel_ms_history(nh).mean())

    # size-independent property at full size: after lock every channel returns its transmitted bits -- on EVERY rank (each has its
    # own channel range and input); the counters are summed over the ranks after the timed region, one failing rank fails the job
    check = None
    if not args.no_check:
        dem.reset()
        step()
        torch.cuda.synchronize(device)
        every = max(1, C // CHECK_CHANNELS)
        hb = bits[::every].cpu().numpy()
        hn = nbits[::every].cpu().numpy()
        errs, ncmp = 0, 0
        for j, c in enumerate(range(0, C, every)):
            txb = pkg.synth.hash_bits(int(seeds[c]), pkg.synth.needed_bits(N))          # channel c's own stream, regenerated from its seed
            lag, e, n = pkg.synth.align_and_count_errors(hb[j][: hn[j]], txb, skip=3 * hn[j] // 4, max_lag=160)
            errs += e
            ncmp += n
        bad = int(errs > 1e-3 * ncmp or ncmp == 0)
        nchk, ranks_ok = len(hn), 1 - bad
        if dist is not None:
            nchk, ncmp, errs, ranks_ok, bad = pkg.shard.sum_over_ranks(dist, [nchk, ncmp, errs, ranks_ok, bad],
                                                                       device=device if args.backend == "nccl" else None)
        check = dict(channels_checked=int(nchk), bits_compared_last_quarter=int(ncmp), bit_errors=int(errs), ranks_checked=world,
                     ranks_green=int(ranks_ok))
        if bad:
            raise SystemExit("known-answer check failed on %d of %d rank(s): %d bit errors in %d bits after lock" % (bad, world, errs, ncmp))

    # The same workload handed over time-major, iq[n][c] (what a channeliser emits; the AGC wave's loads then coalesce across
    # the channel axis as north_star words it): a second handle on the transposed samples, informational A/B.
    tmaj = None
    if not args.no_time_major and rank == 0 and world == 1:
        iq_t = iq.transpose(0, 1).contiguous()
        dem_t = pkg.Demodulator(C, N, device=local_rank, flags=0, layout=pkg.binding.LAYOUT_TIME_MAJOR)
        bits_t = torch.zeros_like(bits)
        nbits_t = torch.zeros_like(nbits)
        reps = max(3, min(args.steps, 10))
        for _ in range(3):
            dem_t.process_device(iq_t, N, bits_t, stride, nbits_t, None, stream)
        dem_t.reset()
        dem.reset()
        dem_t.process_device(iq_t, N, bits_t, stride, nbits_t, None, stream)
        step()
        torch.cuda.synchronize(device)
        same = bool(torch.equal(nbits_t, nbits) and torch.equal(bits_t[:, : int(nbits.min())], bits[:, : int(nbits.min())]))
        # the comparison above left the GPU idle for a moment: bring the clock back up with untimed launches of BOTH handles,
        # then alternate them launch by launch so that neither layout is measured on a different clock than the other
        for _ in range(4):
            dem_t.process_device(iq_t, N, bits_t, stride, nbits_t, None, stream)
            step()
        for _ in range(reps):
            dem_t.process_device(iq_t, N, bits_t, stride, nbits_t, None, stream)
            step()
        torch.cuda.synchronize(device)
        t_ms = float(dem_t.kernel_ms_history(reps).mean())
        c_ms = float(dem.kernel_ms_history(reps).mean())
        tmaj = {"kernel_ms": round(t_ms, 4), "channel_major_kernel_ms_same_session": round(c_ms, 4),
                "msamples_s": round(C * N / t_ms / 1e3, 1), "bits_identical_to_channel_major": same,
                "note": "informational: the headline workload as iq[n][c] (TETRA_LAYOUT_TIME_MAJOR), %d launches of each layout, "
                        "alternating launch by launch on steady clocks" % reps}
        dem_t.close()
        del iq_t, bits_t, nbits_t

    # End-to-end host path (beside the metric, never the metric): host buffers in, host buffers out, PCIe
    # included.  Page-locked caller buffers throughout.  sync = tetra_demod_process (copy, kernel, copy);
    # async = tetra_demod_process_async (time chunks double-buffered, two calls in flight), float and int16 input.
    host = {}
    if not args.no_host_path and rank == 0 and world == 1:
        import ctypes
        B = pkg.binding
        lib = B.load_library()
        vp = ctypes.c_void_p
        h_iq = iq.cpu()
        p_iq = h_iq.pin_memory()
        p_bits = [torch.zeros((C, stride), dtype=torch.uint8).pin_memory() for _ in range(2)]
        p_nb = [torch.zeros(C, dtype=torch.int32).pin_memory() for _ in range(2)]
        p_q = (torch.view_as_real(h_iq) * 32768.0).round().clamp(-32768, 32767).to(torch.int16).pin_memory()
        del h_iq

        def sync_call():
            rc = lib.tetra_demod_process(dem._h, vp(p_iq.data_ptr()), N, vp(p_bits[0].data_ptr()), stride, vp(p_nb[0].data_ptr()), None)
            if rc:
                raise SystemExit("tetra_demod_process failed: %d" % rc)

        def timed(fn, reps, finish=None):
            dem.reset()
            fn(0)
            if finish:
                finish()
            t1 = time.perf_counter()
            for r in range(reps):
                fn(r)
            if finish:
                finish()
            return reps * C * N / (time.perf_counter() - t1) / 1e6

        host["host_path_pinned_msamples_s"] = round(timed(lambda r: sync_call(), 3), 1)
        host["host_path_async_msamples_s"] = round(timed(
            lambda r: dem.process_async(p_iq.data_ptr(), B.IQ_CF32, N, p_bits[r & 1].data_ptr(), stride, p_nb[r & 1].data_ptr()),
            6, dem.wait), 1)
        host["host_path_async_cs16_msamples_s"] = round(timed(
            lambda r: dem.process_async(p_q.data_ptr(), B.IQ_CS16, N, p_bits[r & 1].data_ptr(), stride, p_nb[r & 1].data_ptr()),
            6, dem.wait), 1)
        p_q8 = (torch.view_as_real(iq.cpu()) * 128.0).round().clamp(-128, 127).to(torch.int8).pin_memory()
        host["host_path_async_cs8_msamples_s"] = round(timed(
            lambda r: dem.process_async(p_q8.data_ptr(), B.IQ_CS8, N, p_bits[r & 1].data_ptr(), stride, p_nb[r & 1].data_ptr()),
            6, dem.wait), 1)
        del p_q8
        host["host_path_note"] = ("PCIe-inclusive, page-locked host buffers, informational: sync = tetra_demod_process; async = "
                                  "tetra_demod_process_async with two calls in flight; cs16 / cs8 = int16 / int8 IQ converted on the GPU")
        del p_iq, p_q, p_bits, p_nb

    # Twice the channels on the same GPU (informational, never the metric): above one 16-channel workgroup per CU the library
    # launches 32-channel workgroups whose FLL rows carry 16 channels per wave.
    large = None
    if not args.no_large_batch and rank == 0 and world == 1 and C == CHANNELS_PER_GPU:
        iq2 = torch.cat([iq, iq])
        bits2 = torch.zeros((2 * C, stride), dtype=torch.uint8, device=device)
        nb2 = torch.zeros(2 * C, dtype=torch.int32, device=device)
        dem2 = pkg.Demodulator(2 * C, N, device=local_rank, flags=0)
        for _ in range(RAMP_STEPS):          # steady clocks, like the headline (allocating the buffers above left the GPU idle)
            dem2.process_device(iq2, N, bits2, stride, nb2, None, stream)
        torch.cuda.synchronize(device)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 6
        ev0.record(stream)
        for _ in range(reps):
            dem2.process_device(iq2, N, bits2, stride, nb2, None, stream)
        ev1.record(stream)
        torch.cuda.synchronize(device)
        ms2 = ev0.elapsed_time(ev1) / reps
        same = bool(torch.equal(nb2[:C], nb2[C:]) and torch.equal(bits2[:C], bits2[C:]))      # both halves saw the same input
        large = {"channels": 2 * C, "ms_per_step": round(ms2, 4), "msamples_s": round(2.0 * C * N / ms2 / 1e3, 1),
                 "halves_identical": same, "traffic": pmc_traffic("fused", 2 * C, N)[0],
                 "cycles_per_sample": round(ms2 * 1e-3 * pkg.binding.device_info(local_rank)[0] * 1e3 / N, 1),
                 "note": "informational: %d channels x %d samples on this GPU in one call (32-channel workgroups)" % (2 * C, N)}
        dem2.close()
        del iq2, bits2, nb2

    if rank == 0:
        total_samples = float(world) * C * N * args.steps
        value = total_samples / elapsed / 1e6
        algo_bytes = ALGO_BYTES_PER_SAMPLE * C * N
        traffic, issue, traffic_unit = pmc_traffic("fused", C, N)
        achieved = algo_bytes / (k1_ms * 1e-3) / 1e9
        # the bound that actually binds: FP32 vector issue.  Useful flops / launch time against the vector peak, and the shader
        # clocks one workgroup (16 channels, one CU) spends per sample of its channels
        clk_khz, cus = pkg.binding.device_info(local_rank)
        clk_hz = clk_khz * 1e3
        valu_tflops = FLOP_PER_SAMPLE * C * N / (k1_ms * 1e-3) / 1e12
        waves = (C + 15) // 16
        rounds = max(1, -(-waves // cus))          # workgroups per CU, executed one after the other at this LDS footprint
        valu = {"achieved": round(valu_tflops, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(valu_tflops / VALU_PEAK_TFLOPS, 4), "flop_per_sample": FLOP_PER_SAMPLE,
                "cycles_per_sample": round(k1_ms * 1e-3 * clk_hz / N / rounds, 1),
                "cycles_note": "launch time x %.0f MHz / samples per channel / %d workgroup round(s) per CU" % (clk_hz / 1e6, rounds)}
        out = {
            "metric": "IQ Msamples/s demodulated to bits, batched TETRA channels",
            "value": None if folded else round(value, 3), "unit": "Msamples/s", "n_gpus": min(world, gpus_physical),
            "n_ranks": world, "gpus_physical": gpus_physical, "functional_only": folded, "steps": args.steps,
            "warmup": args.warmup, "ramp_steps": RAMP_STEPS, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%d channels/GPU x %d complex64 samples (1 s @ 36 ksps), 65-tap RRC, pi/4-DQPSK Es/N0 25 dB, every "
                                   "channel from its own seed (bits, carrier / timing offset, amplitude, phase), state carried" % (C, N),
                       "channels_per_gpu": C, "samples_per_channel": N, "sharding": "channel ranges, no collective",
                       "pipeline": "fused"},
            "roofline": {"bound": "hbm", "kernel": "k_fused",
                         "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic,
                         "traffic_unit": traffic_unit,
                         "valu": valu,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "kernel_ms": round(k1_ms, 4),
                         "note": "HBM is the roofline BASELINE.json names; the kernel itself is VALU-issue bound "
                                 "(per-channel serial recurrences), see DESIGN.md",
                         "issue_counters": issue},
            "check": check,
            "ramp_note": "%d untimed launches + a state reset precede the %d warm-up steps: after an idle second the shader clock "
                         "needs ~5 launches to reach its steady value (profiles/r02/r02_f_clock_ramp.md)" % (RAMP_STEPS, args.warmup),
        }
        if folded:
            out["value_functional"] = round(value, 3)
            out["functional_note"] = ("%d ranks shared %d GPU(s): a functional run of the N > 1 path, NOT an %d-GPU measurement"
                                      % (world, gpus_physical, world))
        if tmaj is not None:
            out["time_major"] = tmaj
        out.update(host)
        if large is not None:
            out["large_batch"] = large
        if not args.no_config5 and world == 1 and C == CHANNELS_PER_GPU:
            dem.close()
            dem = None
            out["config5"] = informational(wideband_config5, args, torch, pkg, device, local_rank)      # after the timed region
        if dist is not None:
            out["rccl_world_size"] = dist.get_world_size() if args.backend == "nccl" else None
            out["dist_backend"] = args.backend
            if backend_note:
                out["dist_backend_note"] = backend_note
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["cpu_baseline_fast"] = cpu_baseline(pkg.synth, N)
        if not args.no_chain and world == 1 and C == CHANNELS_PER_GPU:
            if dem is not None:
                dem.close()
            dem = None
            del iq, bits, nbits
            out["chain"] = informational(receive_chain, args, torch, pkg, device, local_rank)      # after the timed region
        print(json.dumps(out))
    if dem is not None:
        dem.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
