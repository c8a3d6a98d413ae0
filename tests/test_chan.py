"""Channeliser front-end (BASELINE config 5): definition-level oracle on CPU, GPU kernel
against it within a float32 tolerance, and the end-to-end chain wideband -> channeliser -> demodulator -> bits."""
import numpy as np
import pytest


def _wideband(synth, M, n_frames, D, carriers, seed=0, sps_out=None):
    """Wideband stream at Fs = M * 25 kHz holding TETRA carriers {channel k: seed}; returns (x, {k: tx_bits})."""
    fs = M * 25000.0
    n = n_frames * D
    x = np.zeros(n, np.complex128)
    tx = {}
    t = np.arange(n)
    for k, sd in carriers.items():
        sps = fs / 18000.0
        bits = np.random.default_rng(sd).integers(0, 2, synth.needed_bits(n, sps), dtype=np.uint8)
        s = synth.modulate(bits, n, sps=sps, beta=0.35, tau=0.37 * sps, ppm=1e-9)   # ppm != 0: general (non-integer sps) path
        kk = k if k <= M // 2 else k - M
        x += 0.3 * s * np.exp(2j * np.pi * kk / M * t)
        tx[k] = bits
    rng = np.random.default_rng(seed)
    x += 0.003 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64), tx


def test_oracle_tone_lands_in_its_channel(oracle):
    M, P, D = 32, 8, 16
    co = oracle.ChanOracle(M, P, D)
    assert abs(co.h.sum() - 1.0) < 1e-5 and np.allclose(co.h, co.h[::-1], atol=1e-9)
    n = 40 * D
    k0 = 5
    x = np.exp(2j * np.pi * k0 / M * np.arange(n)).astype(np.complex64)
    y = co.process(x)
    assert y.shape == (40, M)
    steady = np.abs(y[20:])
    assert np.allclose(steady[:, k0], 1.0, atol=1e-3)                    # unity passband gain, mixed to DC
    others = np.delete(steady, [k0 - 1, k0, k0 + 1], axis=1)
    assert others.max() < 1e-3                                           # > 60 dB rejection two channels away
    assert np.abs(np.angle(y[21:, k0] / y[20:-1, k0])).max() < 1e-3      # constant phase: really at DC


def test_oracle_chunk_invariance(oracle):
    M, P, D = 32, 4, 16
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(700) + 1j * rng.standard_normal(700)).astype(np.complex64)
    ref = oracle.ChanOracle(M, P, D).process(x)
    co = oracle.ChanOracle(M, P, D)
    parts = [co.process(x[a:b]) for a, b in ((0, 5), (5, 37), (37, 38), (38, 400), (400, 700))]
    assert np.array_equal(np.concatenate([p for p in parts if len(p)]), ref)


def test_oracle_memoised_phasors_are_the_definition(oracle):
    """chan_oracle.c evaluates exp(-j 2 pi (k n mod M) / M) once per residue: the same double values as the definition's
    per-term cos / sin -- checked against the definition written out in numpy for one small frame."""
    M, P, D = 12, 3, 5
    co = oracle.ChanOracle(M, P, D)
    rng = np.random.default_rng(2)
    x = (rng.standard_normal(4 * D) + 1j * rng.standard_normal(4 * D)).astype(np.complex64)
    y = co.process(x)
    L = M * P
    xp = np.concatenate([np.zeros(L - 1, np.complex128), x.astype(np.complex128)])
    for m in range(4):
        n_m = (m + 1) * D - 1
        for k in range(M):
            acc = 0j
            for l in range(L):
                n = n_m - l
                acc += float(co.h[l]) * xp[L - 1 + n] * np.exp(-2j * np.pi * ((k * n) % M) / M)
            assert abs(y[m, k] - acc) < 1e-6 * (1 + abs(acc))


def test_fft_kernel_lane_code_on_the_host_matches_the_definition(oracle):
    """Round 5: M = 800 at D = 400 runs as a 32 x 5 x 5 mixed-radix FFT (csrc/chan_fft_core.hpp).  The kernel's lane-level source --
    fold with its slot / class maps, 32-point FFT, transposed LDS block, twiddle, 5 x 5 DFT, store map -- compiled for the host and run
    thread by thread, phase by phase (tests/emul/chan_emul.cpp), against the double-precision definition: the two register-level
    transforms against numpy's FFT, whole frames for 4 / 6 / 8 taps per channel with ragged chunks, carried history and sub-frame
    phase (the samples past a call's end that the last block reads are NaN here: they must never reach a stored frame)."""
    from tests.emul import chan_emul_bind as ce
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(32) + 1j * rng.standard_normal(32)).astype(np.complex64)
    assert np.abs(ce.fft32(x) - np.fft.fft(x.astype(np.complex128))).max() < 3e-6
    x = (rng.standard_normal(25) + 1j * rng.standard_normal(25)).astype(np.complex64)
    assert np.abs(ce.dft25(x) - np.fft.fft(x.astype(np.complex128))).max() < 3e-6
    for P in (8, 6, 4):
        co = oracle.ChanOracle(800, P, 400)
        em = ce.ChanFftEmul(P, co.h)
        nin = 400 * 37 + 123
        x = (rng.standard_normal(nin) + 1j * rng.standard_normal(nin)).astype(np.complex64)
        cuts = [0, 7, 7 + 399, nin // 3, nin // 3 + 1, nin]
        for a, b in zip(cuts, cuts[1:]):
            yo, ye = co.process(x[a:b]), em.process(x[a:b])
            assert yo.shape == ye.shape
            if len(yo):
                assert np.isfinite(ye).all() and np.abs(yo - ye).max() / np.abs(yo).max() < 2e-6, (P, a, b)


def _quantise(x, dtype):
    """complex samples of roughly unit scale -> integer I / Q pairs [n][2] and the complex64 values they stand for (integer / 32768 or / 128)."""
    full = 32768 if dtype == np.int16 else 128
    q = np.stack([np.clip(np.round(x.real * full / 6), -full, full - 1), np.clip(np.round(x.imag * full / 6), -full, full - 1)], axis=1).astype(dtype)
    return q, (q[:, 0].astype(np.float32) + 1j * q[:, 1].astype(np.float32)).astype(np.complex64) / np.float32(full)


@pytest.mark.parametrize("dtype", [np.int16, np.int8])
def test_fft_kernel_lane_code_with_integer_samples(oracle, dtype):
    """Round 6: the FFT kernel's fold reading what an SDR delivers -- interleaved int16 / int8 I, Q pairs, converted in the load
    (tetra_chan_process_device_cs16 / _cs8).  Same lane code on the host: against the definition fed the quantised samples, and EQUAL
    to the complex64 route on the converted samples (the conversion is exact in binary32), with ragged chunks and carried history."""
    from tests.emul import chan_emul_bind as ce
    rng = np.random.default_rng(5)
    co = oracle.ChanOracle(800, 8, 400)
    em_i, em_f = ce.ChanFftEmul(8, co.h), ce.ChanFftEmul(8, co.h)
    nin = 400 * 21 + 77
    q, xq = _quantise(rng.standard_normal(nin) + 1j * rng.standard_normal(nin), dtype)
    cuts = [0, 9, 9 + 399, nin // 2, nin]
    for a, b in zip(cuts, cuts[1:]):
        yo, yi, yf = co.process(xq[a:b]), em_i.process(q[a:b]), em_f.process(xq[a:b])
        assert yo.shape == yi.shape and np.array_equal(yi, yf)
        if len(yo):
            assert np.abs(yo - yi).max() / np.abs(yo).max() < 2e-6, (a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["int16", "int8"])
@pytest.mark.parametrize("M,P,D,flags", [(800, 8, 400, 0), (800, 6, 400, 0), (800, 8, 400, 2), (800, 8, 400, 1), (32, 8, 16, 0)])
def test_gpu_integer_input_matches_definition_and_the_float_route(pkg, oracle, M, P, D, flags, dtype):
    """tetra_chan_process_device_cs16 / _cs8 on the GPU: the FFT kernel reads the integer capture in place, the
    matrix / direct-sum kernels through a converting staging copy; against the double-precision definition on the quantised samples
    (2e-5) and bit for bit the complex64 entry point on the converted samples; ragged chunks, formats mixed on one handle."""
    import torch
    dev = torch.device("cuda", 0)
    np_dtype = np.int16 if dtype == "int16" else np.int8
    rng = np.random.default_rng(M + P)
    nin = D * 150 + 11
    q, xq = _quantise(rng.standard_normal(nin) + 1j * rng.standard_normal(nin), np_dtype)
    ch_i = pkg.Channeliser(M, P, D, max_in=nin, flags=flags)
    ch_f = pkg.Channeliser(M, P, D, max_in=nin, flags=flags)
    co = oracle.ChanOracle(M, P, D)
    d_q = torch.from_numpy(q).to(dev)
    d_x = torch.from_numpy(xq).to(dev)
    cuts = [0, 5, 5 + D - 1, nin // 3, nin // 3 + 2, nin]
    for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
        yo = co.process(xq[a:b])
        out_i = torch.zeros((max(1, ch_i.frames_for(b - a)), M), dtype=torch.complex64, device=dev)
        out_f = torch.zeros_like(out_i)
        # one chunk of the integer handle goes in as complex64: formats may be mixed (the delay line is complex64)
        n_i = ch_i.process_device(d_x[a:b] if i == 3 else d_q[a:b], b - a, out_i)
        n_f = ch_f.process_device(d_x[a:b], b - a, out_f)
        torch.cuda.synchronize()
        assert n_i == n_f == len(yo)
        if n_i:
            assert torch.equal(out_i[:n_i], out_f[:n_f]), (a, b)
            yg = out_i[:n_i].cpu().numpy()
            scale = np.abs(yo).max() + 1e-12
            assert np.abs(yg - yo).max() / scale < 2e-5, (a, b)
    with pytest.raises(pkg.TetraDemodError):
        ch_i.process_device(d_q.reshape(-1)[1:], 4, out_i)      # a pointer into the middle of an I / Q pair: TETRA_ERR_ALIGN
    ch_i.close()
    ch_f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,P,D,nin,flags", [(32, 8, 16, 3000, 0), (800, 8, 400, 800 * 5, 0), (800, 8, 400, 800 * 5, 1), (800, 8, 400, 800 * 5, 2), (60, 4, 20, 1234, 0),
                                             (32, 4, 32, 1000, 0), (800, 8, 400, 400 * 1000 + 123, 0), (800, 8, 400, 400 * 120 + 7, 1),
                                             (800, 8, 400, 400 * 1000 + 123, 2), (800, 6, 400, 400 * 130 + 399, 0), (800, 4, 400, 400 * 41 + 1, 0),
                                             (800, 6, 800, 800 * 40, 0)])
def test_gpu_matches_definition(pkg, oracle, M, P, D, nin, flags):
    """M = 800 at D = 400 (BASELINE config 5's geometry) runs its DFT as a 32 x 5 x 5 mixed-radix FFT in registers / LDS by default
    (round 5), with flags 2 = TETRA_CHAN_FLAG_MATRIX_DFT as 25 x 32 matrix products on the matrix pipe (round 4's form; also what
    other decimations of M = 800 take) and with flags 1 = TETRA_CHAN_FLAG_VALU_DFT as direct sums; all against the double-precision
    definition, also at size (1000 frames of config 5's geometry: VERDICT r3 item 7), for 4 / 6 / 8 taps per channel and with a
    critically sampled bank (D = M)."""
    rng = np.random.default_rng(M)
    x = (rng.standard_normal(nin) + 1j * rng.standard_normal(nin)).astype(np.complex64)
    ch = pkg.Channeliser(M, P, D, max_in=nin, flags=flags)
    co = oracle.ChanOracle(M, P, D)
    assert np.array_equal(ch.prototype(), co.h)
    # ragged chunking with carried history and sub-frame phase
    cuts = [0, 7, 7 + D - 1, nin // 3, nin // 3 + 1, nin]
    for a, b in zip(cuts, cuts[1:]):
        yg = ch.process(x[a:b])
        yo = co.process(x[a:b])
        assert yg.shape == yo.shape
        if len(yo):
            scale = np.abs(yo).max() + 1e-12
            assert np.abs(yg - yo).max() / scale < 2e-5, (a, b, np.abs(yg - yo).max() / scale)   # float32 DFT vs double definition
    ch.close()


@pytest.mark.gpu
def test_gpu_wideband_to_bits(pkg, synth):
    """End to end (BASELINE config 5 in small): 32 x 25 kHz channels at Fs = 800 kHz, three TETRA carriers (one at a
    negative frequency), channeliser at 2x oversampling -> 50 ksps per channel -> demodulator (time-major frames,
    samplerate 50000) -> the transmitted bits of every occupied channel come back after lock."""
    M, P, D = 32, 8, 16
    n_frames = 30000                                    # 0.6 s at 50 ksps per channel
    carriers = {3: 11, 10: 12, 27: 13}
    x, tx = _wideband(synth, M, n_frames, D, carriers)
    ch = pkg.Channeliser(M, P, D, max_in=x.shape[0])
    frames = ch.process(x)
    assert frames.shape == (n_frames, M)
    occupied = np.abs(frames[2000:]).mean(0)
    for k in carriers:
        assert occupied[k] > 20 * np.median(occupied)
    dem = pkg.Demodulator(M, n_frames, layout=pkg.binding.LAYOUT_TIME_MAJOR, samplerate=50000.0)
    bits, nb, _ = dem.process(frames)
    for k, b in tx.items():
        lag, err, n = synth.align_and_count_errors(bits[k][:nb[k]], b, skip=3 * nb[k] // 4, max_lag=600)
        assert n > 4000 and err <= 2, (k, lag, err, n)
    ch.close()
    dem.close()


@pytest.mark.gpu
def test_gpu_channeliser_then_demodulator_equals_the_oracle_chain(pkg, oracle, synth):
    """VERDICT r3 item 7, second half: BASELINE config 5's geometry (800 channels, 8 taps per channel, D = 400, 50 ksps per
    channel) on a capture with three TETRA carriers.  GPU chain: channeliser (matrix-pipe DFT) -> demodulator.  Oracle chain:
    the double-precision definition's frames -> the demodulator oracle.  The two front-ends differ at the float32 level
    (asserted: <= 2e-5 of the peak), so the decision streams are compared where the loops have locked: the last third of every
    carrier's bits is equal, and equal to the transmitted bits."""
    M, P, D = 800, 8, 400
    n_frames = 5400
    carriers = {7: 21, 413: 22, 790: 23}
    x, tx = _wideband(synth, M, n_frames, D, carriers, seed=3)
    ch = pkg.Channeliser(M, P, D, max_in=x.shape[0])
    frames = ch.process(x)
    ch.close()
    # the definition: every channel for the first 150 frames, then (a second oracle, whole stream) the carriers' channels only
    head = oracle.ChanOracle(M, P, D).process(x[:150 * D])
    assert frames.shape == (n_frames, M) and head.shape == (150, M)
    assert np.abs(frames[:150] - head).max() / np.abs(head).max() < 2e-5
    ks = sorted(carriers)
    cols = oracle.ChanOracle(M, P, D).process(x, channels=ks)
    assert np.abs(frames[:, ks] - cols).max() / np.abs(cols).max() < 2e-5
    want = {k: cols[:, i] for i, k in enumerate(ks)}
    dem = pkg.Demodulator(M, n_frames, layout=pkg.binding.LAYOUT_TIME_MAJOR, samplerate=50000.0)
    bits, nb, _ = dem.process(frames)
    dem.close()
    cfg = oracle.default_cfg()
    cfg.samplerate = 50000.0
    for k, b in tx.items():
        r = oracle.Oracle(cfg).process(np.ascontiguousarray(want[k]))
        n = min(nb[k], r["bits"].size)
        assert abs(int(nb[k]) - r["bits"].size) <= 2
        lag_g, err_g, n_g = synth.align_and_count_errors(bits[k][:nb[k]], b, skip=2 * nb[k] // 3, max_lag=600)
        lag_o, err_o, n_o = synth.align_and_count_errors(r["bits"], b, skip=2 * r["bits"].size // 3, max_lag=600)
        assert n_g > 1000 and err_g == 0 and err_o == 0 and lag_g == lag_o, (k, lag_g, err_g, lag_o, err_o)
        assert np.array_equal(bits[k][2 * n // 3:n - 8], r["bits"][2 * n // 3:n - 8]), k
