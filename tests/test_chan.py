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


@pytest.mark.gpu
@pytest.mark.parametrize("M,P,D,nin", [(32, 8, 16, 3000), (800, 8, 400, 800 * 5), (60, 4, 20, 1234), (32, 4, 32, 1000)])
def test_gpu_matches_definition(pkg, oracle, M, P, D, nin):
    rng = np.random.default_rng(M)
    x = (rng.standard_normal(nin) + 1j * rng.standard_normal(nin)).astype(np.complex64)
    ch = pkg.Channeliser(M, P, D, max_in=nin)
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
