"""Rational resampler between the 50 ksps channeliser and the demodulator at the plugin's own operating point (36 ksps, 2 samples
per symbol: /root/reference/src/main.cpp:35,75,84; SURVEY.md section 8(f) #1 "PFB at 50 ksps + 18/25 rational resampler"):
definition-level oracle on the CPU, the kernels' thread code emulated on the host, the GPU kernels against the definition within a
float32 tolerance, and wideband -> channeliser -> resampler -> demodulator with DEFAULT parameters against definition -> definition
-> demodulator oracle."""
import numpy as np
import pytest

TOL = 2e-5          # float32 sums of <= 24 products against double sums, relative to the peak


def _frames(rng, n, C):
    return (rng.standard_normal((n, C)) + 1j * rng.standard_normal((n, C))).astype(np.complex64)


def test_oracle_is_the_textbook_form(oracle):
    """zero-stuff by I, convolve with h, keep every DN-th -- written out with numpy for one channel."""
    for I, DN, T in ((18, 25, 16), (2, 3, 8), (3, 2, 5), (1, 2, 4), (7, 1, 3)):
        ro = oracle.ResampOracle(1, I, DN, T)
        assert abs(ro.h.sum() - I) < 1e-4 * I and np.allclose(ro.h, ro.h[::-1], atol=1e-7)
        rng = np.random.default_rng(I * 100 + DN)
        x = _frames(rng, 200, 1)
        y = ro.process(x)[:, 0]
        up = np.zeros(200 * I, np.complex128)
        up[::I] = x[:, 0]
        v = np.convolve(up, ro.h.astype(np.float64))[: 200 * I]
        want = v[::DN]
        assert y.shape[0] == (200 * I + DN - 1) // DN == want.shape[0]
        assert np.abs(y - want).max() < 1e-6 * np.abs(want).max()


def test_oracle_tone_and_alias_rejection(oracle):
    """18 / 25 at the default design: a tone in the TETRA band passes with unity gain at its own frequency; a tone that would alias
    INTO the band (24.5 kHz at 50 ksps folds to -11.5 kHz at 36 ksps) is suppressed by > 50 dB."""
    ro = oracle.ResampOracle(2, 18, 25, 16)
    n = 5000
    t = np.arange(n)
    x = np.stack([np.exp(2j * np.pi * 9000.0 / 50000.0 * t), np.exp(2j * np.pi * 24500.0 / 50000.0 * t)], 1).astype(np.complex64)
    y = ro.process(x)
    assert y.shape == (3600, 2)
    s = y[200:]
    assert np.allclose(np.abs(s[:, 0]), 1.0, atol=3e-3)
    assert np.abs(np.angle(s[1:, 0] / s[:-1, 0]) - 2 * np.pi * 9000.0 / 36000.0).max() < 1e-3
    assert np.abs(s[:, 1]).max() < 10 ** (-50 / 20)


def test_oracle_chunk_invariance(oracle):
    rng = np.random.default_rng(3)
    x = _frames(rng, 333, 6)
    ref = oracle.ResampOracle(6).process(x)
    ro = oracle.ResampOracle(6)
    cuts = [0, 1, 2, 3, 20, 20, 47, 48, 200, 333]
    parts = [ro.process(x[a:b]) for a, b in zip(cuts, cuts[1:])]
    assert np.array_equal(np.concatenate(parts), ref)


@pytest.mark.parametrize("I,DN,T,C,generic,W", [(18, 25, 16, 10, False, 4), (18, 25, 16, 10, False, 2), (18, 25, 16, 7, False, 2),
                                                (18, 25, 8, 4, False, 4), (18, 25, 12, 6, False, 4), (18, 25, 24, 2, False, 4),
                                                (2, 3, 8, 4, False, 4), (3, 2, 8, 3, False, 2), (1, 2, 8, 8, False, 4),
                                                (18, 25, 16, 10, True, 4), (5, 7, 11, 3, True, 2), (4, 1, 3, 2, True, 4)])
def test_kernel_thread_code_on_the_host_matches_the_definition(oracle, I, DN, T, C, generic, W):
    """The resampler kernels' thread-level source (csrc/resamp_core.hpp) compiled for the host and run over exactly the thread range
    the C ABI launches (tests/emul/resamp_emul.cpp), with the carried delay line and positions, against the definition: ragged calls
    (0 / 1 / 2 frames, calls shorter than the delay line, calls that end inside a group of I outputs), every stored element written
    (the output buffer is NaN-poisoned and exactly sized), for the specialised kernels, both lane-unit widths, and the generic one."""
    from tests.emul import resamp_emul_bind as re_
    ro = oracle.ResampOracle(C, I, DN, T)
    em = re_.ResampEmul(C, I, DN, T, ro.h, generic=generic, W=W)
    rng = np.random.default_rng(I + 7 * DN + T)
    n = 40 * DN + 13
    x = _frames(rng, n, C)
    cuts = [0, 0, 1, 3, 4, T - 1 + 4, 3 * DN, 3 * DN + 1, 7 * DN + 5, 7 * DN + 6, 30 * DN, n]
    cuts = sorted(set(cuts)) + [n]
    cuts = [0] + cuts          # an empty first call
    for a, b in zip(cuts, cuts[1:]):
        yo, ye = ro.process(x[a:b]), em.process(x[a:b])
        assert yo.shape == ye.shape, (a, b)
        if len(yo):
            assert np.isfinite(ye).all(), (a, b)
            assert np.abs(yo - ye).max() / np.abs(yo).max() < 2e-6, (a, b)


def test_resampler_header_symbols_all_exported(pkg):
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "tetra_chan.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(tetra_resamp_[a-z0-9_]+)\s*\(", src)))
    L = pkg.load_library()
    assert len(names) == 9 and set(names) == set(pkg.chan_binding.RESAMP_EXPORTS)
    for n in names:
        assert hasattr(L, n), n


@pytest.mark.gpu
@pytest.mark.parametrize("I,DN,T,C,nin,flags", [(18, 25, 16, 800, 2500, 0), (18, 25, 16, 800, 2500, 2), (18, 25, 16, 800, 700, 1),
                                                (18, 25, 8, 64, 999, 0), (18, 25, 12, 30, 999, 0), (18, 25, 24, 800, 1201, 0),
                                                (18, 25, 16, 7, 1500, 0), (2, 3, 8, 32, 1000, 0), (3, 2, 8, 5, 1000, 0),
                                                (1, 2, 8, 16, 1000, 0), (5, 7, 11, 12, 800, 0), (160, 147, 10, 4, 600, 0)])
def test_gpu_matches_definition(pkg, oracle, I, DN, T, C, nin, flags):
    """Specialised kernels (16-byte and 8-byte lane units), the generic kernel forced and by necessity (5 / 7, 160 / 147), an odd
    channel count; ragged calls with carried delay line and position; against the double-precision definition."""
    rng = np.random.default_rng(I * 31 + C)
    x = _frames(rng, nin, C)
    rs = pkg.Resampler(C, I, DN, T, max_in=nin, flags=flags)
    ro = oracle.ResampOracle(C, I, DN, T)
    assert np.array_equal(rs.prototype(), ro.h)
    cuts = sorted(set([0, 1, 3, T + 2, nin // 3, nin // 3 + 1, nin]))
    for a, b in zip(cuts, cuts[1:]):
        assert rs.frames_for(b - a) == ro.frames_for(b - a)
        yg, yo = rs.process(x[a:b]), ro.process(x[a:b])
        assert yg.shape == yo.shape
        if len(yo):
            assert np.abs(yg - yo).max() / (np.abs(yo).max() + 1e-12) < TOL, (a, b)
    rs.reset()
    ro2 = oracle.ResampOracle(C, I, DN, T)
    yg, yo = rs.process(x[:50]), ro2.process(x[:50])
    assert np.abs(yg - yo).max() / np.abs(yo).max() < TOL
    rs.close()


@pytest.mark.gpu
def test_gpu_caller_prototype_and_argument_checks(pkg, oracle):
    C_, I, DN, T = 16, 18, 25, 16
    proto = np.hanning(I * T).astype(np.float32)
    proto *= I / proto.sum()
    rs = pkg.Resampler(C_, I, DN, T, max_in=400, prototype=proto)
    ro = oracle.ResampOracle(C_, I, DN, T, prototype=proto)
    x = _frames(np.random.default_rng(5), 400, C_)
    yg, yo = rs.process(x), ro.process(x)
    assert yg.shape == yo.shape == (288, C_) and np.abs(yg - yo).max() / np.abs(yo).max() < TOL
    with pytest.raises(pkg.TetraDemodError) as e:
        rs.process(np.zeros((401, C_), np.complex64))
    assert e.value.code == pkg.binding.ERR_SIZE if hasattr(pkg.binding, "ERR_SIZE") else True
    rs.close()
    for bad in (dict(interp=0), dict(decim=0), dict(taps_per_phase=1), dict(taps_per_phase=65), dict(flags=4), dict(cutoff_rel=0.0)):
        kw = dict(n_channels=8, interp=18, decim=25, taps_per_phase=16, max_in=10)
        kw.update(bad)
        with pytest.raises(pkg.TetraDemodError):
            pkg.Resampler(**kw)


def _wideband(synth, M, n_frames, D, carriers, seed=0):
    from tests.test_chan import _wideband as wb
    return wb(synth, M, n_frames, D, carriers, seed=seed)


@pytest.mark.gpu
def test_gpu_wideband_to_bits_at_the_plugins_rate(pkg, oracle, synth):
    """BASELINE config 5's geometry at the plugin's operating point: 800 channels x 8 taps, D = 400 (50 ksps) -> 18 / 25 -> 36 ksps ->
    demodulator created with the DEFAULT configuration (36 ksps, 2 samples per symbol, 65-tap RRC, the loop constants of
    src/main.cpp:35-44,78-84) on a capture with three TETRA carriers.  Oracle chain: the double-precision channeliser definition's
    frames of the carriers' channels -> the double-precision resampler definition -> the demodulator oracle at default parameters.
    The front-ends differ at the float32 level (asserted: <= 2e-5 of the peak after each stage), so the decision streams are compared
    where the loops have locked: the last third of every carrier's bits is equal, and equal to the transmitted bits."""
    M, P, D = 800, 8, 400
    n_frames = 7500
    carriers = {7: 21, 413: 22, 790: 23}
    x, tx = _wideband(synth, M, n_frames, D, carriers, seed=3)
    ch = pkg.Channeliser(M, P, D, max_in=x.shape[0])
    frames = ch.process(x)
    ch.close()
    rs = pkg.Resampler(M, max_in=n_frames)
    y = rs.process(frames)
    rs.close()
    assert y.shape == (n_frames * 18 // 25, M)
    ks = sorted(carriers)
    cols = oracle.ChanOracle(M, P, D).process(x, channels=ks)
    assert np.abs(frames[:, ks] - cols).max() / np.abs(cols).max() < TOL
    ycols = oracle.ResampOracle(len(ks)).process(cols)
    assert ycols.shape == (y.shape[0], len(ks))
    assert np.abs(y[:, ks] - ycols).max() / np.abs(ycols).max() < 2 * TOL
    # the GPU resampler on the GPU channeliser's frames against the definition on those same frames, every channel, first 500 frames
    yo = oracle.ResampOracle(M).process(frames[:500])
    assert np.abs(y[:yo.shape[0]] - yo).max() / np.abs(yo).max() < TOL
    n_out = y.shape[0]
    dem = pkg.Demodulator(M, n_out, layout=pkg.binding.LAYOUT_TIME_MAJOR)          # default configuration: the plugin's
    bits, nb, _ = dem.process(y)
    dem.close()
    cfg = oracle.default_cfg()
    assert cfg.samplerate == 36000.0 and cfg.symbolrate == 18000.0
    for i, k in enumerate(ks):
        b = tx[k]
        r = oracle.Oracle(cfg).process(np.ascontiguousarray(ycols[:, i]))
        n = min(nb[k], r["bits"].size)
        assert abs(int(nb[k]) - r["bits"].size) <= 2
        lag_g, err_g, n_g = synth.align_and_count_errors(bits[k][:nb[k]], b, skip=2 * nb[k] // 3, max_lag=600)
        lag_o, err_o, n_o = synth.align_and_count_errors(r["bits"], b, skip=2 * r["bits"].size // 3, max_lag=600)
        assert n_g > 1000 and err_g == 0 and err_o == 0 and lag_g == lag_o, (k, lag_g, err_g, lag_o, err_o)
        assert np.array_equal(bits[k][2 * n // 3:n - 8], r["bits"][2 * n // 3:n - 8]), k


@pytest.mark.gpu
def test_gpu_sixteen_carriers_known_answer_on_device(pkg, synth):
    """The bench leg in small: 16 carriers over a noise floor, everything through the *_device entry points on one stream
    (channeliser -> resampler -> default demodulator), two passes; every carrier's transmitted bits come back after lock."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    M, P, D = 800, 8, 400
    n_in = 2000000                       # 0.1 s of the 20 MHz capture
    carriers = {k: 900 + i for i, k in enumerate((3, 57, 101, 150, 199, 250, 313, 377, 423, 480, 531, 590, 644, 700, 751, 797))}
    x, tx = bench.wideband_tetra(torch, synth, dev, M, n_in, carriers)
    frames = n_in // D
    n36 = frames * 18 // 25
    ch = pkg.Channeliser(M, P, D, max_in=n_in)
    rs = pkg.Resampler(M, max_in=frames)
    dem = pkg.Demodulator(M, n36, layout=pkg.binding.LAYOUT_TIME_MAJOR)
    f50 = torch.zeros((frames, M), dtype=torch.complex64, device=dev)
    f36 = torch.zeros((n36 + 1, M), dtype=torch.complex64, device=dev)
    stride = dem.bits_stride(n36)
    bits = torch.zeros((M, stride), dtype=torch.uint8, device=dev)
    nbits = torch.zeros(M, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream(dev)
    tot = np.zeros(M, np.int64)
    rows = [[] for _ in range(M)]
    for _ in range(3):
        nf = ch.process_device(x, n_in, f50, s)
        no = rs.process_device(f50, nf, f36, s)
        assert nf == frames and no == n36
        dem.process_device(f36, no, bits, stride, nbits, None, s)
        torch.cuda.synchronize(dev)
        hb, hn = bits.cpu().numpy(), nbits.cpu().numpy()
        for k in tx:
            rows[k].append(hb[k][: hn[k]].copy())
    errs = ncmp = 0
    for k, b in tx.items():
        got = rows[k][-1]                    # third pass: loops locked (the capture repeats: one phase jump per pass)
        lag, e, n = synth.align_and_count_errors(got, b, skip=got.size // 2, max_lag=600)
        errs += e
        ncmp += n
    assert ncmp > 16 * 1500 and errs <= 1e-3 * ncmp, (errs, ncmp)
    ch.close(); rs.close(); dem.close()
