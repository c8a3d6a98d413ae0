"""CPU tests of the kernels' lane-level code through the host emulation (tests/emul/emul.cpp).

The emulation compiles the SAME source the GPU runs (csrc/demod_core.hpp: agc_step, FllRow8 with fll8_replay /
fll8_tile, rrc_direct8, k2_timing, k2_costas, sincos, pcl_advance) with a 16-lane Row16 standing in for one DPP row,
so the systolic FIR schedule, the delay-line replay and the tile bookkeeping are checked bit-for-bit against the
oracle without a GPU.  It is a test tool, not a product path.
"""
import numpy as np
import pytest


def _u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_host_design_equals_oracle_design(emul, oracle):
    """csrc/design.hpp (product) and oracle/tetra_oracle.c design the same tables, bit for bit."""
    t = emul.design()
    o = oracle.Oracle()
    a, b = o.bandedge_taps()
    assert t.ntaps == 65
    assert np.array_equal(np.ctypeslib.as_array(t.rrc)[:65], o.rrc_taps())
    assert np.array_equal(np.ctypeslib.as_array(t.be_re)[:65], a)
    assert np.array_equal(np.ctypeslib.as_array(t.be_im)[:65], b)
    assert np.array_equal(np.ctypeslib.as_array(t.bank).reshape(128, 8), o.interp_bank())
    for f in ("agc_set_point", "agc_rate", "agc_max_gain", "fll_alpha", "fll_beta", "fll_min_freq", "fll_max_freq"):
        assert getattr(t.k1, f) == getattr(o.tab, f), f
    for f in ("tr_alpha", "tr_beta", "tr_min_freq", "tr_max_freq", "costas_alpha", "costas_beta", "costas_min_freq",
              "costas_max_freq"):
        assert getattr(t.k2, f) == getattr(o.tab, f), f
    assert t.tr_omega == o.tab.tr_omega


@pytest.mark.parametrize("rates", [(18000.0, 50000.0), (18000.0, 40000.0), (20000.0, 36000.0), (17000.0, 36000.0)])
def test_host_design_equals_oracle_design_at_other_rates(emul, oracle, rates):
    """BASELINE config 5 runs the demodulator at 50 ksps (2.78 samples per symbol): another band-edge design (fll.cpp:61-95
    with sps != 2), another RRC, another omega and its limits (complex_fd.cpp:12-28) -- product and oracle tables bit for bit."""
    sym, samp = rates
    cfg = emul.default_cfg()
    cfg.symbolrate, cfg.samplerate = sym, samp
    t = emul.design(cfg)
    oc = oracle.default_cfg()
    oc.symbolrate, oc.samplerate = sym, samp
    o = oracle.Oracle(oc)
    a, b = o.bandedge_taps()
    assert np.array_equal(np.ctypeslib.as_array(t.rrc)[:65], o.rrc_taps())
    assert np.array_equal(np.ctypeslib.as_array(t.be_re)[:65], a) and np.array_equal(np.ctypeslib.as_array(t.be_im)[:65], b)
    for f in ("tr_alpha", "tr_beta", "tr_min_freq", "tr_max_freq"):
        assert getattr(t.k2, f) == getattr(o.tab, f), f
    assert t.tr_omega == o.tab.tr_omega == np.float32(samp / sym)


def test_emulated_kernels_match_oracle_at_50_ksps(emul, oracle, synth):
    """The lane-level code at config 5's rate (a longer timing stride, other filters), two calls with carried state."""
    cfg = emul.default_cfg()
    cfg.samplerate = 50000.0
    oc = oracle.default_cfg()
    oc.samplerate = 50000.0
    N = 9000
    iq, _, _ = synth.gen_channel(N, 4242, sps=50000.0 / 18000.0, cfo=0.02)
    o = oracle.Oracle(oc)
    for lanes in (8, 4, 16):
        o = oracle.Oracle(oc)
        e = emul.EmulDemod(1, cfg=cfg, fll_lanes=lanes)
        for a, b in ((0, 5003), (5003, N)):
            r = o.process(iq[a:b], stages=True)
            q = e.process(iq[a:b], want_sym=True)
            nb = int(q["n_bits"][0])
            assert np.array_equal(_u32(q["y"][0]), _u32(r["y"]))
            assert nb == r["bits"].size and np.array_equal(q["bits"][0][:nb], r["bits"])
            assert np.array_equal(_u32(q["sym"][0][:nb // 2]), _u32(r["sym"]))
        assert (e.st[0].mu, e.st[0].omega, e.st[0].offset) == (o.st.mu, o.st.omega, o.st.offset)


@pytest.mark.parametrize("sym,samp,lim,mug", [(18000.0, 36000.0, 0.02, None), (20000.0, 36000.0, 0.02, None), (18000.0, 50000.0, 0.02, None),
                                              (18000.0, 36000.0, 0.3, None), (30000.0, 36000.0, 0.05, 0.1), (18000.0, 36000.0, 0.0, 0.5)])
def test_output_rows_hold_the_worst_stream(emul, oracle, sym, samp, lim, mug):
    """No silent truncation: the product sizes a row from the slowest step its timing loop can take,
    omega (1 - limit) - |mu_gain|.  The oracle (no row limit, like the reference's stream buffer) run on streams that push the
    loop to its limits -- noise, a tone, a clock that is too fast -- never emits more than the row holds, at every call
    length; and the parameter sets whose symbols would stop advancing are the ones the product refuses."""
    cfg = emul.default_cfg()
    cfg.symbolrate, cfg.samplerate, cfg.omega_rel_limit = sym, samp, lim
    oc = oracle.default_cfg()
    oc.symbolrate, oc.samplerate, oc.omega_rel_limit = sym, samp, lim
    if mug is not None:
        cfg.mu_gain = oc.mu_gain = mug
    rng = np.random.default_rng(7)
    N = 6000
    noise = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
    tone = np.exp(2j * np.pi * 0.23 * np.arange(N)).astype(np.complex64)
    fast = emul_fast_clock(N, samp / sym * (1 - 1.5 * lim - 0.01))
    for x in (noise, tone, fast):
        for n in (1, 2, 7, 180, 1000, N):
            o = oracle.Oracle(oc)
            pos = 0
            while pos + n <= N and pos < 3 * max(n, 700):
                nb = o.process(x[pos:pos + n])["bits"].size
                assert nb <= emul.bits_stride_for(cfg, n), (n, pos, nb)
                pos += n


def emul_fast_clock(N, sps):
    """pi/4-DQPSK whose symbol clock runs at sps samples per symbol (faster than nominal: the loop sits on its lower limit)."""
    from tetra_amd import pkg
    return pkg.synth.gen_channel(N, 99, sps=sps, esn0_db=30.0)[0]


def test_product_refuses_timing_loops_that_can_stall(emul):
    """min_step = omega (1 - limit) - |mu_gain|.  Below 1 the reference emits several symbols from one offset (floor(mu) = 0) and
    so do the kernels since ABI 4 (fused kernel down to 0.27, the generic one below).  At min_step <= 0 the reference's own loop
    may never return: the product says so (TETRA_ERR_UNSUPPORTED from create and from the setters).  Everything above is
    accepted -- there is no other limit on the rates."""
    cfg = emul.default_cfg()
    for sym, samp, lim, ok in ((18000.0, 36000.0, 0.02, True), (20000.0, 36000.0, 0.02, True), (30000.0, 36000.0, 0.1, True),
                               (34000.0, 36000.0, 0.02, True), (36000.0, 36000.0, 0.02, True), (18000.0, 36000.0, 0.5, True),
                               (18000.0, 36000.0, 0.49, True), (18000.0, 100000.0, 0.02, True), (18000.0, 36000.0, 1.0, False),
                               (40000.0, 36000.0, 0.02, True), (18000.0, 36000.0, 0.86, True), (18000.0, 9000.0, 0.02, True),
                               (18000.0, 5000.0, 0.02, True), (18000.0, 36000.0, 0.85, True), (18000.0, 36000.0, 0.995, False),
                               (18000.0, 300.0, 0.02, False)):
        cfg.symbolrate, cfg.samplerate, cfg.omega_rel_limit = sym, samp, lim
        assert (emul.bits_stride_for(cfg, 1000) > 0) == ok, (sym, samp, lim)


@pytest.mark.parametrize("N,chunks", [
    (5000, [5000]),
    (5000, [7] * 100 + [4300]),
    (3000, [1, 2, 3, 15, 16, 17, 33, 180, 1000, 1733]),
    (40060, [40060]),
])
def test_emulated_kernels_match_oracle(emul, oracle, synth, N, chunks):
    assert sum(chunks) == N
    iq, _, _ = synth.gen_channel(N, 3 + N)
    o = oracle.Oracle()
    e = emul.EmulDemod(1)
    pos = 0
    for ch in chunks:
        r = o.process(iq[pos:pos + ch], stages=True)
        q = e.process(iq[pos:pos + ch], want_sym=True)
        pos += ch
        nb = int(q["n_bits"][0])
        assert np.array_equal(_u32(q["y"][0]), _u32(r["y"]))            # RRC output
        assert nb == r["bits"].size and np.array_equal(q["bits"][0][:nb], r["bits"])
        assert np.array_equal(_u32(q["sym"][0][:nb // 2]), _u32(r["sym"]))
    st, os_ = e.st[0], o.st
    assert (st.agc_gain, st.fll_phase, st.fll_freq) == (os_.agc_gain, os_.fll_phase, os_.fll_freq)
    assert (st.mu, st.omega, st.offset) == (os_.mu, os_.omega, os_.offset)
    assert (st.costas_phase, st.costas_freq, st.ph2, st.prev) == (os_.costas_phase, os_.costas_freq, os_.ph2, os_.prev)
    assert np.array_equal(_u32(np.array(st.hist[:], np.float32)[32:]), _u32(np.array(os_.hist[:], np.float32)[-128:]))


@pytest.mark.parametrize("fll_lanes", [8, 4, 16])
def test_emulated_wave_of_64_channels_with_degenerate_inputs(emul, oracle, synth, fll_lanes):
    """64 channels through the emulated stages, including a noise-only and an all-zero channel (their timing
    loops wander, so per-lane offsets diverge inside a tile)."""
    Cn, N = 64, 6000
    iq, _, _ = synth.gen_batch(Cn, N, base_seed=99)
    rng = np.random.default_rng(1)
    iq[5] = (0.1 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))).astype(np.complex64)
    iq[6] = 0
    iq[7] = synth.gen_channel(N, 4, ppm=8000.0)[0]     # 0.8 % clock offset
    bits, nb, sym, _ = oracle.process_batch(iq, want_sym=True)
    q = emul.EmulDemod(Cn, fll_lanes=fll_lanes).process(iq, want_sym=True)
    assert np.array_equal(q["n_bits"], nb)
    for c in range(Cn):
        assert np.array_equal(q["bits"][c][:nb[c]], bits[c][:nb[c]]), c


@pytest.mark.parametrize("fll_lanes", [8, 4, 16])
def test_fused_stage_code_equals_oracle(emul, oracle, synth, fll_lanes, lanes=True):
    """(fll_lanes: the FLL row of the 16-channel workgroup, 8 lanes x 9 taps per channel, and of the 32-channel one, 4 x 17.)
    The fused kernel's building blocks (agc_step, the FLL row with its delay-line replay and 32-sample tiles,
    rrc_direct8, k2_timing, k2_costas -- same source as the device) run stage after stage == the oracle, bit for bit:
    RRC output, bits, symbols; ragged call lengths with carried state (1, 7, 31, 33 ... samples: partial tiles, partial
    8-step groups), 5 channels."""
    Cn = 5
    N = 4200
    iq, _, _ = synth.gen_batch(Cn, N, base_seed=321)
    e = emul.EmulDemod(Cn, fused=lanes, fll_lanes=fll_lanes)
    orc = [oracle.Oracle() for _ in range(Cn)]
    pos = 0
    for n in (1, 7, 31, 33, 64, 17, 1000, 0, 2047, 1000):
        n = min(n, N - pos)
        q = e.process(iq[:, pos:pos + n], want_sym=True)
        for c in range(Cn):
            r = orc[c].process(iq[c, pos:pos + n], stages=True)
            assert np.array_equal(_u32(q["y"][c]), _u32(r["y"])), (lanes, n, c)
            nb = int(q["n_bits"][c])
            assert nb == r["bits"].size and np.array_equal(q["bits"][c][:nb], r["bits"]), (lanes, n, c)
            assert np.array_equal(_u32(q["sym"][c][:nb // 2]), _u32(r["sym"])), (lanes, n, c)
        pos += n


@pytest.mark.parametrize("nt,fll_lanes", [(33, 8), (72, 8), (33, 4), (68, 4), (2, 4), (2, 16), (33, 16), (72, 16),
                                          (73, 16), (100, 16), (129, 16), (73, 8), (100, 8), (129, 8)])
def test_fused_stage_code_other_tap_counts(emul, oracle, synth, nt, fll_lanes, lanes=True):
    """(Beyond 72 taps: the LONG rows -- 16 lanes x 9 taps of the 4-channel workgroup, 8 x 17 of the 16-channel one -- with tables
    of 144 entries and the 128-sample delay line, hist_far + hist.)"""
    N = 2500
    iq, _, _ = synth.gen_channel(N, 77)
    ocfg = oracle.default_cfg()
    ocfg.rrc_tap_count = nt
    o = oracle.Oracle(ocfg)
    ecfg = emul.default_cfg()
    ecfg.rrc_tap_count = nt
    e = emul.EmulDemod(1, ecfg, fused=lanes, fll_lanes=fll_lanes)
    for pos in (0, 1250):
        r = o.process(iq[pos:pos + 1250], stages=True)
        q = e.process(iq[pos:pos + 1250])
        assert np.array_equal(_u32(q["y"][0]), _u32(r["y"])), (lanes, nt)
        nb = int(q["n_bits"][0])
        assert np.array_equal(q["bits"][0][:nb], r["bits"]), (lanes, nt)
    if nt > 72:          # the carried delay line of the long rows: all 128 samples, the oracle's
        st = e.st[0]
        line = np.concatenate([np.array(st.hist_far[:], np.float32), np.array(st.hist[:], np.float32)])
        assert np.array_equal(_u32(line), _u32(np.array(o.st.hist[:], np.float32))) and st.rrc_valid == o.st.rrc_valid == 128


@pytest.mark.parametrize("fll_lanes", [16, 8])
def test_long_rows_ragged_calls_equal_oracle(emul, oracle, synth, fll_lanes):
    """The long rows' C++ form (what the kernels run for the partial tile at the end of every call) over ragged calls with carried
    state, 101 taps, three channels: RRC output, bits, symbols against the oracle, bit for bit."""
    Cn, N = 3, 3000
    iq, _, _ = synth.gen_batch(Cn, N, base_seed=654)
    ocfg = oracle.default_cfg()
    ocfg.rrc_tap_count = 101
    ecfg = emul.default_cfg()
    ecfg.rrc_tap_count = 101
    e = emul.EmulDemod(Cn, ecfg, fused=True, fll_lanes=fll_lanes)
    orc = [oracle.Oracle(ocfg) for _ in range(Cn)]
    pos = 0
    for n in (1, 7, 31, 33, 64, 17, 900, 0, 1000, 947):
        q = e.process(iq[:, pos:pos + n], want_sym=True)
        for c in range(Cn):
            r = orc[c].process(iq[c, pos:pos + n], stages=True)
            assert np.array_equal(_u32(q["y"][c]), _u32(r["y"])), (n, c)
            nb = int(q["n_bits"][c])
            assert nb == r["bits"].size and np.array_equal(q["bits"][c][:nb], r["bits"]), (n, c)
            assert np.array_equal(_u32(q["sym"][c][:nb // 2]), _u32(r["sym"])), (n, c)
        pos += n


def test_quality_distance_is_the_reference_expression_including_signed_zeros(emul):
    """DQPSKSymbolExtractor's per-symbol distance (dqpsk_sym_extr.cpp:8-12): |atan2f(ideal) - atan2f(symbol)| with the ideal
    point chosen by `x < 0` tests.  The kernel's polynomial form equals it to float rounding for symbols inside a quadrant,
    and -- found by profiles/fuzz_parity.py on digitally silent channels -- has to follow atan2f's signed zeros where a
    component is exactly zero: the slicer files -0 under "positive", atan2f(-0, x < 0) is -pi."""
    rng = np.random.default_rng(77)
    z = (rng.standard_normal(20000) + 1j * rng.standard_normal(20000)).astype(np.complex64)
    z[:2000] *= np.float32(1e-20)
    z[2000:4000] *= np.float32(1e15)
    vals = [0.0, -0.0, 0.5, -0.5, 1e-30, -1e-30]
    edge = np.array([complex(a, b) for a in vals for b in vals], np.complex64)
    # (numpy's complex(a, b) keeps the zero signs in both parts)
    z = np.concatenate([z, edge])
    re, im = z.real.astype(np.float32), z.imag.astype(np.float32)
    assert np.signbit(re[-36:]).sum() == 18 and np.signbit(im[-36:]).sum() == 18
    ideal_re = np.where(re < 0, np.float32(-0.7071), np.float32(0.7071))
    ideal_im = np.where(im < 0, np.float32(-0.7071), np.float32(0.7071))
    want = np.abs(np.arctan2(ideal_im, ideal_re).astype(np.float32) - np.arctan2(im, re).astype(np.float32))
    got = emul.quality_distance(z)
    assert np.abs(got - want).max() < 1e-6, (np.abs(got - want).argmax(), z[np.abs(got - want).argmax()])
    # the three sign-of-zero cases are really in the set
    assert (want > 5.0).any() and (np.abs(want - 3.9269908) < 1e-6).any() and (np.abs(want - 2.3561945) < 1e-6).any()


def test_constellation_tap_keeps_the_last_complete_block_of_1024(emul):
    """k_constellation's two phases (constellation_core.hpp) on the host against the definition: the plugin regroups the symbol
    stream into consecutive blocks of 1024 (Reshaper keep 1024 / skip 0, src/main.cpp:88) and shows the last complete one
    (:376-383).  Ragged calls incl. empty ones, calls that complete 0 / 1 / several blocks, exact fits."""
    rng = np.random.default_rng(77)
    for nthr in (256, 64, 1):
        tap = emul.ConstellationTap(nthr)
        stream = np.zeros(0, np.complex64)
        calls = [0, 5, 1019, 0, 1, 1023, 1024, 2048, 3, 4000, 1021, 9000, 0, 7, 500, 524, 18000] + list(rng.integers(0, 3000, 40))
        for n in calls:
            z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
            tap.feed(z)
            stream = np.concatenate([stream, z])
            nb = stream.size // 1024
            assert tap.blocks[0] == nb and tap.fill[0] == stream.size % 1024
            want = stream[(nb - 1) * 1024:nb * 1024] if nb else np.zeros(1024, np.complex64)
            assert np.array_equal(tap.blk.view(np.uint32), want.view(np.uint32)), (nthr, n, nb)
            assert np.array_equal(tap.part[:tap.fill[0]].view(np.uint32), stream[nb * 1024:].view(np.uint32))
