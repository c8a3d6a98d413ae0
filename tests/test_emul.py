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


@pytest.mark.parametrize("fll_lanes", [8, 4])
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


@pytest.mark.parametrize("fll_lanes", [8, 4])
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


@pytest.mark.parametrize("nt,fll_lanes", [(33, 8), (72, 8), (33, 4), (68, 4), (2, 4)])
def test_fused_stage_code_other_tap_counts(emul, oracle, synth, nt, fll_lanes, lanes=True):
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
