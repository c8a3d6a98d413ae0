"""The generated FLL assembly (csrc/fll_asm.inc: 8 lanes per channel, fll4_asm.inc: 4 lanes per channel) EXECUTED on the CPU by
the small gfx950 interpreter of tests/gcn_sim.py and compared with the oracle: the derotated samples x the block writes to
its ring and the loop state it hands back are the oracle's, bit for bit.  This checks the emitted text itself -- operand
selection of the packed instructions, DPP hops, register allocation, deferred FMAs across step and loop boundaries, LDS
addressing, the replay of the delay line -- without a GPU (the GPU parity tests then check the same text on the hardware).

The AGC is neutralised (rate 0: gain stays exactly 1, so the FLL's input is the raw IQ) because the block starts behind it."""
import os
import re

import numpy as np
import pytest

from tests import gcn_sim

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdrpp-tetra-demodulator_amd", "csrc")
KFX, KFXP, KHIST, TILE = 256, 8, 80, 32          # kernel_fused.hpp: x ring, its front padding; demod_core.hpp: delay line


def _u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _run(oracle, emul, synth, fname, macro, lanes, taps, mutate=None, ntaps=65, warm=0):
    lines, consts = gcn_sim.parse_block(os.path.join(CSRC, fname), macro)
    folded = "9.67502593994140625e-4f" in re.search(r"#define %s_NEGC1 (.*)" % macro, open(os.path.join(CSRC, fname)).read()).group(1)
    if mutate:
        lines = mutate(lines)
    hop, nch = 16 // lanes, 64 // lanes
    # the long rows (16 x 9 = 144 padded taps, filters of 73 .. 129 taps): tap tables of 144 entries, 128 delay-line samples
    BEP, KH = (144, 128) if lanes * taps > 80 else (80, KHIST)
    nres = taps - 1
    group = max(nres, lanes)
    replay = -(-(lanes * taps) // group) * group
    ntiles = 6
    N = ntiles * TILE
    # input: TETRA channels with carrier offsets, so that the loop moves; AGC neutral
    # (warm > 0: the call under test is the SECOND one -- the delay line in the ring and the loop state come from a first
    # call of `warm` samples through the oracle, so the block's replay rebuilds its in-flight sums from real history)
    iq_all = np.stack([synth.gen_channel(warm + N, 900 + c, cfo=0.04 * ((c % 5) - 2))[0] for c in range(nch)]).astype(np.complex64)
    iq = np.ascontiguousarray(iq_all[:, warm:])
    ocfg = oracle.default_cfg()
    ocfg.agc_rate = 0.0
    ocfg.rrc_tap_count = ntaps
    otab = oracle.Oracle(ocfg).tab          # (the tables: the oracle's design; the emulation's are capped at 80 taps)
    want_x, want_state, hist, start = [], [], [], []
    for c in range(nch):
        o = oracle.Oracle(ocfg)
        if warm:
            r0 = o.process(iq_all[c, :warm], stages=True)
            hist.append(r0["x"][-KH:])
            start.append((o.st.fll_phase, o.st.fll_freq))
        r = o.process(iq[c], stages=True)
        want_x.append(r["x"])
        want_state.append((o.st.fll_phase, o.st.fll_freq))
    # LDS image: [a_buf[2][nch][32] float2] [tap table re[80] | im[80]] [x_ring[nch][8 + 256 + 1] float2]; a_buf first, like in
    # FusedLds: the block flips between its halves with an XOR of the address
    nt = int(otab.ntaps_be)
    be = np.zeros((2, BEP), np.float32)
    be[0, BEP - nt:] = np.array(otab.be_a[:nt], np.float32)
    be[1, BEP - nt:] = np.array(otab.be_b[:nt], np.float32)
    AROW = (TILE + 2) * 8           # kernel_fused.hpp kFAS: rows of the AGC output buffer are padded by two samples
    a_bytes = 2 * nch * AROW
    off_a, off_be = 0, a_bytes
    off_x = off_be + 2 * BEP * 4
    row = (KFXP + KFX + 1) * 8
    lds = np.zeros(off_x + nch * row + 64, np.uint8)
    lds[off_be:off_be + 2 * BEP * 4] = be.view(np.uint8).reshape(-1)

    def put_tile(t):
        if t >= ntiles:
            return
        base = off_a + (t & 1) * nch * AROW
        blk = np.ascontiguousarray(iq[:, t * TILE:(t + 1) * TILE])          # [nch][32] complex64
        for c in range(nch):
            lds[base + c * AROW:base + c * AROW + TILE * 8] = blk[c].view(np.uint8)

    put_tile(0)
    if warm:
        assert warm >= KH
        for c in range(nch):          # the last 80 (128) outputs of the first call sit at the end of the ring (kernel_fused.hpp prologue)
            a = off_x + c * row + (KFXP + KFX - KH) * 8
            lds[a:a + KH * 8] = np.ascontiguousarray(hist[c]).view(np.uint8)
    lane = np.arange(64)
    pos = (lane & 15) // hop
    ch = (lane >> 4) * hop + lane % hop
    tap_off = BEP - lanes * taps
    vec = {
        "ph": np.array([np.float32(start[c][0]).view(np.uint32) for c in ch], np.uint32) if warm else np.zeros(64, np.uint32),
        "fr": np.array([np.float32(start[c][1]).view(np.uint32) for c in ch], np.uint32) if warm else np.zeros(64, np.uint32),
        "a_addr": (off_a + ch * AROW).astype(np.uint32),
        "a_sum": (2 * (off_a + ch * AROW) + nch * AROW).astype(np.uint32),
        "x_rowlane": (off_x + ch * row + KFXP * 8 - 8 * pos).astype(np.uint32),
        "tap_addr": (off_be + 4 * (tap_off + taps * (lanes - 1 - pos))).astype(np.uint32),
        "hist_addr": (off_x + ch * row + (KFXP + KFX - replay) * 8).astype(np.uint32),
        "maxf": np.full(64, np.float32(otab.fll_max_freq).view(np.uint32), np.uint32),
    }
    f32 = lambda x: int(np.float32(x).view(np.uint32))
    sca = {
        # (a block generated with the first two Cody-Waite steps folded into one fma takes -(C1 + C2) here: <macro>_NEGC1)
        "negc1": f32(-(np.float32(3.140625) + np.float32(9.67502593994140625e-4))) if folded else f32(-3.140625), "beta": f32(otab.fll_beta), "minf": f32(otab.fll_min_freq),
        "p4": (f32(0.4), 0), "base": 0, "tiles": ntiles, "st": 0,
        "k1": consts["K1"], "k2": consts["K2"], "k3": consts["K3"], "k4": consts["K4"],
    }
    sim = gcn_sim.Sim(lines, vec, sca, lds, on_barrier=lambda s, k: put_tile(k))
    sim.run()
    assert sim.barriers == ntiles
    ring = lds[off_x:off_x + nch * row].view(np.float32).reshape(nch, KFXP + KFX + 1, 2)
    bad = []
    for c in range(nch):
        got = ring[c, KFXP:KFXP + N]                      # slot i of the ring = sample i (N <= 256: no wrap yet)
        head = 16 * (c // hop) + c % hop                  # the channel's head lane
        if not (np.array_equal(_u32(got.reshape(-1)), _u32(want_x[c])) and
                int(sim.vec["ph"][head]) == int(np.float32(want_state[c][0]).view(np.uint32)) and
                int(sim.vec["fr"][head]) == int(np.float32(want_state[c][1]).view(np.uint32))):
            bad.append(c)
    # the loops really moved: otherwise the comparison above would be vacuous
    assert max(abs(s[1]) for s in want_state) > 1e-4
    return bad


GEOMETRIES = [("fll_asm.inc", "FLL_WAVE", 8, 9), ("fll4_asm.inc", "FLL4_WAVE", 4, 17), ("fll16_asm.inc", "FLL16_WAVE", 16, 5),
              ("fll16l_asm.inc", "FLL16L_WAVE", 16, 9), ("fll8l_asm.inc", "FLL8L_WAVE", 8, 17)]


@pytest.mark.parametrize("fname,macro,lanes,taps", GEOMETRIES)
def test_generated_fll_assembly_executes_to_the_oracle_s_samples(oracle, emul, synth, fname, macro, lanes, taps):
    assert _run(oracle, emul, synth, fname, macro, lanes, taps) == []


@pytest.mark.parametrize("fname,macro,lanes,taps", GEOMETRIES)
def test_generated_fll_assembly_second_call_replays_the_delay_line(oracle, emul, synth, fname, macro, lanes, taps):
    assert _run(oracle, emul, synth, fname, macro, lanes, taps, warm=150) == []


@pytest.mark.parametrize("fname,macro,lanes,taps,ntaps", [GEOMETRIES[0] + (2,), GEOMETRIES[0] + (33,), GEOMETRIES[0] + (72,),
                                                          GEOMETRIES[1] + (2,), GEOMETRIES[1] + (33,), GEOMETRIES[1] + (68,),
                                                          GEOMETRIES[2] + (2,), GEOMETRIES[2] + (33,), GEOMETRIES[2] + (72,),
                                                          GEOMETRIES[3] + (73,), GEOMETRIES[3] + (100,), GEOMETRIES[3] + (129,),
                                                          GEOMETRIES[4] + (73,), GEOMETRIES[4] + (100,), GEOMETRIES[4] + (129,)])
def test_generated_fll_assembly_other_tap_counts(oracle, emul, synth, fname, macro, lanes, taps, ntaps):
    """Band-edge filters shorter than the row (zero-padded at the old end) and as long as the row holds."""
    assert _run(oracle, emul, synth, fname, macro, lanes, taps, ntaps=ntaps) == []


@pytest.mark.parametrize("fname,macro,lanes,taps", GEOMETRIES)
def test_the_execution_check_sees_planted_faults(oracle, emul, synth, fname, macro, lanes, taps):
    """The comparison is not vacuous: one wrong operand selection in one packed FMA of the tile loop, or partial sums that are
    not zero-filled at the tail of the row, and every channel's output differs."""
    def one_op_sel(lines):
        k = next(i for i, ln in enumerate(lines) if ln.startswith(".Ltile"))
        j = next(i for i in range(k, len(lines)) if lines[i].startswith("v_pk_fma_f32") and "op_sel_hi:[1,0,1]" in lines[i] and "%[" not in lines[i])
        return lines[:j] + [lines[j].replace("op_sel_hi:[1,0,1]", "op_sel_hi:[1,1,1]")] + lines[j + 1:]

    def no_zero_fill(lines):
        return [ln.replace(" bound_ctrl:1", "") for ln in lines]
    def wrong_x_hop(lines):          # the x pipeline moved two positions instead of one
        hop = 16 // lanes
        return [ln.replace("row_shr:%d " % hop, "row_shr:%d " % (2 * hop)) if "row_shr" in ln else ln for ln in lines]
    nch = 64 // lanes
    assert len(_run(oracle, emul, synth, fname, macro, lanes, taps, mutate=one_op_sel)) >= nch // 2
    assert len(_run(oracle, emul, synth, fname, macro, lanes, taps, mutate=wrong_x_hop)) == nch
    if lanes * taps - 65 < taps:
        # (with 16 x 5 = 80 padded taps the 15 zero taps in front of a 65-tap filter cover the three outermost positions: the
        # sums born there stay +0 with or without the zero fill, so this fault cannot show at that geometry)
        assert len(_run(oracle, emul, synth, fname, macro, lanes, taps, mutate=no_zero_fill)) == nch
