#!/usr/bin/env python3
"""Generator of fll_asm.inc: the instruction stream of the FLL waves of the fused kernel, as gfx950 assembly.

Why assembly: a gfx950 wavefront issues one instruction of any kind per ~4.7 clocks whether or not it depends on the
previous one (profiles/r02/r02_a_issue_model.md), and the FLL wave has a SIMD to itself, so its time IS its instruction
count.  hipcc spends 71.5 instruction slots on one sample step of FllRow8 (s_nop for the packed-math and DPP hazards,
a re-materialised constant, one LDS load and one s_waitcnt per sample, sine and cosine polynomials as ten scalar FMAs);
the schedule below needs 60:
  * the sine / cosine polynomials of the NCO run as ONE packed Horner chain (the same IEEE operations, two per instruction);
  * the fourteen band-edge FMAs of a step that are not on the way to the error (the "middle" taps of the systolic row) are
    issued during the NEXT step, exactly where that step needs an independent instruction between a packed or DPP producer
    and its consumer -- so no s_nop is left;
  * AGC samples come two per LDS load, x goes to the ring once per eight samples, constants sit where the constant-bus
    limit wants them.
demod_core.hpp is the specification: instruction for instruction the same IEEE operations as
FllRow8<float>::step<PH, false, true> (and <PH, true, true> for the delay-line replay), which tests/emul compiles for the
host and checks against the oracle.  kernel_fused.hpp keeps the C++ form for the partial tile at the end of a call.

The block covers a wave's whole steady state: rebuild of the in-flight sums from the last 72 stored samples, then every
COMPLETE 32-sample tile of the call with one s_barrier per tile.

Hazard rules enforced by the emitter (LLVM GCNHazardRecognizer for gfx940/gfx950, cross-checked against hipcc output):
  H1  a VGPR written by a packed-FP32 instruction must not be read by the next instruction (one wait state);
  H2  a VGPR written by a VALU instruction must not be read (or merged into, as `old`) by a DPP instruction within the next
      two instructions;
  H3  VCC written by v_cmp must not be read by v_cndmask within the next two instructions.
A gap is filled from the queue of deferred FMAs, or with an s_nop when that is empty (counted and reported), so a bad
schedule costs slots, never correctness.

Usage: python gen_fll_asm.py [--check]    (writes fll_asm.inc next to this file; --check only verifies it is current)
"""
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

TILE = 32


class Shape:
    """One FLL row geometry: `lanes` positions per channel x `taps` taps per position; the channels of a 16-lane DPP row are
    interleaved, so a hop between neighbouring positions is a shift by 16/lanes lanes.  The schedule's period is taps - 1
    steps (the residents of a position), which must divide the tile."""

    def __init__(self, lanes, taps, out, prefix, what, fold_c12=True, whole_tile_loads=True, be_pad=80):
        self.lanes, self.taps, self.out, self.prefix, self.what = lanes, taps, os.path.join(HERE, out), prefix, what
        self.be_pad = be_pad          # entries per band-edge tap table in LDS (FusedLds::be80): the imaginary taps sit that far behind the real ones
        # the first two Cody-Waite steps as ONE fma with C1 + C2 (the operand %[negc1] then carries -(C1 + C2)): C1 + C2 is a
        # binary32 number, k is -1, 0 or 1 and x - k*C1 is exact (Sterbenz) for every |x| <= pi, so fma(-k, C1 + C2, x) is
        # fma(-k, C2, fma(-k, C1, x)) bit for bit -- checked over all 2 157 060 024 floats of [-pi, pi] in tests/test_oracle.py.
        # On for all three geometries since round 3: with the timing wave on four lanes per channel the FLL stream is what
        # paces every shape, and the slot shows (4096 x 36000: -0.75 %, 8192: -0.5 %, <= 1024 channels: -1.9 %;
        # profiles/r03/r03_l_exp.log).  In round 2 it changed neither launch (profiles/r02/r02_l, r02_q).
        self.fold_c12 = fold_c12 and not os.environ.get("TETRA_EXP_NO_FOLD")      # (the environment switch: experiment builds)
        # The AGC samples of a WHOLE tile are fetched at the top of the tile (16 ds_read_b128 into 64 registers) with two waits,
        # instead of two samples at a time with a wait per load: 14 slots less per tile, and the first sample no longer waits for
        # its load (the 16 issue slots cover the LDS latency).  Costs 56 registers: not for the 4-lane block, whose wave is at
        # 231 of the 256 VGPRs two waves per SIMD can have.
        self.whole_tile_loads = whole_tile_loads and not os.environ.get("TETRA_EXP_NO_TILE_LOADS")
        self.hop = 16 // lanes
        self.nres = taps - 1
        assert TILE % self.nres == 0 and self.nres % 2 == 0
        self.pad = lanes * taps
        self.replay_groups = -(-self.pad // self.nres)       # whole schedule periods covering the padded delay line


SHAPES = {
    # 16-channel workgroups: two FLL waves of 8 channels, 8 lanes per channel (72 = 8 x 9 padded taps)
    "fll": Shape(8, 9, "fll_asm.inc", "FLL_WAVE", "FLL wave"),
    # 32-channel workgroups: two FLL waves of 16 channels, 4 lanes per channel (68 = 4 x 17 padded taps)
    "fll4": Shape(4, 17, "fll4_asm.inc", "FLL4_WAVE", "FLL wave, 4 lanes per channel", whole_tile_loads=False),
    # 4-channel workgroups (at most 4 channels per CU): one FLL wave of 4 channels, a whole DPP row per channel (80 = 16 x 5)
    "fll16": Shape(16, 5, "fll16_asm.inc", "FLL16_WAVE", "FLL wave, 16 lanes per channel"),
    # the LONG 4-channel workgroup: filters of 73 .. 129 taps (144 = 16 x 9 padded taps; tables of 144 entries)
    "fll16l": Shape(16, 9, "fll16l_asm.inc", "FLL16L_WAVE", "FLL wave, 16 lanes per channel, 9 taps per lane", be_pad=144),
    # the LONG 16-channel workgroup: the same filters on rows of 8 lanes per channel (136 = 8 x 17 padded taps), for more than 1024 channels
    "fll8l": Shape(8, 17, "fll8l_asm.inc", "FLL8L_WAVE", "FLL wave, 8 lanes per channel, 17 taps per lane", be_pad=144),
}
G = SHAPES["fll"]
OUT = G.out
TAPS = G.taps       # taps per position; lanes x taps padded taps
NRES = G.nres       # resident sums per position = schedule period


def f32(x):
    """hex literal of the binary32 nearest to x (the constants of sincos_t in demod_core.hpp)."""
    return "0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


class Emitter:
    def __init__(self):
        self.lines = []
        self.n = 0               # instruction slots emitted
        self.last_write = {}     # reg -> (slot index, kind of writer)
        self.vcc_write = -100
        self.nops = 0
        self.counts = {}
        self.pending = []        # deferred independent instructions (the previous step's middle FMAs)

    def _slot(self, text, kind):
        self.lines.append(text)
        self.n += 1
        self.counts[kind] = self.counts.get(kind, 0) + 1

    def label(self, text):
        self.lines.append(text)

    def comment(self, text):
        self.lines.append("; " + text)

    def _need(self, kind, writes, reads, reads_vcc):
        need = 0
        for r in reads:
            w = self.last_write.get(r)
            if w is None:
                continue
            dist = self.n - w[0] - 1          # instructions between the writer and this one
            if w[1] == "pk" and kind in ("valu", "pk", "dpp", "lds"):
                need = max(need, 1 - dist)
            if kind == "dpp" and w[1] in ("valu", "pk", "dpp"):
                need = max(need, 2 - dist)
        if kind == "dpp":                      # the destination is merged into (bound_ctrl:0 keeps `old`)
            for r in writes:
                w = self.last_write.get(r)
                if w is not None and w[1] in ("valu", "pk", "dpp"):
                    need = max(need, 2 - (self.n - w[0] - 1))
        if reads_vcc:
            need = max(need, 2 - (self.n - self.vcc_write - 1))
        return need

    def _emit(self, text, kind, writes, reads, writes_vcc):
        self._slot(text, kind)
        for r in writes:
            self.last_write[r] = (self.n - 1, kind)
        if writes_vcc:
            self.vcc_write = self.n - 1

    def ins(self, text, kind, writes=(), reads=(), reads_vcc=False, writes_vcc=False):
        """kind: valu | pk | dpp | lds | salu | wait | br.  writes/reads: VGPR numbers.  Hazard gaps are filled from the
        queue of deferred instructions first, with s_nop only when it is empty."""
        need = self._need(kind, writes, reads, reads_vcc)
        while need > 0:
            if self.pending and self._need(self.pending[0][1], self.pending[0][2], self.pending[0][3], False) == 0:
                t, k, w, r = self.pending.pop(0)
                self._emit(t, k, w, r, False)
            else:
                self._slot("s_nop 0", "nop")
                self.nops += 1
            need = self._need(kind, writes, reads, reads_vcc)
        self._emit(text, kind, writes, reads, writes_vcc)

    def flush(self, count=None):
        k = len(self.pending) if count is None else min(count, len(self.pending))
        for _ in range(k):
            t, kind, w, r = self.pending.pop(0)
            self.ins(t, kind, w, r)

    def text(self):
        return "\n".join(self.lines)


def pair(r):
    assert r % 2 == 0, r
    return "v[%d:%d]" % (r, r + 1)


def quad(r):
    assert r % 2 == 0, r
    return "v[%d:%d]" % (r, r + 3)


# ----------------------------------------------------------------------------------------------------------------------
# fixed registers of the block
# ----------------------------------------------------------------------------------------------------------------------
def configure(shape):
    """Select the row geometry and lay out the block's fixed registers for it."""
    global G, OUT, TAPS, NRES, R_TA, R_TB, R_R14, R_R32, R_XS, R_PH, R_FR, R_AQ, R_K, R_R, R_Z, R_Q, R_PP, R_SGN, R_A2, R_T1, R_T2
    global R_C14, R_C32, R_D, R_U, R_MX, R_MN, R_E, R_T, R_CC3, R_2PI, R_MAXF, R_AADDR, R_XLANE, R_XROWL, R_TAPADDR, R_HADDR, CLOBBER
    G, OUT, TAPS, NRES = shape, shape.out, shape.taps, shape.nres
    R_TA = 16                       # ta[0..T-1], tb[0..T-1] (slot j <-> padded tap T*(lanes-1-pos)+j)
    R_TB = R_TA + TAPS
    R_R14 = (R_TB + TAPS + 1) & ~1  # r14[i] = v[R_R14+2i : +1], r32[i] likewise, i = 0..NRES-1
    R_R32 = R_R14 + 2 * NRES
    B = R_R32 + 2 * NRES
    R_XS = (B, B + 2)               # x pipeline, alternating by step parity: step s reads XS[s&1] and writes XS[(s+1)&1]
    R_PH, R_FR = B + 4, B + 5
    R_AQ = (B + 6, B + 10)          # AGC samples, two per load: sample s of a tile sits in AQ[(s>>1)&1] + 2*(s&1)
    R_K, R_R = B + 14, B + 15
    R_Z = B + 16                    # (z, -)
    R_Q = B + 18                    # (S3 constant, first cosine Horner value)
    R_PP = B + 20                   # (sine, cosine) Horner pair
    R_SGN = B + 22                  # (-1)^k of the reduction, low half of an aligned pair
    R_A2 = B + 24                   # the AGC sample times that sign
    R_T1, R_T2 = B + 28, B + 30
    R_C14, R_C32 = B + 32, B + 34
    R_D, R_U = B + 36, B + 38
    R_MX, R_MN = B + 40, B + 42
    R_E, R_T = B + 44, B + 45
    R_CC3, R_2PI, R_MAXF = B + 46, B + 47, B + 48      # constants that must sit in vector registers (constant-bus limit)
    R_AADDR, R_XLANE, R_XROWL, R_TAPADDR, R_HADDR = B + 49, B + 50, B + 51, B + 52, B + 53
    global R_ASUM
    R_ASUM = B + 54                 # a_buf[0] row address + a_buf[1] row address of this lane: the other half = sum - this half
    top = B + 55
    if G.whole_tile_loads:          # sample s of the tile in v[R_AS + 2 s : +1]
        global R_AS
        R_AS = (top + 3) & ~3
        top = R_AS + 2 * TILE
    CLOBBER = list(range(16, top))


configure(G)
assert (R_TA, R_TB, R_R14, R_R32, R_XS, R_HADDR) == (16, 25, 34, 50, (66, 68), 119)

# sincos_t constants (demod_core.hpp)
INV_PI_NEG = f32(-0.318309886183790672)
WRAP_C = "0x3e22f983"            # binary32 nearest 1 / (2 pi): see the phase wrap in fir_and_hop
C2N, C3N = -9.67502593994140625e-4, -1.509957990978376432e-7
S3, S2, S1, S0 = 2.597026877992903e-06, -0.0001980524102691561, 0.008332998491823673, -0.16666656732559204
C4, C3c, C2c, C1c = -2.604826931928983e-07, 2.476031113474164e-05, -0.0013888374669477344, 0.04166663810610771
FL_PI = struct.unpack("<f", struct.pack("<f", 3.1415926535))[0]


def pk_consts():
    """The four constant pairs of the packed sine/cosine Horner chain as 64-bit integers (low word = sine side).  The last
    step multiplies the sine polynomial by z: fma(ps, z, -0.0f) is exactly ps*z (also in the sign of a zero product)."""
    def u(x):
        return struct.unpack("<I", struct.pack("<f", x))[0]
    pairs = [(S2, C2c), (S1, C1c), (S0, -0.5), (-0.0, 1.0)]
    return [u(a) | (u(b) << 32) for (a, b) in pairs]




def tap_operand(base, j):
    """(aligned register pair, modifiers) that broadcast tap register base+j to both halves of a packed multiply."""
    r = base + j
    if r % 2 == 0:
        return pair(r), "op_sel_hi:[1,0,1]"
    return pair(r - 1), "op_sel:[0,1,0]"


def r14(i):
    return R_R14 + 2 * (i % NRES)


def r32(i):
    return R_R32 + 2 * (i % NRES)


def fma_op(dst, xs, base, j, acc):
    op, mod = tap_operand(base, j)
    return ("v_pk_fma_f32 %s, %s, %s, %s %s" % (pair(dst), pair(xs), op, pair(acc), mod), "pk", [dst, dst + 1],
            [xs, xs + 1, base + j, acc, acc + 1])


def middle_ops(ph, xs):
    """The seven middle taps of both sums of a step: r[(ph+q) % 8] += xs * t[8-q], q = 1..7, in the order the next step
    needs them (it reads r[(ph+1) % 8] first)."""
    ops = []
    if os.environ.get("TETRA_EXP_FLL_NO_MIDDLE"):      # timing-only experiment (profiles/build_exp.sh): the FIR bulk taken out
        return ops
    for q in range(1, TAPS - 1):
        ops.append(fma_op(r14(ph + q), xs, R_TA, TAPS - 1 - q, r14(ph + q)))
        ops.append(fma_op(r32(ph + q), xs, R_TB, TAPS - 1 - q, r32(ph + q)))
    return ops


def fir_and_hop(E, s, n, replay):
    """Newest tap on the oldest residents, error and loop filter (unless replay), hop inward with zero fill, tap 0 on the
    arrivals.  The middle taps are NOT issued here (see middle_ops)."""
    ph = s % NRES
    E.ins(*fma_op(R_C14, n, R_TA, TAPS - 1, r14(ph)))
    E.ins(*fma_op(R_C32, n, R_TB, TAPS - 1, r32(ph)))
    sh14, sh32 = r14(ph), r32(ph)

    def hop(k):
        src = (R_C14, R_C14 + 1, R_C32, R_C32 + 1)[k]
        dst = (sh14, sh14 + 1, sh32, sh32 + 1)[k]
        E.ins("v_mov_b32_dpp v%d, v%d row_shl:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % (dst, src, G.hop), "dpp", [dst], [src])

    if replay:
        for k in range(4):
            hop(k)
    else:
        # fll_error: d = c14 - swap(c32) = (lbe.re, hbe.im), u = c14 + swap(c32) = (hbe.re, lbe.im)
        E.ins("v_pk_add_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" % (pair(R_D), pair(R_C14), pair(R_C32)),
              "pk", [R_D, R_D + 1], [R_C14, R_C14 + 1, R_C32, R_C32 + 1])
        E.ins("v_pk_add_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,0]" % (pair(R_U), pair(R_C14), pair(R_C32)),
              "pk", [R_U, R_U + 1], [R_C14, R_C14 + 1, R_C32, R_C32 + 1])
        hop(0)
        E.ins("v_max_f32 v%d, |v%d|, |v%d|" % (R_MX, R_U, R_D + 1), "valu", [R_MX], [R_U, R_D + 1])
        E.ins("v_min_f32 v%d, |v%d|, |v%d|" % (R_MN, R_U, R_D + 1), "valu", [R_MN], [R_U, R_D + 1])
        E.ins("v_max_f32 v%d, |v%d|, |v%d|" % (R_MX + 1, R_D, R_U + 1), "valu", [R_MX + 1], [R_D, R_U + 1])
        E.ins("v_min_f32 v%d, |v%d|, |v%d|" % (R_MN + 1, R_D, R_U + 1), "valu", [R_MN + 1], [R_D, R_U + 1])
        E.ins("v_pk_mul_f32 %s, %s, %%[p4] op_sel_hi:[1,0]" % (pair(R_MN), pair(R_MN)), "pk", [R_MN, R_MN + 1], [R_MN, R_MN + 1])
        hop(1)
        E.ins("v_pk_add_f32 %s, %s, %s" % (pair(R_MX), pair(R_MX), pair(R_MN)), "pk", [R_MX, R_MX + 1], [R_MX, R_MX + 1, R_MN, R_MN + 1])
        hop(2)
        E.ins("v_sub_f32 v%d, v%d, v%d" % (R_E, R_MX, R_MX + 1), "valu", [R_E], [R_MX, R_MX + 1])
        # PhaseControlLoop::advance with alpha == 0: freq = clamp(freq + beta*err), phase = wrap(phase + freq)
        E.ins("v_mul_f32 v%d, %%[beta], v%d" % (R_E, R_E), "valu", [R_E], [R_E])
        hop(3)
        E.ins("v_add_f32 v%d, v%d, v%d" % (R_E, R_FR, R_E), "valu", [R_E], [R_FR, R_E])
        E.ins("v_med3_f32 v%d, v%d, %%[minf], v%d" % (R_FR, R_E, R_MAXF), "valu", [R_FR], [R_E, R_MAXF])
        E.ins("v_add_f32 v%d, v%d, v%d" % (R_PH, R_PH, R_FR), "valu", [R_PH], [R_PH, R_FR])
        # the wrap `phase > pi -> phase - 2 pi, phase < -pi -> phase + 2 pi` as w = rint(phase * WRAP_C) in {-1, 0, 1} and
        # phase = fma(-w, 2 pi, phase): three instructions instead of copysign / subtract / compare / select.  WRAP_C is the
        # binary32 nearest 1 / (2 pi); rint(x * WRAP_C) is 1 exactly for the floats above FL_M_PI, -1 below -FL_M_PI, and the
        # fma is the one rounding of the exact x -+ 2 pi like the reference's subtraction: checked for every binary32 x in
        # [-2 pi, 2 pi] (tests/test_oracle.py::test_rint_phase_wrap_is_exact_for_every_phase; the one difference, x = -0 ->
        # +0, cannot occur: a phase that starts at +0 never becomes -0, and tetra_demod_set_state stores -0 as +0).
        E.ins("v_mul_f32 v%d, %s, v%d" % (R_T, WRAP_C, R_PH), "valu", [R_T], [R_PH])
        E.ins("v_rndne_f32 v%d, v%d" % (R_T, R_T), "valu", [R_T], [R_T])
    E.ins(*fma_op(sh14, n, R_TA, 0, sh14))
    E.ins(*fma_op(sh32, n, R_TB, 0, sh32))
    if not replay:
        E.ins("v_fma_f32 v%d, -v%d, v%d, v%d" % (R_PH, R_T, R_2PI, R_PH), "valu", [R_PH], [R_T, R_2PI, R_PH])


def real_step(E, s):
    """Sample step s of a complete tile (FllRow8<float>::step<s & 7, false, true>).  E.pending holds the previous step's
    middle FMAs (they read the OLD pipeline registers, which the next step's x overwrites: all must be out by the end
    of this step, and the two on r[ph] before this step's first FMA)."""
    o, n = R_XS[s & 1], R_XS[(s + 1) & 1]
    a = R_AS + 2 * s if G.whole_tile_loads else R_AQ[(s >> 1) & 1] + 2 * (s & 1)
    had = len(E.pending)
    E.comment("---- step %d" % s)
    # NCO phasor: sincos_t<float, true>(-ph); sine and cosine polynomials as one packed Horner chain.  Every link of that
    # chain (packed result -> next instruction) needs one instruction in between: the instructions that do not belong to the
    # chain -- the sign (-1)^k, the sample times that sign, the wait for the AGC samples -- are issued exactly there, so that
    # the deferred tap FMAs are left for the gaps further down (the 16-lane block has only six of those per step and used to
    # pad with s_nop).
    E.ins("v_mul_f32 v%d, %s, v%d" % (R_K, INV_PI_NEG, R_PH), "valu", [R_K], [R_PH])
    E.ins("v_rndne_f32 v%d, v%d" % (R_K, R_K), "valu", [R_K], [R_K])
    E.ins("v_fma_f32 v%d, v%d, %%[negc1], -v%d" % (R_R, R_K, R_PH), "valu", [R_R], [R_K, R_PH])
    if not G.fold_c12:
        E.ins("v_fmac_f32 v%d, %s, v%d" % (R_R, f32(C2N), R_K), "valu", [R_R], [R_R, R_K])
    E.ins("v_fmac_f32 v%d, %s, v%d" % (R_R, f32(C3N), R_K), "valu", [R_R], [R_R, R_K])
    E.ins("v_mul_f32 v%d, v%d, v%d" % (R_Z, R_R, R_R), "valu", [R_Z], [R_R])
    E.ins("v_fmamk_f32 v%d, v%d, %s, v%d" % (R_Q + 1, R_Z, f32(C4), R_CC3), "valu", [R_Q + 1], [R_Z, R_CC3])        # c4*z + c3
    E.ins("v_pk_fma_f32 %s, %s, %s, %%[k1] op_sel_hi:[1,0,1]" % (pair(R_PP), pair(R_Q), pair(R_Z)), "pk", [R_PP, R_PP + 1],
          [R_Q, R_Q + 1, R_Z])                                                                                        # (s3*z + s2, . *z + c2)
    E.ins("v_fma_f32 v%d, |v%d|, -2.0, 1.0" % (R_SGN, R_K), "valu", [R_SGN], [R_K])                                     # (-1)^k for |k| <= 1
    E.ins("v_pk_fma_f32 %s, %s, %s, %%[k2] op_sel_hi:[1,0,1]" % (pair(R_PP), pair(R_PP), pair(R_Z)), "pk", [R_PP, R_PP + 1],
          [R_PP, R_PP + 1, R_Z])
    if G.whole_tile_loads:
        if s == 0:
            E.ins("s_waitcnt lgkmcnt(%d)" % (TILE // 2 - 1), "wait")      # the first of the tile's 16 loads (LDS returns in order)
        elif s == 2:
            E.ins("s_waitcnt lgkmcnt(0)", "wait")                         # all of them (issued ~130 slots ago: no stall)
    elif s & 1 == 0:
        E.ins("s_waitcnt lgkmcnt(0)", "wait")          # this pair of AGC samples (loaded two steps ago)
    E.ins("v_pk_fma_f32 %s, %s, %s, %%[k3] op_sel_hi:[1,0,1]" % (pair(R_PP), pair(R_PP), pair(R_Z)), "pk", [R_PP, R_PP + 1],
          [R_PP, R_PP + 1, R_Z])
    E.ins("v_pk_mul_f32 %s, %s, %s op_sel_hi:[1,0]" % (pair(R_A2), pair(a), pair(R_SGN)), "pk", [R_A2, R_A2 + 1], [a, a + 1, R_SGN])
    E.ins("v_pk_fma_f32 %s, %s, %s, %%[k4] op_sel_hi:[1,0,1]" % (pair(R_PP), pair(R_PP), pair(R_Z)), "pk", [R_PP, R_PP + 1],
          [R_PP, R_PP + 1, R_Z])                                                                                      # k4: (ps*z - 0, pc*z + 1)
    E.ins("v_fmac_f32 v%d, v%d, v%d" % (R_R, R_PP, R_R), "valu", [R_R], [R_PP, R_R])                                   # sin (before the sign)
    # The sign (-1)^k of sincos_t goes onto the SAMPLE instead of onto sine and cosine: a' = a * sgn with sgn = 1 - 2|k|
    # (the loop keeps |ph| <= pi, so k is -1, 0 or 1; tetra_demod_set_state refuses other phases).  Multiplying by +-1 is
    # exact and commutes with every rounding below, so x has the same bits as a * ((-1)^k c + j (-1)^k s): one slot less
    # than shift + two xors.
    # x = a' * (c + j s): (ar*c, ai*c) + (-(ai*s), ar*s), c = high half of the Horner pair, s = R
    E.ins("v_pk_mul_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,1]" % (pair(R_T1), pair(R_A2), pair(R_PP)), "pk", [R_T1, R_T1 + 1],
          [R_A2, R_A2 + 1, R_PP + 1])
    E.ins("v_pk_mul_f32 %s, %s, %s op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]" % (pair(R_T2), pair(R_A2), pair(R_K)), "pk",
          [R_T2, R_T2 + 1], [R_A2, R_A2 + 1, R_R])
    E.ins("v_pk_add_f32 %s, %s, %s" % (pair(n), pair(R_T1), pair(R_T2)), "pk", [n, n + 1], [R_T1, R_T1 + 1, R_T2, R_T2 + 1])
    if not G.whole_tile_loads and s & 1 == 1 and s + 3 < TILE:
        # both samples of this pair are consumed: the pair after the next one goes into their registers
        nq = R_AQ[(s >> 1) & 1]
        E.ins("ds_read_b128 %s, v%d offset:%d" % (quad(nq), R_AADDR, 8 * (s + 3)), "lds", list(range(nq, nq + 4)), [R_AADDR])
    E.ins("v_mov_b32_dpp v%d, v%d row_shr:%d row_mask:0xf bank_mask:0xf" % (n, o, G.hop), "dpp", [n], [o])
    E.ins("v_mov_b32_dpp v%d, v%d row_shr:%d row_mask:0xf bank_mask:0xf" % (n + 1, o + 1, G.hop), "dpp", [n + 1], [o + 1])
    used = had - len(E.pending)
    if had and used < 2:
        E.flush(2 - used)           # the deferred FMAs on r[ph] must precede this step's FMAs on it
    fir_and_hop(E, s, n, False)
    E.flush()                       # what is left of the previous step's middle FMAs
    if s % G.lanes == G.lanes - 1:
        # lane (pos) holds x_{s-pos}: `lanes` samples of the tile to the ring
        E.ins("ds_write_b64 v%d, %s offset:%d" % (R_XLANE, pair(n), 8 * s), "lds", [], [R_XLANE, n, n + 1])
    E.pending = middle_ops(s % NRES, n)


def replay_step(E, g, n_reg, o_reg):
    """Replay step g (0..NRES-1) of a group: x is a stored sample (no NCO, no loop update)."""
    E.comment("---- replay step %d" % g)
    E.ins("ds_read_b64 %s, v%d offset:%d" % (pair(n_reg), R_HADDR, 8 * g), "lds", [n_reg, n_reg + 1], [R_HADDR])
    E.flush()                       # the previous step's middle FMAs (they read o_reg)
    E.ins("s_waitcnt lgkmcnt(0)", "wait")
    E.ins("v_mov_b32_dpp v%d, v%d row_shr:%d row_mask:0xf bank_mask:0xf" % (n_reg, o_reg, G.hop), "dpp", [n_reg], [o_reg])
    E.ins("v_mov_b32_dpp v%d, v%d row_shr:%d row_mask:0xf bank_mask:0xf" % (n_reg + 1, o_reg + 1, G.hop), "dpp", [n_reg + 1], [o_reg + 1])
    fir_and_hop(E, g, n_reg, True)
    E.pending = middle_ops(g % NRES, n_reg)


def gen():
    E = Emitter()
    if os.environ.get("TETRA_EXP_SETPRIO"):          # experiment builds: the FLL wave at a raised issue / fetch priority
        E.ins("s_setprio %d" % int(os.environ["TETRA_EXP_SETPRIO"]), "salu")
    E.comment("state, constants and addresses into the block's fixed registers; taps from LDS; sums and pipeline start at zero")
    for (reg, opnd) in ((R_PH, "ph"), (R_FR, "fr"), (R_AADDR, "a_addr"), (R_XROWL, "x_rowlane"), (R_TAPADDR, "tap_addr"),
                        (R_HADDR, "hist_addr"), (R_MAXF, "maxf"), (R_ASUM, "a_sum")):
        E.ins("v_mov_b32 v%d, %%[%s]" % (reg, opnd), "valu", [reg])
    E.ins("v_mov_b32 v%d, %s" % (R_Q, f32(S3)), "valu", [R_Q])
    E.ins("v_mov_b32 v%d, %s" % (R_CC3, f32(C3c)), "valu", [R_CC3])
    E.ins("v_mov_b32 v%d, %s" % (R_2PI, f32(FL_PI - (-FL_PI))), "valu", [R_2PI])
    for j in range(TAPS):
        E.ins("ds_read_b32 v%d, v%d offset:%d" % (R_TA + j, R_TAPADDR, 4 * j), "lds", [R_TA + j])
        E.ins("ds_read_b32 v%d, v%d offset:%d" % (R_TB + j, R_TAPADDR, 4 * G.be_pad + 4 * j), "lds", [R_TB + j])
    for i in range(NRES):
        E.ins("v_mov_b64 %s, 0" % pair(R_R14 + 2 * i), "valu", [R_R14 + 2 * i, R_R14 + 2 * i + 1])
        E.ins("v_mov_b64 %s, 0" % pair(R_R32 + 2 * i), "valu", [R_R32 + 2 * i, R_R32 + 2 * i + 1])
    E.ins("v_mov_b64 %s, 0" % pair(R_XS[0]), "valu", [R_XS[0], R_XS[0] + 1])
    E.ins("v_mov_b64 %s, 0" % pair(R_XS[1]), "valu", [R_XS[1], R_XS[1] + 1])
    E.ins("s_waitcnt lgkmcnt(0)", "wait")
    # ---- rebuild the in-flight sums: replay of the last 72 stored samples, nine groups of eight steps.  The deferred
    # middle FMAs carry over the loop's back edge (and into the first real step); the very first batch meets an all-zero
    # pipeline and all-zero sums, where fma(0, t, 0) changes nothing.
    E.pending = middle_ops(NRES - 1, R_XS[0])
    at_top = [p[0] for p in E.pending]
    E.ins("s_mov_b32 %%[st], %d" % G.replay_groups, "salu")
    E.label(".Lreplay_%=:")
    for g in range(NRES):
        replay_step(E, g, R_XS[(g + 1) & 1], R_XS[g & 1])
    assert [p[0] for p in E.pending] == at_top
    if 8 * NRES <= 64:
        E.ins("v_add_u32 v%d, %d, v%d" % (R_HADDR, 8 * NRES, R_HADDR), "valu", [R_HADDR], [R_HADDR])
    else:
        E.ins("v_add_u32 v%d, 0x%x, v%d" % (R_HADDR, 8 * NRES, R_HADDR), "valu", [R_HADDR], [R_HADDR])
    E.ins("s_sub_u32 %[st], %[st], 1", "salu")
    E.ins("s_cmp_lg_u32 %[st], 0", "salu")
    E.ins("s_cbranch_scc1 .Lreplay_%=", "br")
    # ---- the tiles
    if os.environ.get("TETRA_EXP_ALIGN"):          # experiment builds: the tile loop's head on a 2^k-byte boundary
        E.label(".p2align %d" % int(os.environ["TETRA_EXP_ALIGN"]))
    E.label(".Ltile_%=:")
    # x ring address of this lane for the tile: row + 8*(8 + (base & 255) - pos) (front padding of 8 slots, see FusedLds)
    E.ins("s_and_b32 %%[st], %%[base], 0x%x" % (int(os.environ.get("TETRA_EXP_XRING", "256")) - 1), "salu")
    E.ins("s_lshl_b32 %[st], %[st], 3", "salu")
    E.ins("v_add_u32 v%d, %%[st], v%d" % (R_XLANE, R_XROWL), "valu", [R_XLANE], [R_XROWL])
    if G.whole_tile_loads:
        loads_at = E.n
        for q in range(TILE // 2):
            r0 = R_AS + 4 * q
            E.ins("ds_read_b128 %s, v%d offset:%d" % (quad(r0), R_AADDR, 16 * q), "lds", list(range(r0, r0 + 4)), [R_AADDR])
    else:
        loads_at = None
        E.ins("ds_read_b128 %s, v%d" % (quad(R_AQ[0]), R_AADDR), "lds", list(range(R_AQ[0], R_AQ[0] + 4)), [R_AADDR])
        E.ins("ds_read_b128 %s, v%d offset:16" % (quad(R_AQ[1]), R_AADDR), "lds", list(range(R_AQ[1], R_AQ[1] + 4)), [R_AADDR])
    prologue = E.n if loads_at is None else loads_at          # (the per-tile count includes the tile's sample loads either way)
    for s in range(TILE):
        real_step(E, s)
    per_tile = E.n - prologue
    assert [p[0] for p in E.pending] == at_top, "deferred FMAs must line up across the loop's back edge"
    E.ins("v_sub_u32 v%d, v%d, v%d" % (R_AADDR, R_ASUM, R_AADDR), "valu", [R_AADDR], [R_ASUM, R_AADDR])   # a_buf[0] <-> a_buf[1]
    E.ins("s_add_u32 %[base], %[base], 32", "salu")
    E.ins("s_sub_u32 %[tiles], %[tiles], 1", "salu")
    E.ins("s_waitcnt lgkmcnt(0)", "wait")
    E.ins("s_barrier", "salu")
    E.ins("s_cmp_lg_u32 %[tiles], 0", "salu")
    E.ins("s_cbranch_scc1 .Ltile_%=", "br")
    for (reg, opnd) in ((R_PH, "ph"), (R_FR, "fr")):
        E.ins("v_mov_b32 %%[%s], v%d" % (opnd, reg), "valu")
    E.pending = []
    return E, per_tile


def c_string(text):
    out = []
    for line in text.split("\n"):
        if not line.strip():
            continue
        if line.startswith(";"):
            out.append("    /* %s */" % line[1:].strip())
        else:
            out.append('    "%s\\n"' % line.replace("\\", "\\\\").replace('"', '\\"'))
    return "\n".join(out)


def generate(shape=None):
    configure(shape or SHAPES["fll"])
    E, per_tile = gen()
    P = G.prefix
    parts = []
    parts.append("// %s -- GENERATED by gen_fll_asm.py; do not edit.  See that file for the schedule and the hazard rules.\n" % os.path.basename(G.out))
    parts.append("// %s: %d instruction slots per 32-sample tile (%.2f per sample), %d s_nop in the whole block\n" % (G.what, per_tile, per_tile / 32.0, E.nops))
    parts.append("#define %s_ASM \\\n" % P + c_string(E.text()).replace("\n", " \\\n") + "\n")
    parts.append("#define %s_CLOBBERS %s\n" % (P, ", ".join('"v%d"' % r for r in CLOBBER)))
    k = pk_consts()
    parts.append("#define %s_K1 0x%016xull\n#define %s_K2 0x%016xull\n#define %s_K3 0x%016xull\n#define %s_K4 0x%016xull\n" % (P, k[0], P, k[1], P, k[2], P, k[3]))
    parts.append("#define %s_SLOTS_PER_TILE %d\n" % (P, per_tile))
    # the block's %[negc1] operand: -C1 of the phasor's Cody-Waite reduction, or -(C1 + C2) when the first two steps are folded
    parts.append("#define %s_NEGC1 %s\n" % (P, "(-(3.140625f + 9.67502593994140625e-4f))" if G.fold_c12 else "(-3.140625f)"))
    return "".join(parts), E, per_tile


def main():
    rc = 0
    for name in ("fll", "fll4", "fll16", "fll16l", "fll8l"):
        text, E, per_tile = generate(SHAPES[name])
        out = SHAPES[name].out
        if "--check" in sys.argv:
            cur = open(out).read() if os.path.exists(out) else ""
            if cur != text:
                print("%s is stale: run gen_fll_asm.py" % os.path.basename(out))
                rc = 1
            continue
        with open(out, "w") as f:
            f.write(text)
        print("%s: %d slots / tile = %.2f per sample; block: nops %d, %s" % (SHAPES[name].what, per_tile, per_tile / 32.0, E.nops, dict(sorted(E.counts.items()))))
    configure(SHAPES["fll"])
    return rc


if __name__ == "__main__":
    sys.exit(main())
