#!/usr/bin/env python3
"""Generator of fll_asm.inc: the two instruction streams that set the pace of the fused kernel, as gfx950 assembly.

Why assembly: a gfx950 wavefront issues one instruction of any kind per ~4.7 clocks (profiles/r02/r02_a_issue_model.md), so
the FLL loop wave's time IS its instruction count.  hipcc spends 84 instruction slots on the loop wave's sample step
(scalar bookkeeping, s_nop for the packed-math and DPP hazards, re-materialised constants, one s_waitcnt per LDS load);
the schedule below needs 55, every hazard gap filled with an instruction that has to be issued anyway.

Two streams are generated (demod_core.hpp is the specification: instruction for instruction the same IEEE operations as
FllNear8<float>::step<false, true> and FllFar4<float, 17>::step, which tests/emul compiles for the host and checks against
the oracle; kernel_fused.hpp keeps the C++ form for the partial tile at the end of a call):

  FLL_LOOP_ASM    a loop wave (8 channels): all COMPLETE 32-sample tiles of a call, one s_barrier per tile.  Per sample:
                  NCO sincos, complex multiply, x shift, the 16 newest band-edge taps (8 positions x 2), band-edge error,
                  loop filter, x to the ring, progress counter.
  FLL_HELPER_ASM  the helper wave (16 channels): its whole life -- pipeline rebuild from the stored delay line, then every
                  tile of the call (68 far taps on 4 positions per channel), barriers included.

Hazard rules enforced by the emitter (LLVM GCNHazardRecognizer for gfx940/gfx950, cross-checked against hipcc output):
  H1  a VGPR written by a packed-FP32 instruction must not be read by the next instruction (one wait state);
  H2  a VGPR written by a VALU instruction must not be read (or merged into, as `old`) by a DPP instruction within the next
      two instructions;
  H3  VCC written by v_cmp must not be read by v_cndmask within the next two instructions.
A violated rule gets an s_nop (counted and reported), so a bad schedule costs slots, never correctness.

Usage: python gen_fll_asm.py [--check]    (writes fll_asm.inc next to this file; --check only verifies it is current)
"""
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "fll_asm.inc")

TILE = 32
SPIN_LIMIT = 0x40000
# timing experiments only (results invalid): FLL_ASM_ABLATE=L / H / LH drops the loop waves' / the helper's waits
ABLATE = os.environ.get("FLL_ASM_ABLATE", "")


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
        self.mark = None

    def _slot(self, text, kind):
        self.lines.append(text)
        self.n += 1
        self.counts[kind] = self.counts.get(kind, 0) + 1

    def label(self, text):
        self.lines.append(text)

    def comment(self, text):
        self.lines.append("; " + text)

    def nop(self, count=1):
        self._slot("s_nop %d" % (count - 1), "nop")
        self.nops += 1

    def ins(self, text, kind, writes=(), reads=(), reads_vcc=False, writes_vcc=False):
        """kind: valu | pk | dpp | lds | salu | wait | br.  writes/reads: VGPR numbers."""
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
        if need > 0:
            self.nop(need)
        self._slot(text, kind)
        for r in writes:
            self.last_write[r] = (self.n - 1, kind)
        if writes_vcc:
            self.vcc_write = self.n - 1

    def text(self):
        return "\n".join(self.lines)


def pair(r):
    assert r % 2 == 0, r
    return "v[%d:%d]" % (r, r + 1)


def quad(r):
    assert r % 2 == 0, r
    return "v[%d:%d]" % (r, r + 3)


def spin(E, tag, flag_addr, tmp, need_expr_ins, stuck_addr, back):
    """Slow path of a hand-over wait: spin until `cmp` clears, with the watchdog of kernel_fused.hpp's ho_wait."""
    E.label(".L%s_%%=:" % tag)
    E.ins("s_mov_b32 %[spins], 0", "salu")
    E.label(".L%s_spin_%%=:" % tag)
    E.ins("ds_read_b32 v%d, v%d" % (tmp, flag_addr), "lds")
    E.ins("s_waitcnt lgkmcnt(0)", "wait")
    for t in need_expr_ins(tmp):
        E.ins(t, "valu")
    E.ins("s_cbranch_vccz .L%s_%%=" % back, "br")
    E.ins("s_add_u32 %[spins], %[spins], 1", "salu")
    E.ins("s_cmp_lt_u32 %%[spins], 0x%x" % SPIN_LIMIT, "salu")
    E.ins("s_cbranch_scc0 .L%s_giveup_%%=" % tag, "br")
    E.ins("ds_read_b32 v%d, v%d" % (tmp, stuck_addr), "lds")
    E.ins("s_waitcnt lgkmcnt(0)", "wait")
    E.ins("v_cmp_eq_u32 vcc, 0, v%d" % tmp, "valu")
    E.ins("s_cbranch_vccnz .L%s_spin_%%=" % tag, "br")
    E.label(".L%s_giveup_%%=:" % tag)
    E.ins("v_mov_b32 v%d, 1" % tmp, "valu")
    E.ins("ds_write_b32 v%d, v%d" % (stuck_addr, tmp), "lds")
    E.ins("s_branch .L%s_%%=" % back, "br")


# ----------------------------------------------------------------------------------------------------------------------
# FLL loop wave
# ----------------------------------------------------------------------------------------------------------------------
# fixed registers of the loop-wave block
L_XS = (16, 18)       # x pipeline, alternating by step parity: step s reads L_XS[s&1] and writes L_XS[(s+1)&1]
L_R14, L_R32 = 20, 22
L_PH, L_FR = 24, 25
L_TA, L_TB = 26, 28   # (ta0, ta1), (tb0, tb1)
L_A = (30, 32)        # AGC output of the step, by parity
L_F = (36, 40)        # far sums of the step (f14 = +0, f32 = +2), by parity (quad aligned)
L_K, L_R, L_Z, L_PS, L_PC, L_M = 44, 45, 46, 47, 34, 35
L_CC, L_SS = 48, 50   # (cos, -), (sin, -): the phasor in the low halves of two aligned pairs
L_T1, L_T2 = 52, 54
L_C14, L_C32 = 56, 58
L_D, L_U = 60, 62
L_MX, L_MN = 64, 66
L_E, L_T = 68, 69
L_FLAG, L_NEED, L_TWO = 70, 71, 72
L_CS2, L_CC3, L_2PI, L_MAXF = 73, 74, 75, 76     # constants that must sit in vector registers (constant-bus limit)
L_AADDR, L_FADDR, L_XADDR, L_XDADDR, L_FDADDR, L_STUCK = 77, 78, 79, 80, 81, 82
L_XBASE, L_HEADMASK = 83, 84
L_CLOBBER = list(range(16, 85))

# sincos_t constants (demod_core.hpp)
INV_PI_NEG = f32(-0.318309886183790672)
C1 = 3.140625
C2N, C3N = -9.67502593994140625e-4, -1.509957990978376432e-7
S3, S2, S1, S0 = 2.597026877992903e-06, -0.0001980524102691561, 0.008332998491823673, -0.16666656732559204
C4, C3c, C2c, C1c = -2.604826931928983e-07, 2.476031113474164e-05, -0.0013888374669477344, 0.04166663810610771
FL_PI = struct.unpack("<f", struct.pack("<f", 3.1415926535))[0]

A_BUF_TOGGLE = 16 * TILE * 8     # bytes between a_buf[0] and a_buf[1] (kFCh x kFT float2)


def loop_step(E, s):
    """Sample step s of a complete tile.  %[..] operands are bound in kernel_fused.hpp."""
    o, n = L_XS[s & 1], L_XS[(s + 1) & 1]          # old / new x pipeline registers
    a, F = L_A[s & 1], L_F[s & 1]
    an, Fn = L_A[(s + 1) & 1], L_F[(s + 1) & 1]
    f14, f32_ = F, F + 2
    E.comment("---- step %d" % s)
    # NCO phasor: sincos_t<float, true>(-ph)
    E.ins("v_mul_f32 v%d, %s, v%d" % (L_K, INV_PI_NEG, L_PH), "valu", [L_K], [L_PH])
    E.ins("v_rndne_f32 v%d, v%d" % (L_K, L_K), "valu", [L_K], [L_K])
    E.ins("v_fma_f32 v%d, v%d, %%[negc1], -v%d" % (L_R, L_K, L_PH), "valu", [L_R], [L_K, L_PH])
    E.ins("v_fmac_f32 v%d, %s, v%d" % (L_R, f32(C2N), L_K), "valu", [L_R], [L_R, L_K])
    E.ins("v_fmac_f32 v%d, %s, v%d" % (L_R, f32(C3N), L_K), "valu", [L_R], [L_R, L_K])
    E.ins("v_mul_f32 v%d, v%d, v%d" % (L_Z, L_R, L_R), "valu", [L_Z], [L_R])
    E.ins("v_fmamk_f32 v%d, v%d, %s, v%d" % (L_PS, L_Z, f32(S3), L_CS2), "valu", [L_PS], [L_Z, L_CS2])
    E.ins("v_fmamk_f32 v%d, v%d, %s, v%d" % (L_PC, L_Z, f32(C4), L_CC3), "valu", [L_PC], [L_Z, L_CC3])
    E.ins("v_fmaak_f32 v%d, v%d, v%d, %s" % (L_PS, L_PS, L_Z, f32(S1)), "valu", [L_PS], [L_PS, L_Z])
    E.ins("v_fmaak_f32 v%d, v%d, v%d, %s" % (L_PC, L_PC, L_Z, f32(C2c)), "valu", [L_PC], [L_PC, L_Z])
    E.ins("v_fmaak_f32 v%d, v%d, v%d, %s" % (L_PS, L_PS, L_Z, f32(S0)), "valu", [L_PS], [L_PS, L_Z])
    E.ins("v_fmaak_f32 v%d, v%d, v%d, %s" % (L_PC, L_PC, L_Z, f32(C1c)), "valu", [L_PC], [L_PC, L_Z])
    E.ins("v_mul_f32 v%d, v%d, v%d" % (L_PS, L_Z, L_PS), "valu", [L_PS], [L_PS, L_Z])
    E.ins("v_fma_f32 v%d, v%d, v%d, -0.5" % (L_PC, L_PC, L_Z), "valu", [L_PC], [L_PC, L_Z])
    E.ins("v_fmac_f32 v%d, v%d, v%d" % (L_R, L_PS, L_R), "valu", [L_R], [L_PS, L_R])                 # sin before the sign
    E.ins("v_fma_f32 v%d, v%d, v%d, 1.0" % (L_PC, L_PC, L_Z), "valu", [L_PC], [L_PC, L_Z])           # cos before the sign
    E.ins("v_lshlrev_b32 v%d, 8, v%d" % (L_M, L_K), "valu", [L_M], [L_K])
    E.ins("v_xor_b32 v%d, v%d, v%d" % (L_SS, L_M, L_R), "valu", [L_SS], [L_M, L_R])
    E.ins("v_xor_b32 v%d, v%d, v%d" % (L_CC, L_M, L_PC), "valu", [L_CC], [L_M, L_PC])
    # the loads issued during the previous step (a, F) are due now; everything older has long returned
    E.ins("s_waitcnt lgkmcnt(0)", "wait")
    if s % 4 == 2:
        # the helper's progress counter for the check at the end of this step (as fresh as the schedule allows)
        E.ins("ds_read_b32 v%d, v%d" % (L_FLAG, L_FDADDR), "lds", [L_FLAG], [L_FDADDR])
    # x = a * (c + j s): (ar*c, ai*c) + (-(ai*s), ar*s)
    E.ins("v_pk_mul_f32 %s, %s, %s op_sel_hi:[1,0]" % (pair(L_T1), pair(a), pair(L_CC)), "pk", [L_T1, L_T1 + 1], [a, a + 1, L_CC])
    E.ins("v_pk_mul_f32 %s, %s, %s op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[0,1]" % (pair(L_T2), pair(a), pair(L_SS)), "pk",
          [L_T2, L_T2 + 1], [a, a + 1, L_SS])
    if s + 1 < TILE:
        E.ins("ds_read_b64 %s, v%d offset:%d" % (pair(an), L_AADDR, 8 * (s + 1)), "lds", [an, an + 1], [L_AADDR])
    E.ins("v_pk_add_f32 %s, %s, %s" % (pair(n), pair(L_T1), pair(L_T2)), "pk", [n, n + 1], [L_T1, L_T1 + 1, L_T2, L_T2 + 1])
    # two instructions between the write of x and the DPP that merges the older samples into its register
    if s + 1 < TILE:
        E.ins("ds_read_b128 %s, v%d offset:%d" % (quad(Fn), L_FADDR, 16 * ((s + 1 + 8) & 31)), "lds", list(range(Fn, Fn + 4)), [L_FADDR])
    if s % 4 == 3:
        E.ins("v_add_u32 v%d, 4, v%d" % (L_NEED, L_NEED), "valu", [L_NEED], [L_NEED])       # for the next check
    E.ins("v_mov_b32_dpp v%d, v%d row_shr:2 row_mask:0xf bank_mask:0xf" % (n, o), "dpp", [n], [o])
    E.ins("v_mov_b32_dpp v%d, v%d row_shr:2 row_mask:0xf bank_mask:0xf" % (n + 1, o + 1), "dpp", [n + 1], [o + 1])
    # newest tap of every position (slot 1) on the resident sums; the head lanes now hold the completed sums
    E.ins("v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,1,0]" % (pair(L_C14), pair(n), pair(L_TA), pair(L_R14)), "pk",
          [L_C14, L_C14 + 1], [n, n + 1, L_TA + 1, L_R14, L_R14 + 1])
    E.ins("v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,1,0]" % (pair(L_C32), pair(n), pair(L_TB), pair(L_R32)), "pk",
          [L_C32, L_C32 + 1], [n, n + 1, L_TB + 1, L_R32, L_R32 + 1])
    E.ins("ds_write_b64 v%d, %s offset:%d" % (L_XADDR, pair(n), 8 * s), "lds", [], [L_XADDR, n, n + 1])
    if s & 1:
        # x_{s-1}, x_s are in the ring (LDS executes a wave's instructions in order): tell the helper
        E.ins("ds_add_u32 v%d, v%d" % (L_XDADDR, L_TWO), "lds", [], [L_XDADDR, L_TWO])
    # fll_error: d = c14 - swap(c32) = (lbe.re, hbe.im), u = c14 + swap(c32) = (hbe.re, lbe.im)
    E.ins("v_pk_add_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" % (pair(L_D), pair(L_C14), pair(L_C32)),
          "pk", [L_D, L_D + 1], [L_C14, L_C14 + 1, L_C32, L_C32 + 1])
    E.ins("v_pk_add_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,0]" % (pair(L_U), pair(L_C14), pair(L_C32)),
          "pk", [L_U, L_U + 1], [L_C14, L_C14 + 1, L_C32, L_C32 + 1])
    # the sums hop one position inward; the tail lanes (no source lane) keep the far sums F
    E.ins("v_mov_b32_dpp v%d, v%d row_shl:2 row_mask:0xf bank_mask:0xf" % (f14, L_C14), "dpp", [f14], [L_C14])
    E.ins("v_max_f32 v%d, |v%d|, |v%d|" % (L_MX, L_U, L_D + 1), "valu", [L_MX], [L_U, L_D + 1])
    E.ins("v_min_f32 v%d, |v%d|, |v%d|" % (L_MN, L_U, L_D + 1), "valu", [L_MN], [L_U, L_D + 1])
    E.ins("v_max_f32 v%d, |v%d|, |v%d|" % (L_MX + 1, L_D, L_U + 1), "valu", [L_MX + 1], [L_D, L_U + 1])
    E.ins("v_min_f32 v%d, |v%d|, |v%d|" % (L_MN + 1, L_D, L_U + 1), "valu", [L_MN + 1], [L_D, L_U + 1])
    E.ins("v_pk_mul_f32 %s, %s, %%[p4] op_sel_hi:[1,0]" % (pair(L_MN), pair(L_MN)), "pk", [L_MN, L_MN + 1], [L_MN, L_MN + 1])
    E.ins("v_mov_b32_dpp v%d, v%d row_shl:2 row_mask:0xf bank_mask:0xf" % (f14 + 1, L_C14 + 1), "dpp", [f14 + 1], [L_C14 + 1])
    E.ins("v_pk_add_f32 %s, %s, %s" % (pair(L_MX), pair(L_MX), pair(L_MN)), "pk", [L_MX, L_MX + 1], [L_MX, L_MX + 1, L_MN, L_MN + 1])
    E.ins("v_mov_b32_dpp v%d, v%d row_shl:2 row_mask:0xf bank_mask:0xf" % (f32_, L_C32), "dpp", [f32_], [L_C32])
    E.ins("v_sub_f32 v%d, v%d, v%d" % (L_E, L_MX, L_MX + 1), "valu", [L_E], [L_MX, L_MX + 1])
    # PhaseControlLoop::advance with alpha == 0: freq = clamp(freq + beta*err), phase = wrap(phase + freq)
    E.ins("v_mul_f32 v%d, %%[beta], v%d" % (L_E, L_E), "valu", [L_E], [L_E])
    E.ins("v_mov_b32_dpp v%d, v%d row_shl:2 row_mask:0xf bank_mask:0xf" % (f32_ + 1, L_C32 + 1), "dpp", [f32_ + 1], [L_C32 + 1])
    E.ins("v_add_f32 v%d, v%d, v%d" % (L_E, L_FR, L_E), "valu", [L_E], [L_FR, L_E])
    E.ins("v_med3_f32 v%d, v%d, %%[minf], v%d" % (L_FR, L_E, L_MAXF), "valu", [L_FR], [L_E, L_MAXF])
    E.ins("v_add_f32 v%d, v%d, v%d" % (L_PH, L_PH, L_FR), "valu", [L_PH], [L_PH, L_FR])
    E.ins("v_bfi_b32 v%d, %%[absmask], v%d, v%d" % (L_T, L_2PI, L_PH), "valu", [L_T], [L_2PI, L_PH])
    E.ins("v_sub_f32 v%d, v%d, v%d" % (L_T, L_PH, L_T), "valu", [L_T], [L_PH, L_T])
    E.ins("v_cmp_gt_f32 vcc, |v%d|, %%[pi]" % L_PH, "valu", [], [L_PH], writes_vcc=True)
    # oldest tap of every position (slot 0) on the sums that just arrived
    E.ins("v_pk_fma_f32 %s, %s, %s, %s op_sel_hi:[1,0,1]" % (pair(L_R14), pair(n), pair(L_TA), pair(f14)), "pk",
          [L_R14, L_R14 + 1], [n, n + 1, L_TA, f14, f14 + 1])
    E.ins("v_pk_fma_f32 %s, %s, %s, %s op_sel_hi:[1,0,1]" % (pair(L_R32), pair(n), pair(L_TB), pair(f32_)), "pk",
          [L_R32, L_R32 + 1], [n, n + 1, L_TB, f32_, f32_ + 1])
    E.ins("v_cndmask_b32 v%d, v%d, v%d, vcc" % (L_PH, L_PH, L_T), "valu", [L_PH], [L_PH, L_T], reads_vcc=True)
    if s % 4 == 2:
        # the F loads of the next four steps fetch F_{s+10} .. F_{s+13}: f_done >= base + s + 14 (L_NEED runs with it)
        E.ins("s_waitcnt lgkmcnt(0)", "wait")
        E.ins("v_cmp_lt_i32 vcc, v%d, v%d" % (L_FLAG, L_NEED), "valu", [], [L_FLAG, L_NEED], writes_vcc=True)
        if "L" not in ABLATE:
            E.ins("s_cbranch_vccnz .Lslow%d_%%=" % s, "br")
        E.label(".Lback%d_%%=:" % s)


def gen_loop():
    E = Emitter()
    # ---- entry: operands -> fixed registers
    E.comment("state, taps, constants and addresses into the block's fixed registers")
    for (reg, opnd) in ((L_PH, "ph"), (L_FR, "fr"), (L_NEED, "need"), (L_AADDR, "a_addr"), (L_FADDR, "f_addr"), (L_XBASE, "x_base"),
                        (L_HEADMASK, "headmask"), (L_XDADDR, "xd_addr"), (L_FDADDR, "fd_addr"), (L_STUCK, "stuck_addr"), (L_MAXF, "maxf")):
        E.ins("v_mov_b32 v%d, %%[%s]" % (reg, opnd), "valu", [reg])
    for (reg, opnd) in ((L_XS[0], "xs"), (L_R14, "r14"), (L_R32, "r32"), (L_TA, "ta"), (L_TB, "tb")):
        E.ins("v_mov_b64 %s, %%[%s]" % (pair(reg), opnd), "valu", [reg, reg + 1])
    E.ins("v_mov_b32 v%d, %%[two]" % L_TWO, "valu", [L_TWO])      # only lane 0 addresses the counter, the others their own dump word
    E.ins("v_mov_b32 v%d, %s" % (L_CS2, f32(S2)), "valu", [L_CS2])
    E.ins("v_mov_b32 v%d, %s" % (L_CC3, f32(C3c)), "valu", [L_CC3])
    E.ins("v_mov_b32 v%d, %s" % (L_2PI, f32(FL_PI - (-FL_PI))), "valu", [L_2PI])
    E.label(".Ltile_%=:")
    # ---- per tile: x ring address of the head lanes ((base & 255) * 8 into the row; the other lanes write to a dump row)
    E.ins("s_and_b32 %[st], %[base], 0xff", "salu")
    E.ins("s_lshl_b32 %[st], %[st], 3", "salu")
    E.ins("v_and_b32 v%d, %%[st], v%d" % (L_T, L_HEADMASK), "valu", [L_T], [L_HEADMASK])
    E.ins("v_add_u32 v%d, v%d, v%d" % (L_XADDR, L_T, L_XBASE), "valu", [L_XADDR], [L_T, L_XBASE])
    E.ins("ds_read_b64 %s, v%d" % (pair(L_A[0]), L_AADDR), "lds", [L_A[0], L_A[0] + 1], [L_AADDR])
    E.ins("ds_read_b128 %s, v%d offset:%d" % (quad(L_F[0]), L_FADDR, 16 * 8), "lds", list(range(L_F[0], L_F[0] + 4)), [L_FADDR])
    prologue = E.n
    for s in range(TILE):
        loop_step(E, s)
    per_tile = E.n - prologue
    # ---- tile end: the A wave's other buffer next time, barrier of the epoch, next tile
    E.ins("v_xor_b32 v%d, 0x%x, v%d" % (L_AADDR, A_BUF_TOGGLE, L_AADDR), "valu", [L_AADDR], [L_AADDR])
    E.ins("s_add_u32 %[base], %[base], 32", "salu")
    E.ins("s_sub_u32 %[tiles], %[tiles], 1", "salu")
    E.ins("s_waitcnt lgkmcnt(0)", "wait")
    E.ins("s_barrier", "salu")
    E.ins("s_cmp_lg_u32 %[tiles], 0", "salu")
    E.ins("s_cbranch_scc1 .Ltile_%=", "br")
    # ---- exit: state back (the x pipeline ends a tile in the register set it started in)
    for (reg, opnd) in ((L_PH, "ph"), (L_FR, "fr")):
        E.ins("v_mov_b32 %%[%s], v%d" % (opnd, reg), "valu")
    for (reg, opnd) in ((L_XS[0], "xs"), (L_R14, "r14"), (L_R32, "r32")):
        E.ins("v_mov_b64 %%[%s], %s" % (opnd, pair(reg)), "valu")
    E.ins("s_branch .Lend_%=", "br")
    for s in range(TILE):
        if s % 4 == 2:
            spin(E, "slow%d" % s, L_FDADDR, L_FLAG,
                 lambda tmp: ["v_cmp_lt_i32 vcc, v%d, v%d" % (tmp, L_NEED)],
                 L_STUCK, "back%d" % s)
    E.label(".Lend_%=:")
    return E, per_tile


# ----------------------------------------------------------------------------------------------------------------------
# FLL helper wave
# ----------------------------------------------------------------------------------------------------------------------
TH = 17
H_TA, H_TB = 16, 33            # ta[0..16] = v16..v32, tb[0..16] = v33..v49
H_R14, H_R32 = 50, 82          # r14[i] = v[50+2i : 51+2i], r32[i] = v[82+2i : 83+2i], i = 0..15
H_XIN = (116, 120)             # the two samples of a pair (4 registers), two pairs in flight; the x pipeline lives in
                               # the register pair of the sample processed last
H_C = 124                      # c14 = v[124:125], c32 = v[126:127] (one ds_write_b128)
H_FLAG = (128, 129)
H_NEED, H_ONE, H_T = 130, 131, 132
H_XADDR, H_FADDR, H_XDADDR, H_FDADDR, H_STUCK, H_XROW, H_TAPADDR = 133, 134, 135, 136, 137, 138, 139
H_CLOBBER = list(range(16, 140))
BE_IM_OFFSET = 84 * 4          # byte offset of the imaginary taps behind the real ones in FusedLds::be84


def tap_operand(base, j):
    """(aligned register pair, modifiers) that broadcast tap register base+j to both halves of a packed multiply."""
    r = base + j
    if r % 2 == 0:
        return pair(r), "op_sel_hi:[1,0,1]"
    return pair(r - 1), "op_sel:[0,1,0]"


def helper_step(E, s, xr, xs_old, publish_after_write):
    """One far step, PH = s & 15.  xr = register pair holding x_s in the head lanes; the step shifts the pipeline
    (xs_old) into it.  Emits F_{s+16} to the ring."""
    ph = s & 15
    c14, c32 = H_C, H_C + 2
    E.comment("---- far step %d" % s)
    E.ins("v_mov_b32_dpp v%d, v%d row_shr:4 row_mask:0xf bank_mask:0xf" % (xr, xs_old), "dpp", [xr], [xs_old])
    E.ins("v_mov_b32_dpp v%d, v%d row_shr:4 row_mask:0xf bank_mask:0xf" % (xr + 1, xs_old + 1), "dpp", [xr + 1], [xs_old + 1])
    xs = xr

    def r14(i):
        return H_R14 + 2 * (i % 16)

    def r32(i):
        return H_R32 + 2 * (i % 16)

    def fma(dst, base, j, acc):
        op, mod = tap_operand(base, j)
        E.ins("v_pk_fma_f32 %s, %s, %s, %s %s" % (pair(dst), pair(xs), op, pair(acc), mod), "pk", [dst, dst + 1],
              [xs, xs + 1, base + j, acc, acc + 1])

    fma(c14, H_TA, TH - 1, r14(ph))
    fma(c32, H_TB, TH - 1, r32(ph))
    # middle taps on the other residents: independent of each other, they also fill the hazard gaps
    mids = []
    for q in range(1, TH - 1):
        mids.append((r14(ph + q), H_TA, TH - 1 - q))
        mids.append((r32(ph + q), H_TB, TH - 1 - q))
    for (rr, base, j) in mids[:2]:
        fma(rr, base, j, rr)
    # completed far sums F_{s+16} of the head lanes -> ring
    E.ins("ds_write_b128 v%d, %s offset:%d" % (H_FADDR, quad(H_C), 16 * ((s + 16) & 31)), "lds", [], [H_FADDR] + list(range(H_C, H_C + 4)))
    E.ins("ds_add_u32 v%d, v%d" % (H_FDADDR, H_ONE), "lds", [], [H_FDADDR, H_ONE])
    if publish_after_write:
        publish_after_write()
    # hop: the sums move one position inward (zero fill at the tail), then meet tap 0 of their new position
    sh14, sh32 = r14(ph), r32(ph)
    E.ins("v_mov_b32_dpp v%d, v%d row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (sh14, c14), "dpp", [sh14], [c14])
    E.ins("v_mov_b32_dpp v%d, v%d row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (sh14 + 1, c14 + 1), "dpp", [sh14 + 1], [c14 + 1])
    E.ins("v_mov_b32_dpp v%d, v%d row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (sh32, c32), "dpp", [sh32], [c32])
    E.ins("v_mov_b32_dpp v%d, v%d row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (sh32 + 1, c32 + 1), "dpp", [sh32 + 1], [c32 + 1])
    for (rr, base, j) in mids[2:]:
        fma(rr, base, j, rr)
    fma(sh14, H_TA, 0, sh14)
    fma(sh32, H_TB, 0, sh32)


def gen_helper():
    """The helper wave's whole life.  Tile iterations it = 0, 1, 2 rebuild the pipeline from the stored delay line
    (samples -96 .. -1, zeros before -80 under zero taps) and leave F_0 .. F_15 in the ring; iterations 3 .. ntiles+2 are
    the tiles of the call.  x of a pair is loaded speculatively together with the loop waves' progress counter while the
    previous pair is computed, and used only if the counter (loaded FIRST; LDS is in order) says it had been published;
    otherwise the slow path spins and reloads."""
    E = Emitter()
    E.comment("addresses, constants, taps")
    for (reg, opnd) in ((H_XROW, "x_row"), (H_FADDR, "f_addr"), (H_XDADDR, "xd_addr"), (H_FDADDR, "fd_addr"), (H_STUCK, "stuck_addr"),
                        (H_TAPADDR, "tap_addr")):
        E.ins("v_mov_b32 v%d, %%[%s]" % (reg, opnd), "valu", [reg])
    E.ins("v_mov_b32 v%d, %%[one]" % H_ONE, "valu", [H_ONE])      # only lane 0 addresses the counter, the others their own dump word
    E.ins("v_mov_b32 v%d, %d" % (H_NEED, -96 + 2), "valu", [H_NEED])
    for j in range(TH):
        E.ins("ds_read_b32 v%d, v%d offset:%d" % (H_TA + j, H_TAPADDR, 4 * j), "lds", [H_TA + j])
        E.ins("ds_read_b32 v%d, v%d offset:%d" % (H_TB + j, H_TAPADDR, BE_IM_OFFSET + 4 * j), "lds", [H_TB + j])
    for i in range(16):
        E.ins("v_mov_b64 %s, 0" % pair(H_R14 + 2 * i), "valu", [H_R14 + 2 * i, H_R14 + 2 * i + 1])
        E.ins("v_mov_b64 %s, 0" % pair(H_R32 + 2 * i), "valu", [H_R32 + 2 * i, H_R32 + 2 * i + 1])
    final_xs = H_XIN[1] + 2      # where the pipeline sits at the end of a tile = where a tile expects it
    E.ins("v_mov_b64 %s, 0" % pair(final_xs), "valu", [final_xs, final_xs + 1])
    E.ins("s_waitcnt lgkmcnt(0)", "wait")
    E.ins("s_mov_b32 %[base], -96", "salu")
    E.ins("s_mov_b32 %[it], 0", "salu")
    E.label(".Lhtile_%=:")
    E.ins("s_and_b32 %[st], %[base], 0xff", "salu")
    E.ins("s_lshl_b32 %[st], %[st], 3", "salu")
    E.ins("v_add_u32 v%d, %%[st], v%d" % (H_XADDR, H_XROW), "valu", [H_XADDR], [H_XROW])
    prologue = E.n
    cur_xs = final_xs
    for p in range(16):
        regs = H_XIN[p & 1]
        nxt = H_XIN[(p + 1) & 1]
        E.comment("==== pair %d" % p)
        if p == 0:
            # first pair of a tile: its samples appear only after the tile's barrier, nothing was prefetched
            if "H" not in ABLATE:
                E.ins("s_branch .Lhslow0_%=", "br")
            E.label(".Lhback0_%=:")
            if "H" in ABLATE:
                E.ins("ds_read2_b64 %s, v%d offset0:0 offset1:1" % (quad(regs), H_XADDR), "lds", list(range(regs, regs + 4)), [H_XADDR])
                E.ins("s_waitcnt lgkmcnt(0)", "wait")
        for k in range(2):
            s = 2 * p + k
            xr = regs + 2 * k
            helper_step_split(E, s, xr, cur_xs, k == 1, p, nxt)
            cur_xs = xr
        if p + 1 < 16:
            E.ins("v_add_u32 v%d, 2, v%d" % (H_NEED, H_NEED), "valu", [H_NEED], [H_NEED])
            E.ins("s_waitcnt lgkmcnt(0)", "wait")
            E.ins("v_cmp_lt_i32 vcc, v%d, v%d" % (H_FLAG[(p + 1) & 1], H_NEED), "valu", [], [H_FLAG[(p + 1) & 1], H_NEED], writes_vcc=True)
            if "H" not in ABLATE:
                E.ins("s_cbranch_vccnz .Lhslow%d_%%=" % (p + 1), "br")
            E.label(".Lhback%d_%%=:" % (p + 1))
    assert cur_xs == final_xs
    per_tile = E.n - prologue
    # tile end: need of the next tile's first pair, barriers (none during the rebuild, two after it: the end of the
    # prologue and epoch 0), next tile
    E.ins("v_add_u32 v%d, 2, v%d" % (H_NEED, H_NEED), "valu", [H_NEED], [H_NEED])
    E.ins("s_add_u32 %[base], %[base], 32", "salu")
    E.ins("s_cmp_lt_u32 %[it], 2", "salu")
    E.ins("s_cbranch_scc1 .Lhnobar_%=", "br")
    E.ins("s_waitcnt lgkmcnt(0)", "wait")
    E.ins("s_barrier", "salu")
    E.ins("s_cmp_lg_u32 %[it], 2", "salu")
    E.ins("s_cbranch_scc1 .Lhnobar_%=", "br")
    E.ins("s_barrier", "salu")
    E.label(".Lhnobar_%=:")
    E.ins("s_add_u32 %[it], %[it], 1", "salu")
    E.ins("s_cmp_lt_u32 %[it], %[iters]", "salu")
    E.ins("s_cbranch_scc1 .Lhtile_%=", "br")
    E.ins("s_branch .Lhend_%=", "br")
    for p in range(16):
        regs = H_XIN[p & 1]
        # every iteration loads the counter and THEN the two samples: when the counter is there, so are they
        E.label(".Lhslow%d_%%=:" % p)
        E.ins("s_mov_b32 %[spins], 0", "salu")
        E.label(".Lhspin%d_%%=:" % p)
        E.ins("ds_read_b32 v%d, v%d" % (H_T, H_XDADDR), "lds")
        E.ins("ds_read2_b64 %s, v%d offset0:%d offset1:%d" % (quad(regs), H_XADDR, 2 * p, 2 * p + 1), "lds")
        E.ins("s_waitcnt lgkmcnt(0)", "wait")
        E.ins("v_cmp_lt_i32 vcc, v%d, v%d" % (H_T, H_NEED), "valu")
        E.ins("s_cbranch_vccz .Lhback%d_%%=" % p, "br")
        E.ins("s_add_u32 %[spins], %[spins], 1", "salu")
        E.ins("s_cmp_lt_u32 %%[spins], 0x%x" % SPIN_LIMIT, "salu")
        E.ins("s_cbranch_scc1 .Lhspin%d_%%=" % p, "br")
        E.ins("v_mov_b32 v%d, 1" % H_T, "valu")
        E.ins("ds_write_b32 v%d, v%d" % (H_STUCK, H_T), "lds")
        E.ins("s_branch .Lhback%d_%%=" % p, "br")
    E.label(".Lhend_%=:")
    return E, per_tile


def helper_step_split(E, s, xr, xs_old, second, p, nxt):
    """helper_step plus the pair-level bookkeeping that has to sit at a fixed point of it: the speculative loads of the
    next pair (progress counter FIRST, then the two samples) go right behind the second step's ring write -- as late as
    the LDS latency allows, so that they see as much of the loop waves' progress as possible.  The registers they
    overwrite were left by the x pipeline during the pair's first step."""
    def prefetch():
        E.ins("ds_read_b32 v%d, v%d" % (H_FLAG[(p + 1) & 1], H_XDADDR), "lds", [H_FLAG[(p + 1) & 1]], [H_XDADDR])
        E.ins("ds_read2_b64 %s, v%d offset0:%d offset1:%d" % (quad(nxt), H_XADDR, 2 * (p + 1), 2 * (p + 1) + 1), "lds",
              list(range(nxt, nxt + 4)), [H_XADDR])
    helper_step(E, s, xr, xs_old, prefetch if (second and p + 1 < 16) else None)


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


def generate():
    body, l_tile = gen_loop()
    hbody, h_tile = gen_helper()
    parts = []
    parts.append("// fll_asm.inc -- GENERATED by gen_fll_asm.py; do not edit.  See that file for the schedule and the hazard rules.\n")
    parts.append("// loop wave: %d instruction slots per 32-sample tile (%.2f per sample), %d s_nop in the block\n" % (l_tile, l_tile / 32.0, body.nops))
    parts.append("// helper wave: %d instruction slots per tile (%.2f per sample), %d s_nop in the block\n" % (h_tile, h_tile / 32.0, hbody.nops))
    parts.append("#define FLL_LOOP_ASM \\\n" + c_string(body.text()).replace("\n", " \\\n") + "\n")
    parts.append("#define FLL_HELPER_ASM \\\n" + c_string(hbody.text()).replace("\n", " \\\n") + "\n")
    parts.append("#define FLL_LOOP_CLOBBERS %s\n" % ", ".join('"v%d"' % r for r in L_CLOBBER))
    parts.append("#define FLL_HELPER_CLOBBERS %s\n" % ", ".join('"v%d"' % r for r in H_CLOBBER))
    parts.append("#define FLL_ASM_LOOP_SLOTS_PER_TILE %d\n#define FLL_ASM_HELPER_SLOTS_PER_TILE %d\n" % (l_tile, h_tile))
    return "".join(parts), body, hbody, l_tile, h_tile


def main():
    text, body, hbody, l_tile, h_tile = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("fll_asm.inc is stale: run gen_fll_asm.py")
            return 1
        return 0
    with open(OUT, "w") as f:
        f.write(text)
    print("loop wave  : %d slots / tile = %.2f per sample; block: nops %d, %s" % (l_tile, l_tile / 32.0, body.nops, dict(sorted(body.counts.items()))))
    print("helper wave: %d slots / tile = %.2f per sample; block: nops %d, %s" % (h_tile, h_tile / 32.0, hbody.nops, dict(sorted(hbody.counts.items()))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
