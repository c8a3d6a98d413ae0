"""Independent lint of the generated FLL assembly (csrc/fll_asm.inc, fll4_asm.inc): the gfx950 hazards the hardware does not
interlock must be covered by distance in the instruction stream.  The generator enforces them while it schedules
(gen_fll_asm.py, Emitter._need); this test re-derives them from the emitted TEXT alone, so a scheduling bug cannot hide
behind the generator's own bookkeeping.  Rules (LLVM GCNHazardRecognizer for gfx940/gfx950):
  H1  a VGPR written by a packed-FP32 instruction is not read by the next instruction;
  H2  a VGPR written by a VALU instruction is not read (or merged into, as `old`) by a DPP instruction within the next two;
  H3  VCC written by v_cmp is not read by v_cndmask within the next two.
Loops are checked across their back edge as well (the body followed by itself)."""
import os
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdrpp-tetra-demodulator_amd", "csrc")
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def _regs(text):
    out = []
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def _parse(path, macro):
    """[(kind, writes, reads, reads_vcc, writes_vcc, text)] in stream order, labels as ("label", name)."""
    src = open(path).read()
    body = src[src.index("#define %s_ASM" % macro):]
    body = body[:body.index("\n#define %s_CLOBBERS" % macro)]
    items = []
    for line in re.findall(r'"([^"]*)\\n"', body):
        line = line.strip()
        if line.endswith(":"):
            items.append(("label", line[:-1]))
            continue
        op, _, rest = line.partition(" ")
        ops = [o.strip() for o in re.split(r",(?![^\[]*\])", rest.split(" row_")[0].split(" op_sel")[0].split(" neg_")[0].split(" offset")[0])] if rest else []
        if op.startswith("s_") or op.startswith("ds_write"):
            kind = "nop" if op == "s_nop" else ("br" if op.startswith("s_cbranch") else "other")
            reads = _regs(rest) if op.startswith("ds_write") else []
            items.append((kind, [], reads, False, False, line))
            continue
        if op.startswith("ds_read"):
            items.append(("lds", _regs(ops[0]), _regs(ops[1]), False, False, line))
            continue
        assert op.startswith("v_"), line
        kind = "dpp" if op.endswith("_dpp") else ("pk" if op.startswith("v_pk_") else "valu")
        if op.startswith("v_cmp"):
            items.append((kind, [], _regs(rest), False, True, line))
            continue
        writes = _regs(ops[0])
        reads = [r for o in ops[1:] for r in _regs(o)]
        if op.startswith("v_fmac") or kind == "dpp":          # accumulator / the DPP `old` operand is the destination
            reads = reads + writes
        reads_vcc = op.startswith("v_cndmask") and "vcc" in rest
        # an SGPR operand written as %[name] in the ASM text names no VGPR here except the two loop variables moved at
        # the block's ends (v_mov_b32 v70, %[ph]): nothing to track for them
        items.append((kind, writes, reads, reads_vcc, False, line))
    return items


def _check(seq, where):
    bad = []
    for i, (kind, writes, reads, reads_vcc, _wv, text) in enumerate(seq):
        for back in (1, 2):
            if i - back < 0:
                continue
            pk, pw, _pr, _prv, pwv, ptext = seq[i - back]
            if pk in ("nop", "other", "br", "label"):
                continue
            hit = set(pw) & set(reads)
            if back == 1 and pk == "pk" and hit:
                bad.append((where, "H1", ptext, text))
            if kind == "dpp" and pk in ("valu", "pk", "dpp") and hit:
                bad.append((where, "H2", ptext, text))
            if reads_vcc and pwv:
                bad.append((where, "H3", ptext, text))
    return bad


@pytest.mark.parametrize("fname,macro", [("fll_asm.inc", "FLL_WAVE"), ("fll4_asm.inc", "FLL4_WAVE"), ("fll16_asm.inc", "FLL16_WAVE")])
def test_generated_assembly_respects_the_hazard_distances(fname, macro):
    items = _parse(os.path.join(CSRC, fname), macro)
    instrs = [it for it in items if it[0] != "label"]
    assert len(instrs) > 1500
    bad = _check(instrs, "stream")
    # every loop: the body followed by itself (hazards across the back edge)
    labels = {it[1]: k for k, it in enumerate(items) if it[0] == "label"}
    nloops = 0
    for k, it in enumerate(items):
        if it[0] == "br":
            target = it[5].split()[-1]
            if target in labels and labels[target] < k:
                body = [x for x in items[labels[target]:k + 1] if x[0] != "label"]
                bad += _check(body[-3:] + body[:3], "back edge of " + target)
                nloops += 1
    assert nloops == 2                      # the replay loop and the tile loop
    assert not bad, bad[:5]


def test_the_lint_sees_a_planted_hazard():
    items = _parse(os.path.join(CSRC, "fll_asm.inc"), "FLL_WAVE")
    instrs = [it for it in items if it[0] != "label"]
    k = next(i for i, it in enumerate(instrs) if it[0] == "nop")          # drop an s_nop: its hazard comes back
    assert _check(instrs[:k] + instrs[k + 1:], "stream")
