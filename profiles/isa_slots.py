#!/usr/bin/env python3
"""Instruction-slot census of the fused demodulator kernel's loops (profiling aid, not product code).

A gfx950 wavefront issues ONE instruction of any kind (VALU, SALU, s_nop, s_waitcnt, LDS, branch) per ~4.7-4.9 shader clocks
(profiles/r02/r02_a_issue_model.md), so the time of a role's inner loop is its instruction count.  This script compiles
tetra_demod.hip to gfx950 assembly (no GPU needed), finds every backward branch of one kernel instantiation and prints the
instruction mix of each loop body.

    python profiles/isa_slots.py [--kernel k_fusedILb1ELb0ELi16] [--min 20] [--dump LABEL]
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-S", "--cuda-device-only"]


def cat(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        if "dpp" in ins or "row_" in ins or "quad_perm" in ins:
            return "dpp"
        if op.startswith("v_pk_"):
            return "pk"
        if op.startswith("v_cmp"):
            return "vcmp"
        if op in ("v_sqrt_f32", "v_rcp_f32", "v_rsq_f32"):
            return "trans"
        if op.startswith("v_mov_b32"):
            return "vmov"
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "br"
    if op.startswith("s_barrier"):
        return "bar"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_fusedILb1ELb0ELi16")
    ap.add_argument("--min", type=int, default=20)
    ap.add_argument("--dump", default=None, help="print the body of the loop that starts at this label")
    ap.add_argument("--asm", default="/tmp/isa/tetra_demod.s")
    ap.add_argument("--no-build", action="store_true")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.asm), exist_ok=True)
    if not a.no_build:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", a.asm, os.path.join(CSRC, "tetra_demod.hip")], check=True,
                       stderr=subprocess.DEVNULL)
    text = open(a.asm).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z.*" + re.escape(a.kernel) + r".*:", l))
    end = next(i for i in range(start + 1, len(text)) if text[i].startswith(".Lfunc_end"))
    lines = text[start:end]
    print("kernel %s: %d lines" % (a.kernel, len(lines)))
    for l in lines[-1:]:
        pass
    meta = [l for l in text[end:end + 60] if "NumVgprs" in l or "ScratchSize" in l or "LDSByteSize" in l or "Occupancy" in l]
    print("  " + " | ".join(m.strip("; ").strip() for m in meta))
    lab = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l.strip())
        if m:
            lab[m.group(1)] = i

    def body(i0, i1):
        return [x.strip() for x in lines[i0:i1 + 1]
                if x.strip() and not x.strip().startswith(";") and not x.strip().startswith(".")]

    for i, l in enumerate(lines):
        st = l.strip()
        if st.startswith("s_cbranch") or st.startswith("s_branch"):
            t = st.split()[-1]
            if t in lab and lab[t] < i:
                b = body(lab[t], i)
                if len(b) < a.min:
                    continue
                cnt = {}
                for x in b:
                    c = cat(x)
                    cnt[c] = cnt.get(c, 0) + 1
                print("%-10s lines %5d-%5d  slots %4d  %s" % (t, lab[t], i, len(b), dict(sorted(cnt.items()))))
                if a.dump == t:
                    print("\n".join("    " + x for x in b))
    return 0


if __name__ == "__main__":
    sys.exit(main())
