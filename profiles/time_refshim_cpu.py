"""Reproducible CPU timing of the REFERENCE's own code on this path (container only: needs /root/reference) -- SURVEY.md Appendix B
probe 6 as a committed script.

Builds /root/reference/src/dsp/{pi4dqpsk,fll,complex_fd,pi4dqpsk_costas,dqpsk_sym_extr,bit_unpacker}.cpp where they lie against
the stand-in core headers of tests/refshim (g++ -O3 -march=native, scalar VOLK stand-ins: a real SDR++ build runs VOLK's SIMD dot
products in the FLL / RRC / interpolator, so those stages are faster in the field -- the figure is a floor for the reference, not
its best), runs one channel of the BASELINE generator (400 600 samples, 180-sample calls like SDR++ delivers them, and one-call-per-
4000) on ONE thread, per stage and for the whole chain, and next to it the two CPU ports bench.py times on the GPU box (the oracle =
`cpu_baseline`, kind "port"; oracle/tetra_fast.c = `cpu_baseline_fast`) on the same core, same samples.
Writes profiles/r05/refshim_cpu_timing.{json,md}.  Never runs on the GPU box (no /root/reference there).
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("TETRA_REFERENCE_DIR", "/root/reference")
SHIM = os.path.join(ROOT, "tests", "refshim")
SOURCES = ["pi4dqpsk.cpp", "fll.cpp", "complex_fd.cpp", "pi4dqpsk_costas.cpp", "dqpsk_sym_extr.cpp", "bit_unpacker.cpp"]
STAGES = ["AGC", "FLL", "RRC", "COMPLEX_FD", "COSTAS", "SYM_EXTR", "BIT_UNPACK"]


def build():
    out = os.path.join(SHIM, "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libref_timing.so")
    srcs = [os.path.join(SHIM, "ref_timing.cpp")] + [os.path.join(REF, "src", "dsp", s) for s in SOURCES]
    subprocess.run(["g++", "-std=c++17", "-O3", "-march=native", "-fno-access-control", "-fPIC", "-shared", "-w", "-I", SHIM, "-I",
                    os.path.join(REF, "src"), "-o", so] + srcs, check=True)
    return so


def main():
    import tetra_amd
    from oracle import binding as ob
    if not os.path.isfile(os.path.join(REF, "src", "dsp", "pi4dqpsk.cpp")):
        sys.exit("reference sources not present: this probe runs in the build container only")
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})          # one core for everything
    L = C.CDLL(build())
    L.ref_time_stages.restype = C.c_int
    L.ref_time_stages.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_double, C.c_double, C.c_int] + [C.c_double] * 7 + [C.c_void_p]
    n = 400600
    iq, _, _ = tetra_amd.pkg.synth.gen_channel(n, 7)
    cfg = ob.default_cfg()
    res = {}
    for chunk in (180, 4000):
        ns = np.zeros(8, np.float64)
        best = None
        for _ in range(3):           # best of three passes
            L.ref_time_stages(n, iq.ctypes.data_as(C.c_void_p), chunk, 1, cfg.symbolrate, cfg.samplerate, cfg.rrc_tap_count, cfg.rrc_beta,
                              cfg.agc_rate, cfg.costas_bandwidth, cfg.fll_bandwidth, cfg.omega_gain, cfg.mu_gain, cfg.omega_rel_limit,
                              ns.ctypes.data_as(C.c_void_p))
            if best is None or ns[7] < best[7]:
                best = ns.copy()
        res["reference_chunk_%d" % chunk] = dict(ns_per_sample={s: float(best[i]) for i, s in enumerate(STAGES)}, chain_ns_per_sample=float(best[7]),
                                                 msamples_per_s_per_core=1e3 / float(best[7]))

    def time_port(fn):
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best
    o_iq = iq[None, :]
    t_oracle = time_port(lambda: ob.process_batch(o_iq, chunk=4000, threads=1))
    t_fast = time_port(lambda: ob.fast_process_batch(o_iq, chunk=4000, threads=1))
    res["port_oracle_contract"] = dict(chain_ns_per_sample=1e9 * t_oracle / n, msamples_per_s_per_core=n / t_oracle / 1e6)
    res["port_fast"] = dict(chain_ns_per_sample=1e9 * t_fast / n, msamples_per_s_per_core=n / t_fast / 1e6)
    cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][:1]
    res["host"] = dict(cpu=cpu[0] if cpu else "?", threads=1, samples=n, compiler=subprocess.run(["g++", "--version"], capture_output=True, text=True).stdout.splitlines()[0],
                       flags="-O3 -march=native, scalar VOLK stand-ins (tests/refshim)")
    out = os.path.join(ROOT, "profiles", "r05")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "refshim_cpu_timing.json"), "w") as f:
        json.dump(res, f, indent=1)
    lines = ["# Reference src/dsp code (shim build) vs the CPU ports, one thread on %s, %d samples" % (res["host"]["cpu"], n), "",
             "| what | " + " | ".join(STAGES) + " | chain ns/sample | Msamples/s/core |", "|---|" + "---|" * (len(STAGES) + 2)]
    for chunk in (180, 4000):
        r = res["reference_chunk_%d" % chunk]
        lines.append("| reference objects, %d-sample calls | " % chunk + " | ".join("%.1f" % r["ns_per_sample"][s] for s in STAGES) +
                     " | %.1f | %.2f |" % (r["chain_ns_per_sample"], r["msamples_per_s_per_core"]))
    for k, name in (("port_oracle_contract", "oracle/tetra_oracle.c (bench.py cpu_baseline, kind \"port\")"), ("port_fast", "oracle/tetra_fast.c (cpu_baseline_fast)")):
        lines.append("| %s | " % name + " | ".join("" for _ in STAGES) + " | %.1f | %.2f |" % (res[k]["chain_ns_per_sample"], res[k]["msamples_per_s_per_core"]))
    md = "\n".join(lines)
    with open(os.path.join(out, "refshim_cpu_timing.md"), "w") as f:
        f.write(md + "\n")
    print(md)


if __name__ == "__main__":
    main()
