"""The pin that CAN be closed (missing 1 / next 1): in an SDR++ build the dsp::block mirror designs its tables with the
INCLUDED core headers' generators -- taps::rootRaisedCosine, math::sinc / math::phasor (the reference's own band-edge arithmetic,
fll.cpp:61-95), taps::windowedSinc + window::nuttall + multirate::buildPolyphaseBank -- in init() and in every re-designing setter
(host/sdrpp_tables.h, host/pi4dqpsk_gpu.cpp under TETRA_WITH_SDRPP), hands them over as caller tables (cfg.rrc_taps /
bandedge_taps / interp_bank, tetra_demod_set_tables) and the kernels run exactly those.

Here "SDR++'s headers" are tests/refshim (ours) in two variants: pi as the double constant (default) and
pi as the float macro FL_M_PI inside the double expressions (-DREFSHIM_FLOAT_PI; how upstream is recalled to spell it -- not
verifiable in this image).  The tests show (1) the route: whatever the included headers compute IS what the handle holds, bit for
bit, after init and after every setter; (2) what hangs on the spelling of pi: most taps change their bit pattern, no bit after
lock does; (3) the kernels given the float-pi tables equal the oracle given the same tables, bit for bit.
"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PK = os.path.join(ROOT, "sdrpp-tetra-demodulator_amd")
SHIM = os.path.join(ROOT, "tests", "refshim")
SRC = os.path.join(ROOT, "tests", "host", "test_sdrpp_tables.cpp")


def _build(pkg, float_pi):
    pkg.build.build()
    exe = os.path.join(ROOT, "tests", "host", "test_sdrpp_tables_fpi" if float_pi else "test_sdrpp_tables")
    srcs = [SRC, os.path.join(PK, "host", "pi4dqpsk_gpu.cpp")]
    deps = srcs + [os.path.join(PK, "host", f) for f in ("pi4dqpsk_gpu.h", "sdrpp_tables.h", "dsp_compat.h")] + \
        [os.path.join(ROOT, "include", "tetra_demod.h"), pkg.build.LIB] + \
        [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(SHIM, "dsp")) for f in fs]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-DTETRA_WITH_SDRPP", "-I", SHIM] +
                       (["-DREFSHIM_FLOAT_PI"] if float_pi else []) + srcs +
                       ["-L", PK, "-ltetra_demod_hip", "-Wl,-rpath," + PK, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def _tables(exe, tmp_path, count=65, beta=0.35, symbolrate=18000, samplerate=36000, tag=""):
    out = tmp_path / ("tables%s.f32" % tag)
    r = subprocess.run([exe, "tables", str(out), str(count), repr(beta), repr(symbolrate), repr(samplerate)], capture_output=True, text=True)
    assert r.returncode == 0 and "from the included headers: yes" in r.stdout, (r.returncode, r.stdout, r.stderr)
    return _split(np.fromfile(out, np.float32), count, count)


def _split(a, nt, nbe):
    assert a.size == nt + 2 * nbe + 1024, (a.size, nt, nbe)
    return dict(rrc=a[:nt].copy(), be_re=a[nt:nt + nbe].copy(), be_im=a[nt + nbe:nt + 2 * nbe].copy(), bank=a[nt + 2 * nbe:].reshape(128, 8).copy())


def _install(o, t):
    """The oracle with caller tables: what tetra_oracle_design made is overwritten with `t` (lengths must agree)."""
    assert o.tab.ntaps == len(t["rrc"]) and o.tab.ntaps_be == len(t["be_re"])
    for i, v in enumerate(t["rrc"]):
        o.tab.rrc[i] = float(v)
    for i, (a, b) in enumerate(zip(t["be_re"], t["be_im"])):
        o.tab.be_a[i] = float(a)
        o.tab.be_b[i] = float(b)
    for p in range(128):
        for k in range(8):
            o.tab.bank[p][k] = float(t["bank"][p, k])


def _u32(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_included_headers_design_the_tables_double_pi_equals_the_restatement(pkg, oracle, tmp_path):
    """Default shim (pi as the double constant): the mirror's SDR++ route reproduces the library's own restatement
    (csrc/design.hpp = oracle/tetra_oracle.c's design) bit for bit -- RRC, both band-edge halves, interpolator bank -- for the
    plugin's parameters and for other counts / roll-offs / rates."""
    exe = _build(pkg, False)
    for i, (count, beta, sr, fs) in enumerate([(65, 0.35, 18000, 36000), (49, 0.5, 17000, 34000), (71, 1.0, 18000, 36000), (33, 0.2, 18000, 50000),
                                               (129, 0.35, 18000, 36000)]):
        t = _tables(exe, tmp_path, count, beta, sr, fs, tag=str(i))
        cfg = oracle.default_cfg()
        cfg.rrc_tap_count, cfg.rrc_beta, cfg.symbolrate, cfg.samplerate = count, beta, sr, fs
        o = oracle.Oracle(cfg)
        a, b = o.bandedge_taps()
        assert np.array_equal(_u32(t["rrc"]), _u32(o.rrc_taps())), (count, beta)
        assert np.array_equal(_u32(t["be_re"]), _u32(a)) and np.array_equal(_u32(t["be_im"]), _u32(b)), (count, beta)
        assert np.array_equal(_u32(t["bank"]), _u32(o.interp_bank()))


def test_float_pi_variant_changes_most_taps_and_no_bit_after_lock(pkg, oracle, synth, tmp_path):
    """What hangs on how upstream spells pi.  With FL_M_PI (3.1415926535f = 3.14159274...) in place of the double constant, 58 of
    the 65 RRC taps and 849 of the 1024 interpolator taps change their bit pattern (the band-edge taps do not: fll.cpp:89-90
    already computes with FL_M_PI in float) -- by at most 1e-4 of the largest tap.  On the BASELINE workload's channels the oracle run
    with either table set decides every bit after lock the same; symbols differ by what a change of the reduction order is
    worth (rms 2e-3, max 1.1e-2 -- the interpolator's phase index floor(128 mu) flips now and then): measured
    rms <= 4.1e-3, max 1.3e-2 over the 24 channels, modulo the Costas loop's lock quadrant."""
    d = _tables(_build(pkg, False), tmp_path, tag="d")
    f = _tables(_build(pkg, True), tmp_path, tag="f")
    n_rrc = int((_u32(d["rrc"]) != _u32(f["rrc"])).sum())
    n_bank = int((_u32(d["bank"]) != _u32(f["bank"])).sum())
    assert np.array_equal(_u32(d["be_re"]), _u32(f["be_re"])) and np.array_equal(_u32(d["be_im"]), _u32(f["be_im"]))
    assert n_rrc >= 50 and n_bank >= 800, (n_rrc, n_bank)          # measured: 58 / 65 and 849 / 1024
    assert np.abs(d["rrc"] - f["rrc"]).max() <= 1e-4 * np.abs(d["rrc"]).max()
    assert np.abs(d["bank"] - f["bank"]).max() <= 1e-4 * np.abs(d["bank"]).max()
    N = 36000
    worst_rms = worst_max = 0.0
    differing_before_lock = other_quadrant = 0
    for seed in range(1234, 1234 + 24):
        iq, _, _ = synth.gen_channel(N, seed)
        od, of = oracle.Oracle(), oracle.Oracle()
        _install(of, f)
        rd, rf = od.process(iq), of.process(iq)
        assert rd["bits"].size == rf["bits"].size, seed
        half = rd["bits"].size // 2
        assert np.array_equal(rd["bits"][half:], rf["bits"][half:]), seed            # after lock: never
        differing_before_lock += int((rd["bits"][:half] != rf["bits"][:half]).sum())
        # the Costas loop may settle on another of its four equivalent lock points (same differential bits, symbols turned by a
        # multiple of pi/2) when an early decision falls the other way: compare modulo that turn
        a, b = rd["sym"][half // 2:], rf["sym"][half // 2:]
        errs = [np.abs(a - b * (1j ** k)) for k in range(4)]
        k = int(np.argmin([float(e.mean()) for e in errs]))
        other_quadrant += k != 0
        worst_rms = max(worst_rms, float(np.sqrt((errs[k] ** 2).mean())))
        worst_max = max(worst_max, float(errs[k].max()))
    assert worst_rms <= 6e-3 and worst_max <= 3e-2, (worst_rms, worst_max)
    print("float-pi tables: %d / 65 RRC taps, %d / 1024 bank taps differ; symbols after lock rms %.2e max %.2e (modulo the lock quadrant: %d of 24 "
          "channels settle on another one); %d bits differ before lock" % (n_rrc, n_bank, worst_rms, worst_max, other_quadrant, differing_before_lock))


REF = os.environ.get("TETRA_REFERENCE_DIR", "/root/reference")


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "src", "dsp", "pi4dqpsk.cpp")), reason="reference sources not present (container-only)")
@pytest.mark.parametrize("float_pi", [False, True])
def test_sdrpp_route_hands_over_what_the_reference_objects_design_for_themselves(pkg, tmp_path, float_pi):
    """The closure itself (container only): the REFERENCE's own objects, compiled where they lie against a header variant, design
    their tables inside PI4DQPSK::init / FLL::init / COMPLEX_FD::init and the setters; host/sdrpp_tables.h, compiled against the SAME
    headers, must produce those very tables -- RRC, both band-edge halves (the one piece of the reference's own arithmetic restated on
    the route, fll.cpp:61-95), interpolator bank -- bit for bit: after init for several parameter sets, and after setSymbolrate /
    setSamplerate / setRRCParams / setRRCTapCount / setRRCBeta(int).  Together with test_handle_runs_the_included_headers_tables
    (GPU: handle tables == sdrpp_tables output) this says: in an SDR++ build the kernels run the tables the reference would."""
    import ctypes as C
    out_dir = os.path.join(SHIM, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libref_shim_fpi.so" if float_pi else "libref_shim_tables.so")
    srcs = [os.path.join(SHIM, "ref_driver.cpp")] + [os.path.join(REF, "src", "dsp", f) for f in
                                                     ("pi4dqpsk.cpp", "fll.cpp", "complex_fd.cpp", "pi4dqpsk_costas.cpp", "dqpsk_sym_extr.cpp", "bit_unpacker.cpp")]
    deps = srcs + [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(SHIM, "dsp")) for f in fs]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-access-control", "-fPIC", "-shared", "-w", "-I", SHIM, "-I",
                        os.path.join(REF, "src")] + (["-DREFSHIM_FLOAT_PI"] if float_pi else []) + ["-o", so] + srcs, check=True)
    L = C.CDLL(so)
    L.ref_create.restype = C.c_void_p
    L.ref_create.argtypes = [C.c_double, C.c_double, C.c_int] + [C.c_double] * 7
    L.ref_destroy.argtypes = [C.c_void_p]
    L.ref_set_param.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.ref_set_rrc_params.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.ref_get_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.POINTER(C.c_int)]
    L.ref_get_tables.restype = C.c_int
    exe = _build(pkg, float_pi)

    def ref_tables(h):
        rrc, re, im, bank = (np.zeros(n, np.float32) for n in (129, 129, 129, 1024))
        nb = C.c_int(0)
        n = L.ref_get_tables(h, *(a.ctypes.data_as(C.c_void_p) for a in (rrc, re, im, bank)), C.byref(nb))
        return dict(rrc=rrc[:n], be_re=re[:nb.value], be_im=im[:nb.value], bank=bank.reshape(128, 8))

    def same(a, b):
        return a.shape == b.shape and np.array_equal(_u32(a), _u32(b))

    omega_gain, mu_gain = 1.5636e-4, 0.017603
    for i, (count, beta, sr, fs) in enumerate([(65, 0.35, 18000, 36000), (49, 0.5, 17000, 34000), (71, 1.0, 18000, 36000), (33, 0.2, 18000, 50000)]):
        h = L.ref_create(sr, fs, count, beta, 0.02, 0.01, 0.006, omega_gain, mu_gain, 0.02)
        want = ref_tables(h)
        got = _tables(exe, tmp_path, count, beta, sr, fs, tag="i%d" % i)
        assert all(same(want[k], got[k]) for k in ("rrc", "be_re", "be_im", "bank")), (count, beta, sr, fs)
        L.ref_destroy(h)
    # the setters: what the reference's objects hold afterwards == the route's re-design with the mirror's bookkeeping
    h = L.ref_create(18000.0, 36000.0, 65, 0.35, 0.02, 0.01, 0.006, omega_gain, mu_gain, 0.02)
    be0 = ref_tables(h)
    count, beta, sr, fs = 65, 0.35, 18000.0, 36000.0
    for step, (what, arg) in enumerate([("sr", 17000.0), ("fs", 34000.0), ("rrc", (49, 0.5)), ("count", 71), ("beta_int", 1), ("rrc", (65, 0.35))]):
        if what == "sr":
            L.ref_set_param(h, 0, arg); sr = arg
        elif what == "fs":
            L.ref_set_param(h, 1, arg); fs = arg
        elif what == "rrc":
            L.ref_set_rrc_params(h, arg[0], arg[1]); count, beta = arg
        elif what == "count":
            L.ref_set_param(h, 2, float(arg)); count = arg
        else:
            L.ref_set_param(h, 3, float(arg)); beta = float(int(arg))
        want = ref_tables(h)
        got = _tables(exe, tmp_path, count, beta, sr, fs, tag="s%d" % step)
        assert same(want["rrc"], got["rrc"]), (step, what)
        # ... and the FLL's filters and the bank are still the construction-time ones (no PI4DQPSK setter re-designs them)
        assert same(want["be_re"], be0["be_re"]) and same(want["be_im"], be0["be_im"]) and same(want["bank"], be0["bank"]), (step, what)
    L.ref_destroy(h)


STEPS = [  # (what the driver's step does to the oracle, (count, quirks-mode tap-count setter needed))
    lambda o: None,
    lambda o: o.set_param(0, 17000, quirks=True),
    lambda o: o.set_param(1, 34000, quirks=True),
    lambda o: o.set_param(2, 49, quirks=True),
    lambda o: o.set_param(2, 71, quirks=True),
    lambda o: None,
    lambda o: (o.set_param(0, 18000, quirks=True), o.set_param(1, 36000, quirks=True), o.set_param(2, 65, quirks=True)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("float_pi", [False, True])
def test_handle_runs_the_included_headers_tables(pkg, oracle, synth, tmp_path, float_pi):
    """The mirror built with -DTETRA_WITH_SDRPP against either shim variant: after init() and after each of setSymbolrate,
    setSamplerate, setRRCParams, setRRCTapCount (growing), setRRCBeta(int) and the way back, tetra_demod_get_tables returns the
    included headers' output bit for bit (the driver checks; a difference is its exit code), the band-edge filters and the bank
    stay the construction-time ones like the reference's (pi4dqpsk.cpp:32-118) -- and the kernels' symbols, bits and counts
    equal the oracle's when the oracle is handed the same tables (float-pi: tables the library's own restatement never
    produces)."""
    exe = _build(pkg, float_pi)
    N, chunk = 12000, 4000
    iq, _, _ = synth.gen_channel(N, 4242)
    f_in = tmp_path / "iq.f32"
    iq.view(np.float32).tofile(f_in)
    prefix = str(tmp_path / "run")
    r = subprocess.run([exe, "mirror", str(f_in), str(chunk), prefix], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    o = oracle.Oracle()
    counts = [65, 65, 65, 49, 71, 71, 65]
    for step in range(7):
        STEPS[step](o)
        t = _split(np.fromfile("%s.%d.tables" % (prefix, step), np.float32), counts[step], 65)
        if float_pi and step == 0:
            assert not np.array_equal(_u32(t["rrc"]), _u32(o.rrc_taps()))      # really other tables than the restatement's
        _install(o, t)
        sym = np.fromfile("%s.%d.sym" % (prefix, step), np.float32).view(np.complex64)
        bits = np.fromfile("%s.%d.bits" % (prefix, step), np.uint8)
        ws, wb = [], []
        for pos in range(0, N, chunk):
            w = o.process(iq[pos:pos + chunk])
            ws.append(w["sym"])
            wb.append(w["bits"])
        ws, wb = np.concatenate(ws), np.concatenate(wb)
        assert bits.size == wb.size and np.array_equal(bits, wb), step
        assert np.array_equal(sym.view(np.uint32), ws.view(np.uint32)), step
