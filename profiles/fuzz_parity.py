#!/usr/bin/env python3
"""Time-bounded parity fuzzer: GPU (through the C ABI) against the oracle on random draws over everything the pytest cases vary
one or two at a time -- rates (0.12 ... 4 samples per symbol, at create; below ~1 several symbols leave one offset), tap counts 2 ... 129, RRC tap count and roll-off, every loop constant, the
three workgroup shapes and the automatic plan, input layout, symbol output, the quality statistic, TETRA_FLAG_REFERENCE_QUIRKS
with resets, call lengths 0 ... 3000 with carried state, a loop setter in the middle of the stream, and inputs the synthetic
TETRA channels do not contain (silence, noise only, amplitudes 1e-6 and 2).  Bits, bit counts, symbols (bit patterns) and
the loop state after the last call must equal the oracle's for every channel.

    python profiles/fuzz_parity.py --seconds 240 [--seed 1] > gpurun_out/fuzz.json

One JSON line: cases run, channel-calls compared, bits compared, the first failures (empty = all equal).  Test infrastructure:
uses oracle/ as the checker, like tests/."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tetra_amd  # noqa: E402

from oracle import binding as oracle  # noqa: E402

pkg = tetra_amd.pkg
B = pkg.binding
LOOP_PARAMS = ["agc_rate", "costas_bandwidth", "fll_bandwidth", "omega_gain", "mu_gain", "omega_rel_limit"]


def _u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


def draw_params(rng):
    sps = float(rng.choice([2.0, 2.0, 50000 / 18000, 40000 / 18000, 1.4, 1.06, 3.0, 4.0, 1.8]))
    p = dict(symbolrate=18000.0, samplerate=18000.0 * sps)
    if rng.integers(0, 2):
        p["rrc_tap_count"] = int(rng.integers(2, 73))
    if rng.integers(0, 2):
        p["rrc_beta"] = float(rng.uniform(0.2, 0.5))
    if rng.integers(0, 2):
        p["agc_rate"] = float(10 ** rng.uniform(-2.5, -1))
    if rng.integers(0, 2):
        p["costas_bandwidth"] = float(10 ** rng.uniform(-2.7, -1.3))
    if rng.integers(0, 2):
        p["fll_bandwidth"] = float(10 ** rng.uniform(-3, -1.7))
    if rng.integers(0, 2):
        p["omega_rel_limit"] = float(rng.uniform(0.002, 0.03))
    if rng.integers(0, 3) == 0:
        p["mu_gain"] = float(rng.uniform(0.005, 0.03))
        p["omega_gain"] = float(10 ** rng.uniform(-4.5, -3.5))
    return sps, p


def oracle_cfg(p):
    cfg = oracle.default_cfg()
    for k, v in p.items():
        setattr(cfg, k, v)
    return cfg


def special_channel(rng, kind, n):
    z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    if kind == 0:
        return np.zeros(n, np.complex64)
    if kind == 1:
        return z * np.float32(0.3)
    if kind == 2:
        return z * np.float32(1e-6)
    return z * np.float32(2.0)      # (FastAGC diverges to Inf/NaN once rate x amplitude reaches 2: the reference's UB, not a parity case)


DRY = False      # --dry: the oracle side only (checks the script's own mechanics where there is no GPU)


def one_case(rng, stats):
    sps, p = draw_params(rng)
    if rng.integers(0, 6) == 0:          # round 4: at and below one sample per symbol step (several symbols from one offset, floor(mu) = 0)
        sps = float(rng.choice([1.02, 1.0, 0.9, 0.6, 0.35, 0.2, 0.12]))          # (below 0.27 + 0.0176: 4-channel workgroups with the 1024-deep symbol ring)
        p["samplerate"] = 18000.0 * sps
    generic = 0
    if rng.integers(0, 10) == 0:         # filters beyond the 72 taps of the fused kernel's regular rows: its long rows, or the generic kernel
        p["rrc_tap_count"] = int(rng.integers(73, 130))
        generic = B.FLAG_GENERIC_KERNEL if rng.integers(0, 3) == 0 else 0
    Cn = int(rng.integers(1, 71))
    tm = bool(rng.integers(0, 2))
    shape = int(rng.choice([0, B.FLAG_WIDE_WORKGROUPS, B.FLAG_NARROW_WORKGROUPS, B.FLAG_SMALL_WORKGROUPS]))
    quality = bool(rng.integers(0, 3) == 0)
    quirks = bool(rng.integers(0, 3) == 0)
    constel = bool(rng.integers(0, 3) == 0)       # the plugin's constellation tap riding along (tetra_demod_get_constellation)
    chunks = [int(rng.choice([0, 1, 5, 31, 32, 33, 63, 64, 65, 180, 255, 700, 1500, 3000])) for _ in range(5)]
    N = max(sum(chunks), 1)
    seed = int(rng.integers(0, 1 << 30))
    iq, _, _ = pkg.synth.gen_batch(Cn, N, base_seed=seed, sps=max(sps, 1.02))
    for c in range(Cn):
        if rng.integers(0, 8) == 0:
            iq[c] = special_channel(rng, int(rng.integers(0, 4)), N)
    desc = dict(params=p, C=Cn, time_major=tm, shape=shape | generic, quality=quality, quirks=quirks, constellation=constel, chunks=chunks, seed=seed)
    try:
        orcs = [oracle.Oracle(oracle_cfg(p)) for _ in range(Cn)]
    except ValueError:
        orcs = None      # outside the oracle's own limits (it mirrors the library's): the library must refuse it too
    if DRY:
        if orcs is None:
            stats["refused"] += 1
            return None
        pos = 0
        for n in chunks:
            for c in range(Cn):
                stats["bits"] += int(orcs[c].process(iq[c, pos:pos + n], stages=True)["bits"].size)
                stats["channel_calls"] += 1
            pos += n
        return None
    try:
        d = pkg.Demodulator(Cn, 3000, layout=B.LAYOUT_TIME_MAJOR if tm else B.LAYOUT_CHANNEL_MAJOR,
                            flags=shape | generic | (B.FLAG_QUALITY if quality else 0) | (B.FLAG_REFERENCE_QUIRKS if quirks else 0) |
                            (B.FLAG_CONSTELLATION if constel else 0), **p)
    except B.TetraDemodError as e:
        stats["refused"] += 1
        # what the library refuses, the oracle's design must not be asked to run either: only the documented limits
        assert e.status == -2, (desc, str(e))
        return None
    if orcs is None:
        return dict(desc, what="the library accepted parameters the oracle refuses")
    pos = 0
    cd_streams = [np.zeros(0, np.complex64) for _ in range(Cn)]
    for k, n in enumerate(chunks):
        blk = iq[:, pos:pos + n]
        want_sym = bool(rng.integers(0, 2))
        bits, nb, sym = d.process(np.ascontiguousarray(blk.T) if tm else blk, want_sym=want_sym)
        for c in range(Cn):
            r = orcs[c].process(blk[c], stages=True)
            stats["channel_calls"] += 1
            stats["bits"] += int(r["bits"].size)
            if nb[c] != r["bits"].size:
                return dict(desc, call=k, ch=c, what="n_bits", gpu=int(nb[c]), ref=int(r["bits"].size))
            if not np.array_equal(bits[c][:nb[c]], r["bits"]):
                return dict(desc, call=k, ch=c, what="bits", first=int(np.flatnonzero(bits[c][:nb[c]] != r["bits"])[0]))
            if want_sym and not np.array_equal(_u32(sym[c][:nb[c] // 2]), _u32(r["sym"])):
                return dict(desc, call=k, ch=c, what="sym")
            if constel:
                cd_streams[c] = np.concatenate([cd_streams[c], r["sym"]])
        if constel:
            cblk, cnb = d.constellation()
            stats["constellation_checks"] = stats.get("constellation_checks", 0) + Cn
            for c in range(Cn):
                done = cd_streams[c].size // 1024
                want = cd_streams[c][(done - 1) * 1024:done * 1024] if done else np.zeros(1024, np.complex64)
                if cnb[c] != done or not np.array_equal(_u32(cblk[c]), _u32(want)):
                    return dict(desc, call=k, ch=c, what="constellation", gpu_blocks=int(cnb[c]), ref_blocks=int(done))
        if quality:
            err, _ = d.quality()
            for c in range(Cn):
                e0 = float(orcs[c].st.standarderr)
                if not (abs(float(err[c]) - e0) < 2e-6 or (np.isnan(err[c]) and np.isnan(e0))):
                    return dict(desc, call=k, ch=c, what="standarderr", gpu=float(err[c]), ref=e0)
        if k == 1:      # a loop setter in the middle of the stream, both sides
            name = LOOP_PARAMS[int(rng.integers(0, len(LOOP_PARAMS)))]
            cur = getattr(orcs[0].cfg, name)
            val = float(cur * rng.uniform(0.5, 1.5))
            try:
                d.set_param(name, val)
            except B.TetraDemodError as e:
                assert e.status == -2, (desc, name, val, str(e))
            else:
                for o in orcs:
                    o.set_param(B.PARAMS[name], val, quirks=quirks)
        if k == 3 and rng.integers(0, 3) == 0 and orcs[0].ntaps <= 72 and int(orcs[0].tab.ntaps_be) <= 72:
            # round 5: caller-designed tables on the live handle (tetra_demod_set_tables): another RRC of another length (regular rows
            # before and after), scaled band-edge filters of the same length, a perturbed bank -- both sides
            import ctypes as C
            nt_old = orcs[0].ntaps
            nt = int(rng.integers(2, 73))
            rrc = (rng.standard_normal(nt) / nt).astype(np.float32) if rng.integers(0, 2) else None
            a, b = orcs[0].bandedge_taps()
            be = np.stack([a * np.float32(rng.uniform(0.5, 1.5)), b * np.float32(rng.uniform(0.5, 1.5))]).astype(np.float32) if rng.integers(0, 2) else None
            bank = (orcs[0].interp_bank() * (1 + 0.02 * rng.standard_normal((128, 8)))).astype(np.float32) if rng.integers(0, 2) else None
            if rrc is not None or be is not None or bank is not None:
                d.set_tables(rrc_taps=rrc, bandedge_taps=be, interp_bank=bank)
                for o in orcs:
                    if rrc is not None:
                        o.tab.ntaps = nt
                        o.tab.cfg.rrc_tap_count = nt
                        o.cfg.rrc_tap_count = nt
                        for i, v in enumerate(rrc):
                            o.tab.rrc[i] = float(v)
                        if quirks and nt > nt_old:
                            oracle.lib().tetra_oracle_rrc_taps_grown(C.byref(o.st), nt_old)
                    if be is not None:
                        for i in range(be.shape[1]):
                            o.tab.be_a[i] = float(be[0, i])
                            o.tab.be_b[i] = float(be[1, i])
                    if bank is not None:
                        for ph in range(128):
                            for t in range(8):
                                o.tab.bank[ph][t] = float(bank[ph, t])
                stats["set_tables"] = stats.get("set_tables", 0) + 1
        if k == 2:
            c = int(rng.integers(0, Cn))
            d.reset(c)
            if quirks:
                orcs[c].reset_reference()
            else:
                orcs[c].reset()
                cd_streams[c] = np.zeros(0, np.complex64)      # without the quirks flag a reset restarts the channel's tap too
        pos += n
    for c in range(0, Cn, max(1, Cn // 6)):
        st, o = d.get_state(c), orcs[c].st
        for f in ("agc_gain", "fll_phase", "fll_freq", "mu", "omega", "costas_phase", "costas_freq", "ph2"):
            if np.float32(getattr(st, f)).tobytes() != np.float32(getattr(o, f)).tobytes():
                return dict(desc, ch=c, what="state." + f, gpu=float(getattr(st, f)), ref=float(getattr(o, f)))
        if st.offset != o.offset or st.prev != o.prev:
            return dict(desc, ch=c, what="state.offset/prev")
    if d.overruns() != 0:
        return dict(desc, what="overruns", n=d.overruns())
    d.close()
    return None


_POOL = {}


def plan_case(rng, stats):
    """The launch plan over the channel axis: C log-uniform in 1 ... 20000 on the automatic plan (whole rounds of 32-channel
    workgroups, the rest in 32-, 16- or 4-channel ones: tetra_demod_create), two short calls with carried state, against the
    oracle's batch entry (OpenMP over channels)."""
    sps = float(rng.choice([2.0, 2.0, 50000 / 18000]))
    if sps not in _POOL:
        _POOL[sps] = pkg.synth.gen_batch(256, 800, base_seed=4242, sps=sps)[0]
    pool = _POOL[sps]
    Cn = int(np.exp(rng.uniform(0, np.log(20000.0))))
    tm = bool(rng.integers(0, 2))
    n1, n2 = int(rng.integers(1, 401)), int(rng.integers(1, 401))
    idx = rng.integers(0, 256, Cn)
    scale = (rng.uniform(0.2, 1.5, Cn) * np.exp(1j * rng.uniform(-np.pi, np.pi, Cn))).astype(np.complex64)
    iq = np.ascontiguousarray(pool[idx, :n1 + n2] * scale[:, None])
    p = dict(symbolrate=18000.0, samplerate=18000.0 * sps)
    if rng.integers(0, 8) == 0:          # the long rows' plan: 4-channel workgroups up to 4 channels per CU, 16-channel ones beyond
        p["rrc_tap_count"] = int(rng.integers(73, 130))
    desc = dict(mode="plan", C=Cn, sps=sps, time_major=tm, n=[n1, n2], params=p)
    d = pkg.Demodulator(Cn, 400, layout=B.LAYOUT_TIME_MAJOR if tm else B.LAYOUT_CHANNEL_MAJOR, **p)
    states = None
    pos = 0
    for k, n in enumerate((n1, n2)):
        blk = np.ascontiguousarray(iq[:, pos:pos + n])
        bits, nb, _ = d.process(np.ascontiguousarray(blk.T) if tm else blk)
        rb, rnb, _, states = oracle.process_batch(blk, cfg=oracle_cfg(p), states=states, stride=bits.shape[1])
        stats["channel_calls"] += Cn
        stats["bits"] += int(rnb.sum())
        if not np.array_equal(nb, rnb):
            c = int(np.flatnonzero(nb != rnb)[0])
            return dict(desc, call=k, ch=c, what="n_bits", gpu=int(nb[c]), ref=int(rnb[c]))
        m = np.arange(bits.shape[1])[None, :] < nb[:, None]
        if not np.array_equal(bits[m], rb[m]):
            c = int(np.flatnonzero(((bits != rb) & m).any(axis=1))[0])
            return dict(desc, call=k, ch=c, what="bits")
        pos += n
    d.close()
    return None


class Pinned:
    """Page-locked host arrays from tetra_demod_host_alloc (no torch in this script)."""

    def __init__(self):
        self.lib = B.load_library()
        self.ptrs = []

    def array(self, shape, dtype):
        import ctypes as C
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = self.lib.tetra_demod_host_alloc(max(n, 1))
        assert ptr
        self.ptrs.append(ptr)
        return np.frombuffer((C.c_uint8 * max(n, 1)).from_address(ptr), dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        for ptr in self.ptrs:
            self.lib.tetra_demod_host_free(ptr)
        self.ptrs = []


def async_case(rng, stats):
    """tetra_demod_process_async: call lengths around its time-chunk boundaries (n / 4096 chunks, at most 8, lengths rounded to
    32), float / int16 / int8 input, both layouts, three calls of which two are in flight together, rates that change the row
    length -- against the oracle on the values the GPU converts to."""
    sps = float(rng.choice([2.0, 50000 / 18000, 1.4]))
    Cn = int(rng.integers(1, 65))
    tm = bool(rng.integers(0, 2))
    fmt = int(rng.choice([B.IQ_CF32, B.IQ_CS16, B.IQ_CS8]))
    def length():
        k = int(rng.integers(0, 10))
        return max(1, min(40000, k * 4096 + int(rng.choice([-33, -32, -1, 0, 1, 31, 32, 33, 100, 2000]))))
    sizes = [length() for _ in range(3)]
    iq, _, _ = pkg.synth.gen_batch(Cn, sum(sizes), base_seed=int(rng.integers(0, 1 << 30)), sps=sps, amp=0.5)
    f = iq.view(np.float32)
    if fmt == B.IQ_CS16:
        raw = np.clip(np.round(f * 32768.0), -32768, 32767).astype(np.int16)
        val = (raw.astype(np.float32) / np.float32(32768.0)).view(np.complex64)
    elif fmt == B.IQ_CS8:
        raw = np.clip(np.round(f * 128.0), -128, 127).astype(np.int8)
        val = (raw.astype(np.float32) / np.float32(128.0)).view(np.complex64)
    else:
        raw, val = f, iq
    raw = raw.reshape(Cn, -1, 2)
    p = dict(symbolrate=18000.0, samplerate=18000.0 * sps)
    if rng.integers(0, 8) == 0:
        p["rrc_tap_count"] = int(rng.integers(73, 130))
    desc = dict(mode="async", C=Cn, sps=sps, time_major=tm, fmt=fmt, sizes=sizes, params=p)
    d = pkg.Demodulator(Cn, max(sizes), layout=B.LAYOUT_TIME_MAJOR if tm else B.LAYOUT_CHANNEL_MAJOR, **p)
    pin = Pinned()
    keep, pos = [], 0
    for k, n in enumerate(sizes):
        blk = raw[:, pos:pos + n]
        src = np.swapaxes(blk, 0, 1) if tm else blk
        host_in = pin.array(src.shape, raw.dtype)
        host_in[...] = src
        stride = d.bits_stride(n)
        bits = pin.array((Cn, stride), np.uint8)
        bits[...] = 7
        nb = pin.array((Cn,), np.int32)
        nb[...] = -1
        d.process_async(host_in.ctypes.data, fmt, n, bits.ctypes.data, stride, nb.ctypes.data)
        keep.append((bits, nb, pos, n, stride))
        pos += n
        if k == 0:
            d.wait()                # calls 1 and 2 are then in flight together
    d.wait()
    states = None
    for k, (bits, nb, p0, n, stride) in enumerate(keep):
        rb, rnb, _, states = oracle.process_batch(np.ascontiguousarray(val[:, p0:p0 + n]), cfg=oracle_cfg(p), states=states, stride=stride)
        stats["channel_calls"] += Cn
        stats["bits"] += int(rnb.sum())
        if not np.array_equal(nb, rnb):
            c = int(np.flatnonzero(nb != rnb)[0])
            return dict(desc, call=k, ch=c, what="n_bits", gpu=int(nb[c]), ref=int(rnb[c]))
        m = np.arange(stride)[None, :] < rnb[:, None]
        if not np.array_equal(bits[m], rb[m]):
            c = int(np.flatnonzero(((bits != rb) & m).any(axis=1))[0])
            return dict(desc, call=k, ch=c, what="bits")
    d.close()
    pin.free()
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cases", type=int, default=1 << 30)
    ap.add_argument("--dry", action="store_true")
    ap.add_argument("--mode", choices=["chain", "plan", "async"], default="chain")
    a = ap.parse_args()
    global DRY
    DRY = a.dry
    rng = np.random.default_rng(a.seed)
    stats = dict(cases=0, refused=0, channel_calls=0, bits=0)
    failures = []
    t0 = time.time()
    kinds = {}
    while time.time() - t0 < a.seconds and stats["cases"] < a.max_cases and len(failures) < 12:
        f = plan_case(rng, stats) if a.mode == "plan" else async_case(rng, stats) if a.mode == "async" else one_case(rng, stats)
        stats["cases"] += 1
        if f is not None:
            kinds[f["what"]] = kinds.get(f["what"], 0) + 1
            if kinds[f["what"]] <= 3:          # at most three examples of a kind; the run goes on to find other kinds
                failures.append(f)
    stats["failed_cases_by_kind"] = kinds
    stats["seconds"] = round(time.time() - t0, 1)
    stats["seed"] = a.seed
    stats["mode"] = a.mode
    stats["failures"] = failures
    print(json.dumps(stats, default=str))
    return 1 if kinds else 0


if __name__ == "__main__":
    sys.exit(main())
