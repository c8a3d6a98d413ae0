"""ctypes binding of the CPU oracle (oracle/tetra_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.  PARITY UNPINNED
(see tetra_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtetra_oracle.so")

MAX_TAPS = 129
PHASES = 128
ITAPS = 8


class Cfg(C.Structure):
    _fields_ = [
        ("symbolrate", C.c_double), ("samplerate", C.c_double),
        ("rrc_tap_count", C.c_int), ("rrc_beta", C.c_double),
        ("agc_rate", C.c_double), ("costas_bandwidth", C.c_double),
        ("fll_bandwidth", C.c_double), ("omega_gain", C.c_double),
        ("mu_gain", C.c_double), ("omega_rel_limit", C.c_double),
    ]


class State(C.Structure):
    _fields_ = [
        ("agc_gain", C.c_float),
        ("fll_phase", C.c_float), ("fll_freq", C.c_float),
        ("hist", C.c_float * (2 * (MAX_TAPS - 1))),
        ("mu", C.c_float), ("omega", C.c_float),
        ("offset", C.c_int32),
        ("ybuf", C.c_float * (2 * (ITAPS - 1))),
        ("costas_phase", C.c_float), ("costas_freq", C.c_float),
        ("ph2", C.c_float),
        ("prev", C.c_uint8),
        ("errorbuf", C.c_float * 4096),
        ("errorptr", C.c_int32), ("errordisplayptr", C.c_int32),
        ("standarderr", C.c_float), ("sync", C.c_int32),
        ("rrc_valid", C.c_int32),
    ]


class Tables(C.Structure):
    _fields_ = [
        ("cfg", Cfg),
        ("ntaps", C.c_int),
        ("ntaps_be", C.c_int),
        ("rrc", C.c_float * MAX_TAPS),
        ("be_a", C.c_float * MAX_TAPS),
        ("be_b", C.c_float * MAX_TAPS),
        ("bank", (C.c_float * ITAPS) * PHASES),
        ("agc_rate", C.c_float), ("agc_set_point", C.c_float), ("agc_max_gain", C.c_float),
        ("fll_alpha", C.c_float), ("fll_beta", C.c_float),
        ("fll_min_freq", C.c_float), ("fll_max_freq", C.c_float),
        ("tr_alpha", C.c_float), ("tr_beta", C.c_float),
        ("tr_min_freq", C.c_float), ("tr_max_freq", C.c_float), ("tr_omega", C.c_float),
        ("costas_alpha", C.c_float), ("costas_beta", C.c_float),
        ("costas_min_freq", C.c_float), ("costas_max_freq", C.c_float),
    ]


def build(force=False):
    """Compile the oracle with gcc if the .so is missing or older than its sources."""
    srcs = [os.path.join(_HERE, f) for f in ("tetra_oracle.c", "tetra_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libtetra_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.tetra_oracle_default_cfg.argtypes = [C.POINTER(Cfg)]
        L.tetra_oracle_default_cfg.restype = None
        L.tetra_oracle_design.argtypes = [C.POINTER(Cfg), C.POINTER(Tables)]
        L.tetra_oracle_design.restype = C.c_int
        L.tetra_oracle_reset.argtypes = [C.POINTER(Tables), C.POINTER(State)]
        L.tetra_oracle_reset.restype = None
        L.tetra_oracle_reset_reference.argtypes = [C.POINTER(Tables), C.POINTER(State)]
        L.tetra_oracle_reset_reference.restype = None
        L.tetra_oracle_reset_timing.argtypes = [C.POINTER(Tables), C.POINTER(State)]
        L.tetra_oracle_reset_timing.restype = None
        L.tetra_oracle_set_param.argtypes = [C.POINTER(Tables), C.c_int, C.c_double, C.c_int]
        L.tetra_oracle_set_param.restype = C.c_int
        L.tetra_oracle_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.tetra_oracle_sincosf.restype = None
        vp = C.c_void_p
        L.tetra_oracle_process.argtypes = [C.POINTER(Tables), C.POINTER(State), C.c_int, vp, vp, vp, vp, vp, vp]
        L.tetra_oracle_process.restype = C.c_int
        L.tetra_oracle_process_mode.argtypes = [C.POINTER(Tables), C.POINTER(State), C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
        L.tetra_oracle_process_mode.restype = C.c_int
        L.tetra_oracle_process_batch.argtypes = [C.POINTER(Tables), C.POINTER(State), C.c_int, C.c_int,
                                                 C.c_int, C.c_int, vp, vp, C.c_int, vp, vp]
        L.tetra_oracle_process_batch.restype = C.c_int
        L.tetra_oracle_max_threads.restype = C.c_int
        L.tetra_oracle_fmaf_chain_matmul.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
        L.tetra_oracle_fmaf_chain_matmul.restype = None
        _lib = L
    return _lib


def default_cfg():
    cfg = Cfg()
    lib().tetra_oracle_default_cfg(C.byref(cfg))
    return cfg


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One channel of the reference chain: PI4DQPSK -> DQPSKSymbolExtractor -> BitUnpacker."""

    def __init__(self, cfg=None, reference_floats=False):
        """reference_floats: compute with the reference's own float recipe (libm phasors, plain sums, two complex band-edge
        dots: tetra_oracle.h TETRA_ORACLE_REFERENCE_FLOATS) instead of the arithmetic contract the kernels implement."""
        self.mode = 1 if reference_floats else 0
        self.cfg = cfg if cfg is not None else default_cfg()
        self.tab = Tables()
        rc = lib().tetra_oracle_design(C.byref(self.cfg), C.byref(self.tab))
        if rc != 0:
            raise ValueError("tetra_oracle_design failed: %d" % rc)
        self.st = State()
        self.reset()

    def reset(self):
        lib().tetra_oracle_reset(C.byref(self.tab), C.byref(self.st))

    def forget_far_history(self, keep=80):
        """What a launch of the product's FUSED kernel does to the delay line: only the newest `keep` = 80 FLL outputs are carried
        (its filters have at most 72 taps); the samples before them read as zeros afterwards -- to filters a setter grows beyond
        81 taps and to the RRC's visibility count.  Tests that switch a handle between the fused and the generic kernel call this
        at the same points."""
        n = 2 * ((MAX_TAPS - 1) - keep)
        for i in range(n):
            self.st.hist[i] = 0.0
        if self.st.rrc_valid > keep:
            self.st.rrc_valid = keep

    def reset_reference(self):
        """PI4DQPSK::reset as the reference does it (ph2, COMPLEX_FD's delay buffer and the slicer keep their values)."""
        lib().tetra_oracle_reset_reference(C.byref(self.tab), C.byref(self.st))

    def set_param(self, param_id, value, quirks=False):
        """A PI4DQPSK setter (ids = TETRA_PARAM_*); rate setters also reset this channel's timing loop."""
        old_ntaps = int(self.tab.ntaps)
        rc = lib().tetra_oracle_set_param(C.byref(self.tab), int(param_id), float(value), 1 if quirks else 0)
        if rc != 0:
            raise ValueError("tetra_oracle_set_param failed: %d" % rc)
        if quirks and int(self.tab.ntaps) > old_ntaps:
            lib().tetra_oracle_rrc_taps_grown(C.byref(self.st), old_ntaps)
        if param_id in (0, 1):
            lib().tetra_oracle_reset_timing(C.byref(self.tab), C.byref(self.st))

    # tables as numpy (copies)
    @property
    def ntaps(self):
        return int(self.tab.ntaps)

    def rrc_taps(self):
        return np.ctypeslib.as_array(self.tab.rrc)[: self.ntaps].copy()

    def bandedge_taps(self):
        nb = int(self.tab.ntaps_be)
        a = np.ctypeslib.as_array(self.tab.be_a)[:nb].copy()
        b = np.ctypeslib.as_array(self.tab.be_b)[:nb].copy()
        return a, b

    def interp_bank(self):
        return np.ctypeslib.as_array(self.tab.bank).copy()

    def process(self, iq, stages=False):
        """iq: complex64[count].  Returns dict(sym, dibits, bits[, x, y])."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        n = iq.shape[0]
        # symbols one call can emit: every symbol moves mu by at least tr_min_freq - |tr_alpha| (complex_fd.cpp:136-143); below one
        # sample per symbol the reference emits several symbols from one offset (floor(mu) = 0), so the bound exceeds n there
        step = float(self.tab.tr_min_freq) - abs(float(self.tab.tr_alpha))
        if not step > 0.01:
            raise ValueError("timing loop may stall (omega_min - |mu_gain| = %g): the reference would never leave the call" % step)
        cap = int((n + 1) / min(step, 1.0)) + 16
        sym = np.zeros(cap, np.complex64)
        dib = np.zeros(cap, np.uint8)
        bits = np.zeros(2 * cap, np.uint8)
        x = np.zeros(n, np.complex64) if stages else None
        y = np.zeros(n, np.complex64) if stages else None
        S = lib().tetra_oracle_process_mode(C.byref(self.tab), C.byref(self.st), self.mode, n, _ptr(iq), _ptr(x), _ptr(y),
                                            _ptr(sym), _ptr(dib), _ptr(bits))
        assert S >= 0
        out = dict(sym=sym[:S], dibits=dib[:S], bits=bits[: 2 * S])
        if stages:
            out["x"] = x
            out["y"] = y
        return out


def process_batch(iq, cfg=None, chunk=0, threads=0, want_sym=False, states=None, stride=None):
    """iq: complex64[C][N] channel-major.  Returns (bits[C][stride] u8, n_bits[C] i32, sym or None, states)."""
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    Cn, N = iq.shape
    if threads <= 0:
        threads = usable_cpus()[0]          # not OpenMP's default (every CPU the host shows, whatever the container's quota)
    cfg = cfg if cfg is not None else default_cfg()
    tab = Tables()
    rc = lib().tetra_oracle_design(C.byref(cfg), C.byref(tab))
    if rc != 0:
        raise ValueError("tetra_oracle_design failed: %d" % rc)
    if states is None:
        states = (State * Cn)()
        for c in range(Cn):
            lib().tetra_oracle_reset(C.byref(tab), C.byref(states[c]))
    stride = bits_stride(N) if stride is None else int(stride)
    bits = np.zeros((Cn, stride), np.uint8)
    nb = np.zeros(Cn, np.int32)
    sym = np.zeros((Cn, stride // 2), np.complex64) if want_sym else None
    rc = lib().tetra_oracle_process_batch(C.byref(tab), states, Cn, N, chunk, threads, _ptr(iq), _ptr(bits),
                                          stride, _ptr(nb), _ptr(sym))
    if rc != 0:
        raise RuntimeError("oracle batch overflow: %d" % rc)
    return bits, nb, sym, states


_FAST_LIB_PATH = os.path.join(_HERE, "libtetra_fast.so")
_fast = None


def fast_lib():
    """oracle/tetra_fast.c: the speed-oriented CPU port (bench.py's "port-fast" baseline leg).  Built -march=native ON THE HOST
    THAT RUNS IT (a prebuilt file from another machine is rebuilt: its instruction set may not match)."""
    global _fast
    if _fast is None:
        src = os.path.join(_HERE, "tetra_fast.c")
        stamp = _FAST_LIB_PATH + ".host"
        host = open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0] if os.path.exists("/proc/cpuinfo") else ""
        fresh = (os.path.exists(_FAST_LIB_PATH) and os.path.getmtime(src) <= os.path.getmtime(_FAST_LIB_PATH) and
                 os.path.exists(stamp) and open(stamp).read() == host)
        if not fresh:
            subprocess.run(["make", "-C", _HERE, "-B", "libtetra_fast.so"], check=True, stdout=subprocess.DEVNULL)
            with open(stamp, "w") as f:
                f.write(host)
        L = C.CDLL(_FAST_LIB_PATH)
        vp = C.c_void_p
        L.tetra_fast_process_batch.argtypes = [C.POINTER(Tables), C.POINTER(State), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp,
                                               C.c_int, vp]
        L.tetra_fast_process_batch.restype = C.c_int
        _fast = L
    return _fast


def fast_process_batch(iq, cfg=None, chunk=0, threads=0, states=None):
    """Like process_batch, through the speed-oriented port.  Returns (bits, n_bits, states)."""
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    Cn, N = iq.shape
    if threads <= 0:
        threads = usable_cpus()[0]
    cfg = cfg if cfg is not None else default_cfg()
    tab = Tables()
    if lib().tetra_oracle_design(C.byref(cfg), C.byref(tab)) != 0:
        raise ValueError("tetra_oracle_design failed")
    if states is None:
        states = (State * Cn)()
        for c in range(Cn):
            lib().tetra_oracle_reset(C.byref(tab), C.byref(states[c]))
    stride = bits_stride(N)
    bits = np.zeros((Cn, stride), np.uint8)
    nb = np.zeros(Cn, np.int32)
    if fast_lib().tetra_fast_process_batch(C.byref(tab), states, Cn, N, chunk, threads, _ptr(iq), _ptr(bits), stride, _ptr(nb)) != 0:
        raise RuntimeError("fast port: output row overflow")
    return bits, nb, states


def physical_cores():
    """Distinct (package, core) pairs of this host; falls back to the logical count."""
    try:
        ids, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    ids.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            ids.add((phys, core))
        return len(ids) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def usable_cpus():
    """CPUs this process may actually use: the scheduler affinity, cut by the cgroup CPU quota (a container that sees 256
    logical CPUs but has `cpu.max = 1600000 100000` gets 16 CPUs' worth of time; more threads than that only throttle)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.999)))
    return n, quota


def fmaf_chain_matmul(a, b):
    """d[i][j] = fmaf chain over ascending k from +0 (the contract's FIR sum), float32."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    d = np.zeros((a.shape[0], b.shape[1]), np.float32)
    lib().tetra_oracle_fmaf_chain_matmul(_ptr(a), _ptr(b), a.shape[0], b.shape[1], a.shape[1], _ptr(d))
    return d


def bits_stride(n_samples):
    """Output row stride used by the tests: >= 2*(N/1.94+1) bits, multiple of 16."""
    s = int(n_samples / 0.95) + 16
    return (s + 15) // 16 * 16


def max_threads():
    return int(lib().tetra_oracle_max_threads())


# ---------------------------------------------------------------------------------------------------------
# Channeliser front-end definition (oracle/chan_oracle.c) -- test infrastructure, like everything in here.
# ---------------------------------------------------------------------------------------------------------
_CHAN_LIB_PATH = os.path.join(_HERE, "libchan_oracle.so")
_chan = None


def chan_lib():
    global _chan
    if _chan is None:
        src = os.path.join(_HERE, "chan_oracle.c")
        if not os.path.exists(_CHAN_LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(_CHAN_LIB_PATH):
            subprocess.run(["make", "-C", _HERE, "-B", "libchan_oracle.so"], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(_CHAN_LIB_PATH)
        vp = C.c_void_p
        L.chan_oracle_prototype.argtypes = [C.c_int, C.c_int, C.c_double, vp]
        L.chan_oracle_prototype.restype = None
        L.chan_oracle_process.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                          C.c_int, vp, vp]
        L.chan_oracle_process.restype = C.c_int
        L.chan_oracle_process_channels.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                                   C.c_int, vp, vp, C.c_int, vp]
        L.chan_oracle_process_channels.restype = C.c_int
        L.resamp_oracle_prototype.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp]
        L.resamp_oracle_prototype.restype = None
        L.resamp_oracle_process.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                            C.c_int, vp, vp]
        L.resamp_oracle_process.restype = C.c_int
        _chan = L
    return _chan


class ChanOracle:
    """Definition-level analysis filter bank in double precision (slow: O(M * L) per frame)."""

    def __init__(self, M, P, D, cutoff_rel=1.2):
        self.M, self.P, self.D = M, P, D
        self.h = np.zeros(M * P, np.float32)
        chan_lib().chan_oracle_prototype(M, P, cutoff_rel, _ptr(self.h))
        self.hist = np.zeros(M * P - 1, np.complex64)
        self.phase = C.c_int(0)
        self.frame = C.c_int64(0)

    def process(self, x, channels=None):
        """All M channels, or -- channels = a list of channel indices -- only those (columns in that order): the definition is
        evaluated per channel, so a subset costs proportionally less."""
        x = np.ascontiguousarray(x, np.complex64)
        nf = (self.phase.value + x.shape[0]) // self.D
        sel = None if channels is None else np.ascontiguousarray(channels, np.int32)
        ncol = self.M if sel is None else int(sel.size)
        out = np.zeros((max(nf, 1), ncol), np.complex64)
        got = chan_lib().chan_oracle_process_channels(self.M, self.P, self.D, _ptr(self.h), _ptr(self.hist), C.byref(self.phase),
                                                      C.byref(self.frame), x.shape[0], _ptr(x), _ptr(out), ncol, _ptr(sel))
        return out[:got]


class ResampOracle:
    """Definition-level rational resampler I / DN on time-major frames [n][C] complex64, double-precision sums
    (oracle/chan_oracle.c: resamp_oracle_process).  Carries the T - 1 newest frames and the output position."""

    def __init__(self, C_, I=18, DN=25, T=16, cutoff_rel=1.0, beta=6.0, prototype=None):
        self.C, self.I, self.DN, self.T = C_, I, DN, T
        if prototype is not None:
            self.h = np.ascontiguousarray(prototype, np.float32)
            assert self.h.size == I * T
        else:
            self.h = np.zeros(I * T, np.float32)
            chan_lib().resamp_oracle_prototype(I, DN, T, cutoff_rel, beta, _ptr(self.h))
        self.hist = np.zeros((T - 1, C_), np.complex64)
        self.n_total = C.c_int64(0)
        self.m_next = C.c_int64(0)

    def frames_for(self, n_in):
        return int(((self.n_total.value + n_in) * self.I + self.DN - 1) // self.DN - self.m_next.value)

    def process(self, x):
        x = np.ascontiguousarray(x, np.complex64).reshape(-1, self.C)
        n_out = self.frames_for(x.shape[0])
        out = np.zeros((max(n_out, 1), self.C), np.complex64)
        got = chan_lib().resamp_oracle_process(self.I, self.DN, self.T, _ptr(self.h), self.C, _ptr(self.hist), C.byref(self.n_total),
                                               C.byref(self.m_next), x.shape[0], _ptr(x), _ptr(out))
        assert got == n_out
        return out[:got]


# ---------------------------------------------------------------------------------------------------------------------
# Burst synchroniser + burst demultiplexer restatement (oracle/burst_sync_oracle.c)
# ---------------------------------------------------------------------------------------------------------------------
_BSYNC_LIB_PATH = os.path.join(_HERE, "libbsync_oracle.so")
_bsync = None
RX_S_UNLOCKED, RX_S_KNOW_FSTART, RX_S_LOCKED = 0, 1, 2


def bsync_lib():
    global _bsync
    if _bsync is None:
        src = os.path.join(_HERE, "burst_sync_oracle.c")
        if not os.path.exists(_BSYNC_LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(_BSYNC_LIB_PATH):
            subprocess.run(["make", "-C", _HERE, "-B", "libbsync_oracle.so"], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(_BSYNC_LIB_PATH)
        vp = C.c_void_p
        L.bs_oracle_find_train_seq.argtypes = [vp, C.c_uint, C.c_uint32, C.POINTER(C.c_uint)]
        L.bs_oracle_find_train_seq.restype = C.c_int
        L.bs_oracle_reset.argtypes = [vp]
        L.bs_oracle_reset.restype = None
        L.bs_oracle_feed.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int]
        L.bs_oracle_feed.restype = C.c_int
        L.bs_oracle_demux.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.bs_oracle_demux.restype = C.c_int
        L.bs_oracle_state_size.restype = C.c_int
        L.ts_indicator_init.argtypes = [vp]
        L.ts_indicator_init.restype = None
        L.ts_indicator_feed.argtypes = [vp, vp, C.c_int]
        L.ts_indicator_feed.restype = None
        _bsync = L
    return _bsync


def bsync_find_train_seq(bits, end_of_in, mask=0x1f):
    b = np.ascontiguousarray(bits, np.uint8)
    assert b.size >= end_of_in + 21
    off = C.c_uint(0)
    t = bsync_lib().bs_oracle_find_train_seq(_ptr(b), int(end_of_in), int(mask), C.byref(off))
    return (t, int(off.value)) if t >= 0 else (-1, -1)


class BurstSyncOracle:
    """One channel's tetra_rx_state restatement.  feed() returns (frames uint8 [n][510], types int32 [n], bitnums
    uint32 [n]) for the frames the LOCKED state consumed during the call(s)."""

    def __init__(self):
        self._st = np.zeros(bsync_lib().bs_oracle_state_size(), np.uint8)

    @property
    def state(self):
        v = self._st[:16].view(np.uint32)
        return int(v[0]), int(v[1]), int(v[2]), int(v[3])     # state, bits_in_buf, bitbuf_start_bitnum, next_frame_start

    def feed(self, bits, chunk=1):
        b = np.ascontiguousarray(bits, np.uint8)
        cap = b.size // 510 + 16
        frames = np.zeros((cap, 512), np.uint8)
        types = np.zeros(cap, np.int32)
        bitnums = np.zeros(cap, np.uint32)
        n = bsync_lib().bs_oracle_feed(_ptr(self._st), _ptr(b), b.size, int(chunk), _ptr(frames), _ptr(types), _ptr(bitnums), cap)
        assert n <= cap
        return frames[:n, :510].copy(), types[:n].copy(), bitnums[:n].copy()


def bsync_demux(burst, train, tpsap, blk_num):
    b = np.ascontiguousarray(burst, np.uint8)
    out = np.zeros(432, np.uint8)
    n = bsync_lib().bs_oracle_demux(_ptr(b), int(train), int(tpsap), int(blk_num), _ptr(out))
    return out[:n].copy()


class TsIndicatorOracle:
    """The plugin's training-sequence indicator (src/main.cpp:385-414) for one channel, literal restatement."""

    def __init__(self):
        self._st = np.zeros(45 + 3 + 8, np.uint8)      # tsfind_buffer[45], padding, tsfound, symsbeforeexpire
        bsync_lib().ts_indicator_init(_ptr(self._st))

    def feed(self, bits):
        b = np.ascontiguousarray(bits, np.uint8)
        bsync_lib().ts_indicator_feed(_ptr(self._st), _ptr(b), int(b.size))
        v = self._st[48:56].view(np.int32)
        return bool(v[0]), int(v[1])
