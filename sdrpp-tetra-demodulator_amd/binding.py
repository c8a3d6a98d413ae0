"""ctypes binding of the C ABI (include/tetra_demod.h) -> libtetra_demod_hip.so.

Mirrors the reference's block interface for this path: one `Demodulator` = C copies of
PI4DQPSK -> DQPSKSymbolExtractor -> BitUnpacker (src/dsp/pi4dqpsk.h:27-81,
src/dsp/dqpsk_sym_extr.h:19-46, src/dsp/bit_unpacker.h:16-34) with init / process / reset /
setters.  There is no CPU fallback: if the HIP library is missing or no GPU is usable, every
call raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

LAYOUT_CHANNEL_MAJOR = 0
LAYOUT_TIME_MAJOR = 1
FLAG_WIDE_WORKGROUPS, FLAG_NARROW_WORKGROUPS, FLAG_SMALL_WORKGROUPS = 16, 32, 64
FLAG_RETIRED_TWO_KERNEL = 1      # refused by tetra_demod_create since ABI 2
FLAG_KEEP_RRC_OUT = 2
FLAG_QUALITY = 4
FLAG_REFERENCE_QUIRKS = 8
FLAG_CONSTELLATION = 256         # keep the constellation diagram's 1024-symbol blocks per channel (Demod.constellation)
CONSTELLATION_SYMBOLS = 1024
FLAG_GENERIC_KERNEL = 128        # filters of 73 .. 129 taps / loops below 0.27 samples per symbol in the one-lane-per-channel kernel

PARAMS = dict(symbolrate=0, samplerate=1, rrc_tap_count=2, rrc_beta=3, agc_rate=4, costas_bandwidth=5,
              fll_bandwidth=6, omega_gain=7, mu_gain=8, omega_rel_limit=9)

EXPORTS = [
    "tetra_demod_default_config", "tetra_demod_device_count", "tetra_demod_create", "tetra_demod_destroy",
    "tetra_demod_bits_stride", "tetra_demod_process_device", "tetra_demod_process", "tetra_demod_reset",
    "tetra_demod_set_param", "tetra_demod_get_state", "tetra_demod_set_state", "tetra_demod_get_tables",
    "tetra_demod_debug_read_rrc_out", "tetra_demod_last_kernel_ms", "tetra_demod_strerror",
    "tetra_demod_last_hip_error", "tetra_demod_abi_version", "tetra_demod_debug_selftest", "tetra_demod_kernel_ms_history", "tetra_demod_get_quality",
    "tetra_demod_bandedge_tap_count", "tetra_demod_process_async", "tetra_demod_wait", "tetra_demod_host_alloc",
    "tetra_demod_host_free", "tetra_demod_device_info", "tetra_demod_bits_stride_for", "tetra_demod_get_overruns",
    "tetra_demod_set_rrc_params", "tetra_demod_process_resident", "tetra_demod_debug_mfma_selftest", "tetra_demod_build_id",
    "tetra_demod_set_tables", "tetra_demod_get_constellation",
]
ERR_OVERRUN = -8
IQ_CF32, IQ_CS16, IQ_CS8 = 0, 1, 2


class Config(C.Structure):
    _fields_ = [
        ("n_channels", C.c_int32), ("max_samples", C.c_int32), ("layout", C.c_int32), ("device", C.c_int32),
        ("symbolrate", C.c_double), ("samplerate", C.c_double),
        ("rrc_tap_count", C.c_int32), ("flags", C.c_int32),
        ("rrc_beta", C.c_double), ("agc_rate", C.c_double), ("costas_bandwidth", C.c_double),
        ("fll_bandwidth", C.c_double), ("omega_gain", C.c_double), ("mu_gain", C.c_double),
        ("omega_rel_limit", C.c_double),
        ("rrc_taps", C.c_void_p), ("bandedge_taps", C.c_void_p), ("interp_bank", C.c_void_p),
    ]


class ChannelState(C.Structure):
    _fields_ = [
        ("agc_gain", C.c_float), ("fll_phase", C.c_float), ("fll_freq", C.c_float),
        ("mu", C.c_float), ("omega", C.c_float), ("offset", C.c_int32),
        ("costas_phase", C.c_float), ("costas_freq", C.c_float), ("ph2", C.c_float), ("prev", C.c_int32),
        ("hist", C.c_float * 160), ("ybuf", C.c_float * 14), ("rrc_valid", C.c_int32),
        ("hist_far", C.c_float * 96),
    ]


class TetraDemodError(RuntimeError):
    def __init__(self, status, what, hip=0):
        self.status = status
        self.hip = hip
        msg = "%s failed: %d (%s)" % (what, status, _strerror(status))
        if hip:
            msg += " [hipError %d]" % hip
        super().__init__(msg)


_lib = None


def load_library(rebuild_if_stale=True):
    """Load libtetra_demod_hip.so (building it with hipcc when missing/stale).  Raises if impossible.
    When the process also uses torch on the GPU, `import torch` BEFORE calling this: torch bundles its own HIP runtime, this
    library links the system one, and the first one loaded serves both (a torch imported second finds no GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    override = os.environ.get("TETRA_DEMOD_LIB")     # profiling scripts: an experimental build of the same ABI
    if override:
        path = os.path.abspath(override)
    elif rebuild_if_stale and _build.is_stale():
        path = _build.build()
    if not os.path.exists(path):
        raise RuntimeError("HIP library %s is missing; run __graft_entry__.build() (no CPU fallback exists)" % path)
    if not override and _build.lib_build_id(path) != _build.source_hash():
        # never run (or measure) a library built from other sources than the tree holds
        raise RuntimeError("HIP library %s was built from other sources than this tree holds (build id %s, sources %s); run "
                           "__graft_entry__.build()" % (path, _build.lib_build_id(path), _build.source_hash()))
    L = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int
    if hasattr(L, "tetra_demod_build_id"):          # (an override may be an older experimental build without it)
        L.tetra_demod_build_id.argtypes = []
        L.tetra_demod_build_id.restype = C.c_char_p
    L.tetra_demod_default_config.argtypes = [C.POINTER(Config)]
    L.tetra_demod_device_count.argtypes = []
    L.tetra_demod_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.tetra_demod_destroy.argtypes = [vp]
    L.tetra_demod_bits_stride.argtypes = [i32]
    L.tetra_demod_process_device.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
    L.tetra_demod_process.argtypes = [vp, vp, i32, vp, i32, vp, vp]
    L.tetra_demod_reset.argtypes = [vp, i32]
    L.tetra_demod_set_param.argtypes = [vp, i32, C.c_double]
    L.tetra_demod_get_state.argtypes = [vp, i32, C.POINTER(ChannelState)]
    L.tetra_demod_set_state.argtypes = [vp, i32, C.POINTER(ChannelState)]
    L.tetra_demod_get_tables.argtypes = [vp, C.POINTER(i32), vp, C.POINTER(i32), vp, vp, vp]
    L.tetra_demod_bits_stride_for.argtypes = [vp, i32]
    L.tetra_demod_get_overruns.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.tetra_demod_set_rrc_params.argtypes = [vp, i32, C.c_double]
    L.tetra_demod_process_resident.argtypes = [vp, vp, i32, vp, i32, vp, vp]
    L.tetra_demod_set_tables.argtypes = [vp, vp, i32, vp, i32, vp]
    L.tetra_demod_debug_mfma_selftest.argtypes = [vp, i32, i32, vp, vp, vp]
    L.tetra_demod_bandedge_tap_count.argtypes = [vp]
    L.tetra_demod_debug_read_rrc_out.argtypes = [vp, vp, i32]
    L.tetra_demod_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.tetra_demod_strerror.argtypes = [i32]
    L.tetra_demod_strerror.restype = C.c_char_p
    L.tetra_demod_last_hip_error.argtypes = [vp]
    L.tetra_demod_abi_version.argtypes = []
    L.tetra_demod_debug_selftest.argtypes = [vp, vp, vp]
    L.tetra_demod_kernel_ms_history.argtypes = [vp, i32, vp]
    L.tetra_demod_get_quality.argtypes = [vp, vp, vp]
    L.tetra_demod_get_constellation.argtypes = [vp, i32, i32, vp, vp]
    L.tetra_demod_process_async.argtypes = [vp, vp, i32, i32, vp, i32, vp]
    L.tetra_demod_wait.argtypes = [vp]
    L.tetra_demod_device_info.argtypes = [i32, C.POINTER(i32), C.POINTER(i32)]
    L.tetra_demod_host_alloc.argtypes = [C.c_size_t]
    L.tetra_demod_host_free.argtypes = [vp]
    for name in EXPORTS:
        if name not in ("tetra_demod_strerror", "tetra_demod_host_alloc", "tetra_demod_host_free", "tetra_demod_build_id"):
            getattr(L, name).restype = i32
    L.tetra_demod_host_alloc.restype = vp
    L.tetra_demod_host_free.restype = None
    _lib = L
    return L


def _strerror(status):
    try:
        return load_library(False).tetra_demod_strerror(status).decode()
    except Exception:  # pragma: no cover
        return "?"


def device_info(device=0):
    """(shader clock in kHz, compute units) of a device."""
    clk, cus = C.c_int32(0), C.c_int32(0)
    rc = load_library().tetra_demod_device_info(int(device), C.byref(clk), C.byref(cus))
    if rc != 0:
        raise TetraDemodError(rc, "tetra_demod_device_info")
    return int(clk.value), int(cus.value)


def default_config():
    cfg = Config()
    rc = load_library().tetra_demod_default_config(C.byref(cfg))
    if rc:
        raise TetraDemodError(rc, "tetra_demod_default_config")
    return cfg


def device_count():
    return int(load_library().tetra_demod_device_count())


def bits_stride(n_samples):
    """Handle-free row length: covers every handle at ~2 samples per symbol (tetra_demod_bits_stride); Demodulator.bits_stride
    is the one for a particular handle's rates."""
    return int(load_library().tetra_demod_bits_stride(int(n_samples)))


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Demodulator:
    """C batched reference chains on one GPU."""

    def __init__(self, n_channels=1, max_samples=65536, layout=LAYOUT_CHANNEL_MAJOR, device=-1,
                 rrc_taps=None, bandedge_taps=None, interp_bank=None, flags=0, **params):
        self._lib = load_library()
        cfg = default_config()
        cfg.n_channels = n_channels
        cfg.max_samples = max_samples
        cfg.layout = layout
        cfg.device = device
        cfg.flags = flags
        for k, v in params.items():
            if k not in PARAMS:
                raise TypeError("unknown parameter %r" % k)
            setattr(cfg, k, v)
        keep = []
        for name, arr in (("rrc_taps", rrc_taps), ("bandedge_taps", bandedge_taps), ("interp_bank", interp_bank)):
            if arr is not None:
                a = np.ascontiguousarray(arr, dtype=np.float32)
                keep.append(a)
                setattr(cfg, name, a.ctypes.data)
        self.cfg = cfg
        self.n_channels = n_channels
        self.max_samples = max_samples
        self.layout = layout
        h = C.c_void_p()
        rc = self._lib.tetra_demod_create(C.byref(cfg), C.byref(h))
        if rc:
            raise TetraDemodError(rc, "tetra_demod_create")
        self._h = h

    def _check(self, rc, what):
        if rc:
            raise TetraDemodError(rc, what, self._lib.tetra_demod_last_hip_error(self._h))

    def bits_stride(self, n_samples):
        """tetra_demod_bits_stride_for: the row length this handle's timing loop needs for calls of n_samples."""
        rc = int(self._lib.tetra_demod_bits_stride_for(self._h, int(n_samples)))
        if rc < 0:
            raise TetraDemodError(rc, "tetra_demod_bits_stride_for")
        return rc

    def overruns(self):
        """(channel, launch) events cut off at the row capacity since create (tetra_demod_get_overruns)."""
        v = C.c_longlong(0)
        self._check(self._lib.tetra_demod_get_overruns(self._h, C.byref(v)), "tetra_demod_get_overruns")
        return int(v.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tetra_demod_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- PI4DQPSK::process + DQPSKSymbolExtractor::process + BitUnpacker::process --------------------
    def process(self, iq, want_sym=False, allow_overrun=False):
        """Host path.  iq complex64 [C][N] (channel-major) or [N][C] (time-major handle).
        Returns (bits u8 [C][stride], n_bits i32 [C], sym complex64 [C][stride/2] or None).  TETRA_ERR_OVERRUN (a poisoned
        channel filled its row; everything was delivered) raises unless allow_overrun; self.last_status keeps it."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        if iq.ndim == 1:
            iq = iq[None, :] if self.layout == LAYOUT_CHANNEL_MAJOR else iq[:, None]
        n = iq.shape[1] if self.layout == LAYOUT_CHANNEL_MAJOR else iq.shape[0]
        cdim = iq.shape[0] if self.layout == LAYOUT_CHANNEL_MAJOR else iq.shape[1]
        if cdim != self.n_channels:
            raise ValueError("expected %d channels, got %d" % (self.n_channels, cdim))
        stride = max(bits_stride(n), self.bits_stride(n))
        bits = np.zeros((self.n_channels, stride), np.uint8)
        nb = np.zeros(self.n_channels, np.int32)
        sym = np.zeros((self.n_channels, stride // 2), np.complex64) if want_sym else None
        rc = self._lib.tetra_demod_process(self._h, _np_ptr(iq), n, _np_ptr(bits), stride, _np_ptr(nb), _np_ptr(sym))
        self.last_status = rc
        if not (allow_overrun and rc == ERR_OVERRUN):
            self._check(rc, "tetra_demod_process")
        return bits, nb, sym

    def process_async(self, iq_ptr, iq_format, n_samples, bits_ptr, bits_stride_, n_bits_ptr):
        """tetra_demod_process_async on raw host addresses (page-locked buffers, e.g. torch pinned tensors' data_ptr() or
        host_alloc()); the buffers must stay alive and untouched until wait()."""
        rc = self._lib.tetra_demod_process_async(self._h, C.c_void_p(int(iq_ptr)), int(iq_format), int(n_samples),
                                                 C.c_void_p(int(bits_ptr)), int(bits_stride_), C.c_void_p(int(n_bits_ptr)))
        self._check(rc, "tetra_demod_process_async")

    def wait(self):
        self._check(self._lib.tetra_demod_wait(self._h), "tetra_demod_wait")

    def process_resident(self, d_iq, n_samples, d_bits, bits_stride_, d_n_bits, d_sym=None):
        """tetra_demod_process_resident: device buffers, the handle's own stream, returns when the launch is done."""
        def p(x):
            if x is None:
                return None
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        self._check(self._lib.tetra_demod_process_resident(self._h, p(d_iq), int(n_samples), p(d_bits), int(bits_stride_),
                                                           p(d_n_bits), p(d_sym)), "tetra_demod_process_resident")

    def process_device(self, d_iq, n_samples, d_bits, bits_stride_, d_n_bits, d_sym=None, stream=None):
        """Device path: arguments are objects with .data_ptr() (torch tensors on this GPU) or ints."""
        def p(x):
            if x is None:
                return None
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        s = None
        if stream is not None:
            s = C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
        rc = self._lib.tetra_demod_process_device(self._h, p(d_iq), int(n_samples), p(d_bits), int(bits_stride_),
                                                  p(d_n_bits), p(d_sym), s)
        self._check(rc, "tetra_demod_process_device")

    # --- PI4DQPSK::reset and the setters -----------------------------------------------------------
    def reset(self, channel=-1):
        self._check(self._lib.tetra_demod_reset(self._h, channel), "tetra_demod_reset")

    def set_param(self, name, value):
        self._check(self._lib.tetra_demod_set_param(self._h, PARAMS[name], float(value)), "tetra_demod_set_param")

    def set_rrc_params(self, rrc_tap_count, rrc_beta):
        """PI4DQPSK::setRRCParams: both in one re-design."""
        self._check(self._lib.tetra_demod_set_rrc_params(self._h, int(rrc_tap_count), float(rrc_beta)), "tetra_demod_set_rrc_params")

    def set_tables(self, rrc_taps=None, bandedge_taps=None, interp_bank=None):
        """tetra_demod_set_tables: FIR::setTaps with caller-designed tables (rrc [n]; band-edge [2][n_be] = re, im of the lower
        filter; bank [128][8]); None keeps a table."""
        r = None if rrc_taps is None else np.ascontiguousarray(rrc_taps, np.float32)
        b = None if bandedge_taps is None else np.ascontiguousarray(bandedge_taps, np.float32).reshape(2, -1)
        k = None if interp_bank is None else np.ascontiguousarray(interp_bank, np.float32).reshape(128, 8)
        self._check(self._lib.tetra_demod_set_tables(self._h, _np_ptr(r), 0 if r is None else r.size, _np_ptr(b),
                                                     0 if b is None else b.shape[1], _np_ptr(k)), "tetra_demod_set_tables")

    def get_state(self, channel):
        st = ChannelState()
        self._check(self._lib.tetra_demod_get_state(self._h, channel, C.byref(st)), "tetra_demod_get_state")
        return st

    def set_state(self, channel, st):
        self._check(self._lib.tetra_demod_set_state(self._h, channel, C.byref(st)), "tetra_demod_set_state")

    def tables(self):
        nt, nbe = C.c_int(0), C.c_int(0)
        rrc = np.zeros(129, np.float32)          # TETRA_DEMOD_MAX_TAPS
        re = np.zeros(129, np.float32)
        im = np.zeros(129, np.float32)
        bank = np.zeros((128, 8), np.float32)
        self._check(self._lib.tetra_demod_get_tables(self._h, C.byref(nt), _np_ptr(rrc), C.byref(nbe), _np_ptr(re), _np_ptr(im),
                                                     _np_ptr(bank)), "tetra_demod_get_tables")
        assert nbe.value == self._lib.tetra_demod_bandedge_tap_count(self._h)
        return dict(rrc=rrc[:nt.value].copy(), be_re=re[:nbe.value].copy(), be_im=im[:nbe.value].copy(), bank=bank)

    def read_rrc_out(self, n_samples):
        y = np.zeros((self.n_channels, n_samples), np.complex64)
        self._check(self._lib.tetra_demod_debug_read_rrc_out(self._h, _np_ptr(y), n_samples),
                    "tetra_demod_debug_read_rrc_out")
        return y

    def selftest(self, in128):
        a = np.ascontiguousarray(in128, np.float32)
        assert a.shape == (128,)
        out = np.zeros(320, np.float32)
        self._check(self._lib.tetra_demod_debug_selftest(self._h, _np_ptr(a), _np_ptr(out)), "tetra_demod_debug_selftest")
        return out.reshape(5, 64)

    def mfma_selftest(self, a, b):
        """d = a . b on the matrix pipe (chained f32 MFMAs over ascending k from +0); a [M][K], b [K][M], M = 16 or 32."""
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        m, k = a.shape
        assert b.shape == (k, m)
        d = np.zeros((m, m), np.float32)
        self._check(self._lib.tetra_demod_debug_mfma_selftest(self._h, m, k, _np_ptr(a), _np_ptr(b), _np_ptr(d)),
                    "tetra_demod_debug_mfma_selftest")
        return d

    def quality(self):
        """(standarderr float32[C], sync bool[C]) -- DQPSKSymbolExtractor's public members per channel."""
        err = np.zeros(self.n_channels, np.float32)
        sync = np.zeros(self.n_channels, np.uint8)
        self._check(self._lib.tetra_demod_get_quality(self._h, _np_ptr(err), _np_ptr(sync)), "tetra_demod_get_quality")
        return err, sync.astype(bool)

    def constellation(self, first=0, count=None):
        """(blocks complex64[count][1024], n_blocks int32[count]) -- the last complete 1024-symbol block of the plugin's constellation
        tap per channel (src/main.cpp:85-89, :376-383) and the number of blocks completed; needs FLAG_CONSTELLATION."""
        count = self.n_channels - first if count is None else count
        blk = np.zeros((max(count, 0), CONSTELLATION_SYMBOLS), np.complex64)
        nb = np.zeros(max(count, 0), np.int32)
        self._check(self._lib.tetra_demod_get_constellation(self._h, first, count, _np_ptr(blk), _np_ptr(nb)),
                    "tetra_demod_get_constellation")
        return blk, nb

    def kernel_ms_history(self, n):
        """GPU ms of the n most recent process calls' launches, oldest first (HIP events on each call's stream)."""
        ms = np.zeros(n, np.float32)
        self._check(self._lib.tetra_demod_kernel_ms_history(self._h, n, _np_ptr(ms)), "tetra_demod_kernel_ms_history")
        return ms

    def last_kernel_ms(self):
        a = C.c_float(0)
        self._check(self._lib.tetra_demod_last_kernel_ms(self._h, C.byref(a)), "tetra_demod_last_kernel_ms")
        return a.value
