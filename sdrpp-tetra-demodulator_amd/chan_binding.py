"""ctypes binding of the channeliser front-end C ABI (include/tetra_chan.h)."""
import ctypes as C

import numpy as np

from .binding import TetraDemodError, load_library

CHAN_EXPORTS = ["tetra_chan_default_config", "tetra_chan_create", "tetra_chan_destroy", "tetra_chan_frames_for",
                "tetra_chan_process_device", "tetra_chan_process", "tetra_chan_reset", "tetra_chan_get_prototype",
                "tetra_chan_last_kernel_ms", "tetra_chan_process_device_cs16", "tetra_chan_process_device_cs8"]
RESAMP_EXPORTS = ["tetra_resamp_default_config", "tetra_resamp_create", "tetra_resamp_destroy", "tetra_resamp_frames_for",
                  "tetra_resamp_process_device", "tetra_resamp_process", "tetra_resamp_reset", "tetra_resamp_get_prototype",
                  "tetra_resamp_last_kernel_ms"]


class ChanConfig(C.Structure):
    _fields_ = [("n_channels", C.c_int32), ("taps_per_channel", C.c_int32), ("decimation", C.c_int32),
                ("max_in", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32),
                ("cutoff_rel", C.c_double), ("prototype", C.c_void_p)]


class ResampConfig(C.Structure):
    _fields_ = [("n_channels", C.c_int32), ("interp", C.c_int32), ("decim", C.c_int32), ("taps_per_phase", C.c_int32),
                ("max_in", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32),
                ("cutoff_rel", C.c_double), ("kaiser_beta", C.c_double), ("prototype", C.c_void_p)]


_ready = False


def _lib():
    global _ready
    L = load_library()
    if not _ready:
        vp, i32 = C.c_void_p, C.c_int
        L.tetra_chan_default_config.argtypes = [C.POINTER(ChanConfig)]
        L.tetra_chan_create.argtypes = [C.POINTER(ChanConfig), C.POINTER(vp)]
        L.tetra_chan_destroy.argtypes = [vp]
        L.tetra_chan_frames_for.argtypes = [vp, i32]
        L.tetra_chan_process_device.argtypes = [vp, vp, i32, vp, C.POINTER(i32), vp]
        L.tetra_chan_process.argtypes = [vp, vp, i32, vp, C.POINTER(i32)]
        L.tetra_chan_process_device_cs16.argtypes = [vp, vp, i32, vp, C.POINTER(i32), vp]
        L.tetra_chan_process_device_cs8.argtypes = [vp, vp, i32, vp, C.POINTER(i32), vp]
        L.tetra_chan_reset.argtypes = [vp]
        L.tetra_chan_get_prototype.argtypes = [vp, vp]
        L.tetra_chan_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.tetra_resamp_default_config.argtypes = [C.POINTER(ResampConfig)]
        L.tetra_resamp_create.argtypes = [C.POINTER(ResampConfig), C.POINTER(vp)]
        L.tetra_resamp_destroy.argtypes = [vp]
        L.tetra_resamp_frames_for.argtypes = [vp, i32]
        L.tetra_resamp_process_device.argtypes = [vp, vp, i32, vp, C.POINTER(i32), vp]
        L.tetra_resamp_process.argtypes = [vp, vp, i32, vp, C.POINTER(i32)]
        L.tetra_resamp_reset.argtypes = [vp]
        L.tetra_resamp_get_prototype.argtypes = [vp, vp]
        L.tetra_resamp_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        for n in CHAN_EXPORTS + RESAMP_EXPORTS:
            getattr(L, n).restype = i32
        _ready = True
    return L


class Channeliser:
    """M-channel analysis filter bank on one GPU; emits time-major frames [frames][M] complex64."""

    FLAG_VALU_DFT = 1      # TETRA_CHAN_FLAG_VALU_DFT: keep the direct-sum DFT kernel where a faster form exists (M = 800)
    FLAG_MATRIX_DFT = 2    # TETRA_CHAN_FLAG_MATRIX_DFT: M = 800 at D = M / 2 as 25 x 32 matrix products (round 4's kernel) instead of the mixed-radix FFT

    def __init__(self, n_channels=800, taps_per_channel=8, decimation=None, max_in=1 << 20, device=-1, cutoff_rel=1.2,
                 prototype=None, flags=0):
        self._lib = _lib()
        cfg = ChanConfig()
        self._lib.tetra_chan_default_config(C.byref(cfg))
        cfg.n_channels = n_channels
        cfg.taps_per_channel = taps_per_channel
        cfg.decimation = decimation if decimation is not None else n_channels // 2
        cfg.max_in = max_in
        cfg.device = device
        cfg.cutoff_rel = cutoff_rel
        cfg.reserved = flags
        keep = None
        if prototype is not None:
            keep = np.ascontiguousarray(prototype, np.float32)
            cfg.prototype = keep.ctypes.data
        self.M, self.P, self.D = n_channels, taps_per_channel, cfg.decimation
        h = C.c_void_p()
        rc = self._lib.tetra_chan_create(C.byref(cfg), C.byref(h))
        if rc:
            raise TetraDemodError(rc, "tetra_chan_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tetra_chan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def frames_for(self, n_in):
        return int(self._lib.tetra_chan_frames_for(self._h, int(n_in)))

    def process(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        nf = self.frames_for(x.shape[0])
        out = np.zeros((max(nf, 1), self.M), np.complex64)
        got = C.c_int(0)
        rc = self._lib.tetra_chan_process(self._h, x.ctypes.data_as(C.c_void_p), x.shape[0],
                                          out.ctypes.data_as(C.c_void_p), C.byref(got))
        if rc:
            raise TetraDemodError(rc, "tetra_chan_process")
        return out[: got.value]

    def process_device(self, d_x, n_in, d_out, stream=None):
        def p(t):
            return C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        s = None
        if stream is not None:
            s = C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
        got = C.c_int(0)
        # the input format follows the tensor: complex64, or interleaved I / Q pairs [n][2] int16 / int8 (tetra_chan_process_device_cs16 / _cs8)
        name = {"torch.int16": "tetra_chan_process_device_cs16", "torch.int8": "tetra_chan_process_device_cs8"}.get(str(getattr(d_x, "dtype", "")),
                                                                                                                  "tetra_chan_process_device")
        rc = getattr(self._lib, name)(self._h, p(d_x), int(n_in), p(d_out), C.byref(got), s)
        if rc:
            raise TetraDemodError(rc, name)
        return got.value

    def reset(self):
        rc = self._lib.tetra_chan_reset(self._h)
        if rc:
            raise TetraDemodError(rc, "tetra_chan_reset")

    def prototype(self):
        h = np.zeros(self.M * self.P, np.float32)
        self._lib.tetra_chan_get_prototype(self._h, h.ctypes.data_as(C.c_void_p))
        return h

    def last_kernel_ms(self):
        v = C.c_float(0)
        rc = self._lib.tetra_chan_last_kernel_ms(self._h, C.byref(v))
        if rc:
            raise TetraDemodError(rc, "tetra_chan_last_kernel_ms")
        return v.value


class Resampler:
    """Rational resampler I / DN on time-major frames [n][C] complex64 on one GPU (include/tetra_chan.h, tetra_resamp_*): the
    18 / 25 stage between the 50 ksps channeliser and a demodulator at the plugin's 36 ksps."""

    FLAG_GENERIC = 1      # TETRA_RESAMP_FLAG_GENERIC: keep the run-time-ratio kernel where a specialised one exists
    FLAG_NARROW_UNITS = 2 # TETRA_RESAMP_FLAG_NARROW_UNITS: 8-byte lane units (one channel per lane) instead of 16-byte ones

    def __init__(self, n_channels=800, interp=18, decim=25, taps_per_phase=16, max_in=1 << 16, device=-1, cutoff_rel=1.0,
                 kaiser_beta=6.0, prototype=None, flags=0):
        self._lib = _lib()
        cfg = ResampConfig()
        self._lib.tetra_resamp_default_config(C.byref(cfg))
        cfg.n_channels, cfg.interp, cfg.decim, cfg.taps_per_phase = n_channels, interp, decim, taps_per_phase
        cfg.max_in, cfg.device, cfg.flags = max_in, device, flags
        cfg.cutoff_rel, cfg.kaiser_beta = cutoff_rel, kaiser_beta
        keep = None
        if prototype is not None:
            keep = np.ascontiguousarray(prototype, np.float32)
            assert keep.size == interp * taps_per_phase
            cfg.prototype = keep.ctypes.data
        self.C, self.I, self.DN, self.T = n_channels, interp, decim, taps_per_phase
        h = C.c_void_p()
        rc = self._lib.tetra_resamp_create(C.byref(cfg), C.byref(h))
        if rc:
            raise TetraDemodError(rc, "tetra_resamp_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tetra_resamp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def frames_for(self, n_in):
        return int(self._lib.tetra_resamp_frames_for(self._h, int(n_in)))

    def process(self, x):
        x = np.ascontiguousarray(x, np.complex64).reshape(-1, self.C)
        nf = self.frames_for(x.shape[0])
        out = np.zeros((max(nf, 1), self.C), np.complex64)
        got = C.c_int(0)
        rc = self._lib.tetra_resamp_process(self._h, x.ctypes.data_as(C.c_void_p), x.shape[0], out.ctypes.data_as(C.c_void_p), C.byref(got))
        if rc:
            raise TetraDemodError(rc, "tetra_resamp_process")
        return out[: got.value]

    def process_device(self, d_in, n_in, d_out, stream=None):
        def p(t):
            return C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        s = None
        if stream is not None:
            s = C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
        got = C.c_int(0)
        rc = self._lib.tetra_resamp_process_device(self._h, p(d_in), int(n_in), p(d_out), C.byref(got), s)
        if rc:
            raise TetraDemodError(rc, "tetra_resamp_process_device")
        return got.value

    def reset(self):
        rc = self._lib.tetra_resamp_reset(self._h)
        if rc:
            raise TetraDemodError(rc, "tetra_resamp_reset")

    def prototype(self):
        h = np.zeros(self.I * self.T, np.float32)
        self._lib.tetra_resamp_get_prototype(self._h, h.ctypes.data_as(C.c_void_p))
        return h

    def last_kernel_ms(self):
        v = C.c_float(0)
        rc = self._lib.tetra_resamp_last_kernel_ms(self._h, C.byref(v))
        if rc:
            raise TetraDemodError(rc, "tetra_resamp_last_kernel_ms")
        return v.value
