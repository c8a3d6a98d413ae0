"""ctypes binding of tests/emul/chan_emul.cpp (host emulation of the channeliser's FFT kernel; TEST TOOL)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_SO = os.path.join(_HERE, "libchan_emul.so")
_lib = None


def build():
    deps = [os.path.join(_HERE, "chan_emul.cpp"), os.path.join(_ROOT, "sdrpp-tetra-demodulator_amd", "csrc", "chan_fft_core.hpp")]
    if not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", deps[0], "-o", _SO], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.chan_fft_emul.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
        L.chan_fft_emul.restype = C.c_int
        L.chan_fft_emul_fmt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
        L.chan_fft_emul_fmt.restype = C.c_int
        L.chan_fft32.argtypes = [C.c_void_p, C.c_void_p]
        L.chan_dft25.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class ChanFftEmul:
    """The FFT kernel's arithmetic for M = 800, D = 400, P taps per channel, with carried history and sub-frame phase."""

    def __init__(self, P, proto):
        self.P, self.L = P, 800 * P
        self.h = np.ascontiguousarray(proto, np.float32)
        self.hist = np.zeros(self.L - 1, np.complex64)
        self.phase, self.consumed = 0, 0

    def process(self, x):
        """x: complex64 samples, or integer I / Q pairs [n][2] int16 / int8 (tetra_chan_process_device_cs16 / _cs8)."""
        x = np.ascontiguousarray(x)
        if x.dtype == np.int16 or x.dtype == np.int8:
            fmt = 1 if x.dtype == np.int16 else 2
            xc = (x[:, 0].astype(np.float32) + 1j * x[:, 1].astype(np.float32)).astype(np.complex64) / np.float32(32768 if fmt == 1 else 128)
        else:
            fmt, x = 0, np.ascontiguousarray(x, np.complex64)
            xc = x
        frames = (self.phase + len(x)) // 400
        out = np.zeros((max(frames, 1), 800), np.complex64)
        # exact-size buffers in their own allocations (no slack behind the new samples: the kernel must not read past them)
        xs = x.copy() if len(x) else np.zeros((1, 2), x.dtype) if fmt else np.zeros(1, np.complex64)
        got = lib().chan_fft_emul_fmt(self.hist.ctypes.data, xs.ctypes.data, fmt, len(x), self.P, self.phase, self.consumed, self.h.ctypes.data,
                                      out.ctypes.data)
        assert got == frames
        x = xc
        self.hist = np.concatenate([self.hist, x])[len(x):].copy()
        self.phase = (self.phase + len(x)) % 400
        self.consumed += len(x)
        return out[:frames]


def fft32(x):
    x = np.ascontiguousarray(x, np.complex64)
    y = np.zeros(32, np.complex64)
    lib().chan_fft32(x.ctypes.data, y.ctypes.data)
    return y


def dft25(x):
    x = np.ascontiguousarray(x, np.complex64)
    y = np.zeros(25, np.complex64)
    lib().chan_dft25(x.ctypes.data, y.ctypes.data)
    return y
