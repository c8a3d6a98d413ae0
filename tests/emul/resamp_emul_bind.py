"""ctypes binding of tests/emul/resamp_emul.cpp (host emulation of the rational resampler's kernels; TEST TOOL)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_SO = os.path.join(_HERE, "libresamp_emul.so")
_lib = None


def build():
    deps = [os.path.join(_HERE, "resamp_emul.cpp"), os.path.join(_ROOT, "sdrpp-tetra-demodulator_amd", "csrc", "resamp_core.hpp")]
    if not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", deps[0], "-o", _SO], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp = C.c_void_p
        L.resamp_emul.argtypes = [C.c_int] * 6 + [vp, vp, vp, C.c_int, C.c_longlong, C.c_longlong, vp]
        L.resamp_emul.restype = C.c_int
        _lib = L
    return _lib


class ResampEmul:
    """The resampler kernels' arithmetic and index maps with the C ABI's carried state (delay line of T - 1 frames, positions)."""

    def __init__(self, n_channels, I, DN, T, proto, generic=False, W=None):
        self.C, self.I, self.DN, self.T = n_channels, I, DN, T
        self.W = W if W is not None else (4 if n_channels % 2 == 0 else 2)
        self.h = np.ascontiguousarray(proto, np.float32)
        self.generic = generic
        self.hist = np.zeros((T - 1, n_channels), np.complex64)
        self.n_total, self.m_next = 0, 0

    def process(self, x):
        x = np.ascontiguousarray(x, np.complex64).reshape(-1, self.C)
        n_in = x.shape[0]
        m1 = ((self.n_total + n_in) * self.I + self.DN - 1) // self.DN
        n_out = m1 - self.m_next
        # exact-size buffers in their own allocations, NaN-poisoned output: every stored element must be written, nothing beyond
        out = np.full((max(n_out, 1), self.C), np.nan + 0j, np.complex64)
        xs = x.copy() if n_in else np.zeros((1, self.C), np.complex64)
        got = lib().resamp_emul(self.I, self.DN, self.T, self.C, self.W, int(self.generic), self.h.ctypes.data, self.hist.ctypes.data,
                                xs.ctypes.data, n_in, self.n_total, self.m_next, out.ctypes.data)
        assert got == n_out, (got, n_out)
        self.hist = np.concatenate([self.hist, x])[n_in:].copy()
        self.n_total += n_in
        self.m_next = m1
        return out[:n_out]
